// jpeg.hip -- JPEG front end on the device: baseline files -> luma planes, for gfx950.
//
// The reference's image route takes JPEG uploads (src/modality/image.rs:54; decoders enabled at Cargo.toml:143) and
// decodes them inside the SDK call (image.rs:68-70, :176-179: imgfprint -> image::load_from_memory).  What the hash needs
// of a JPEG is its LUMA COMPONENT (DESIGN J1): the Y plane of a YCbCr file or the single plane of a greyscale one, at
// full resolution.  Chroma blocks are parsed -- the entropy-coded stream interleaves them -- and never transformed; no
// upsampling, no colour conversion.  The result is bit-equal to libjpeg's own luma output (out_color_space =
// JCS_GRAYSCALE, JDCT_ISLOW), which the oracle restates and tests/test_oracle_jpeg.py pins against Pillow.
//
// Four kernels, everything else is bookkeeping:
//   jpeg_scan_kernel   one wave per file.  Lane 0 walks the marker segments (SOF0/1, DQT, DHT, DRI, APP0/14, SOS) and
//                      records where things are; then the wave removes the byte stuffing of the entropy-coded data
//                      (FF 00 -> FF) 64 bytes per step -- ballot + prefix count compaction -- and cuts it at the RSTn
//                      markers into SEGMENTS (one per restart interval, each starting on a byte).
//   jpeg_huff_kernel   files WITH restart markers: one wave per file, one LANE per segment (Huffman decoding is serial
//                      within one).  Code tables are built once per file in LDS: a 9-bit look-ahead table per Huffman table
//                      + the canonical (mincode / maxcode / valptr) form for longer codes.  A lane keeps a 64-bit window of
//                      its segment (aligned dword loads, the next one always in flight) and decodes MCU by MCU; coefficients
//                      of luma blocks go, de-zigzagged, to the lane's 128 bytes of LDS and from there as eight 16-byte
//                      stores to the coefficient plane.  Chroma coefficients are decoded and dropped.
//   jpeg_huff_spec_kernel   files WITHOUT restart markers (what encoders write by default): one wave per file, the ONE
//                      segment decoded by 64 lanes at once through speculation on the stream's self-synchronisation --
//                      see the section's own comment; 2.8-5 x the serial lane (445 k against 160 k files/s at 1000 config-1
//                      files, 909 k against 292 k at 8000).
//   jpeg_idct_kernel   one THREAD per luma block over the whole batch: dequantisation + the accurate integer inverse DCT
//                      (IJG jidctint "islow": 13-bit fixed point, two 1-D passes of 8, all in registers), range limit,
//                      eight 8-byte row stores.
// Anything this file does not decode -- progressive / arithmetic / 12-bit / CMYK / RGB-coded files, several scans, luma
// coded below the MCU's resolution, 16-bit quantisation tables -- and any irregularity of the entropy-coded data gets
// UCFP_IMAGE_NEEDS_HOST: decoders differ in what they forgive, the host's decides.  No SOI at all: UCFP_E_MODALITY.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ucfp_hip.h"
#include "common.h"

namespace ucfp {

namespace {

constexpr int kMaxSeg = 1 << 20;          // restart intervals per file the segment table may hold (bounded by its allocation)

struct JpgInfo {
    int32_t status;
    uint32_t scan_off;      // offset of the entropy-coded data inside the file
    uint32_t clean_len;     // bytes of it after unstuffing
    uint32_t nseg;          // segments found (restart intervals)
    uint32_t restart;       // MCUs per restart interval (0: none)
    uint32_t dqt_off[4];    // file offset of each quantisation table's 64 bytes (0: absent)
    uint32_t dht_off[8];    // file offset of the 16 length counts of table (class << 2 | id) (0: absent)
    uint8_t ncomp, hmax, vmax, tq0;
    uint8_t hs[3], vs[3], td[3], ta[3];
};

__constant__ uint8_t c_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

__device__ __forceinline__ uint32_t be16(const uint8_t* p) { return (uint32_t)p[0] << 8 | p[1]; }

// ---- header walk (lane 0; a few dozen dependent byte loads) ----
__device__ int jpeg_parse(const uint8_t* p, size_t n, uint32_t width, uint32_t height, JpgInfo& J) {
    J.status = 0;
    J.scan_off = J.clean_len = J.nseg = J.restart = 0;
    for (int i = 0; i < 4; i++) J.dqt_off[i] = 0;
    for (int i = 0; i < 8; i++) J.dht_off[i] = 0;
    J.ncomp = J.hmax = J.vmax = J.tq0 = 0;
    if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) return UCFP_E_MODALITY;
    size_t pos = 2;
    bool have_sof = false, jfif = false;
    int adobe = -1;
    uint32_t w = 0, h = 0;
    uint8_t cid[3] = {0, 0, 0}, tq[3] = {0, 0, 0};
    for (;;) {
        if (pos + 4 > n) return have_sof ? UCFP_IMAGE_NEEDS_HOST : UCFP_E_MODALITY;
        if (p[pos] != 0xFF) return UCFP_IMAGE_NEEDS_HOST;
        while (pos < n && p[pos] == 0xFF) pos++;
        if (pos >= n) return UCFP_IMAGE_NEEDS_HOST;
        const int m = p[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return UCFP_IMAGE_NEEDS_HOST;
        if (pos + 2 > n) return UCFP_IMAGE_NEEDS_HOST;
        const size_t len = be16(p + pos);
        if (len < 2 || pos + len > n) return UCFP_IMAGE_NEEDS_HOST;
        const uint8_t* s = p + pos + 2;
        const size_t sl = len - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (have_sof || sl < 6) return UCFP_IMAGE_NEEDS_HOST;
            have_sof = true;
            if (s[0] != 8) return UCFP_IMAGE_NEEDS_HOST;
            h = be16(s + 1);
            w = be16(s + 3);
            J.ncomp = s[5];
            if (w == 0 || h == 0 || (J.ncomp != 1 && J.ncomp != 3) || sl < 6 + 3 * (size_t)J.ncomp) return UCFP_IMAGE_NEEDS_HOST;
            for (int c = 0; c < J.ncomp; c++) {
                cid[c] = s[6 + 3 * c];
                J.hs[c] = s[7 + 3 * c] >> 4;
                J.vs[c] = s[7 + 3 * c] & 15;
                tq[c] = s[8 + 3 * c];
                if (J.hs[c] < 1 || J.hs[c] > 4 || J.vs[c] < 1 || J.vs[c] > 4 || tq[c] > 3) return UCFP_IMAGE_NEEDS_HOST;
                if (J.hs[c] > J.hmax) J.hmax = J.hs[c];
                if (J.vs[c] > J.vmax) J.vmax = J.vs[c];
            }
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8) {
            return UCFP_IMAGE_NEEDS_HOST;       // progressive, lossless, hierarchical, arithmetic coding (incl. DAC)
        } else if (m == 0xC4) {
            size_t o = 0;
            while (o < sl) {
                if (o + 17 > sl) return UCFP_IMAGE_NEEDS_HOST;
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) return UCFP_IMAGE_NEEDS_HOST;
                int cnt = 0, code = 0;
                for (int l = 1; l <= 16; l++) {
                    const int b = s[o + l];
                    cnt += b;
                    code += b;
                    if (code > (1 << l)) return UCFP_IMAGE_NEEDS_HOST;     // over-subscribed: not a prefix code
                    code <<= 1;
                }
                if (cnt > 256 || o + 17 + (size_t)cnt > sl) return UCFP_IMAGE_NEEDS_HOST;
                J.dht_off[tc << 2 | th] = (uint32_t)(pos + 2 + o + 1);
                o += 17 + (size_t)cnt;
            }
        } else if (m == 0xDB) {
            size_t o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, t = s[o] & 15;
                if (pq != 0 || t > 3 || o + 65 > sl) return UCFP_IMAGE_NEEDS_HOST;
                J.dqt_off[t] = (uint32_t)(pos + 2 + o + 1);
                o += 65;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return UCFP_IMAGE_NEEDS_HOST;
            J.restart = be16(s);
        } else if (m == 0xE0) {
            if (sl >= 5 && s[0] == 'J' && s[1] == 'F' && s[2] == 'I' && s[3] == 'F' && s[4] == 0) jfif = true;
        } else if (m == 0xEE) {
            if (sl >= 12 && s[0] == 'A' && s[1] == 'd' && s[2] == 'o' && s[3] == 'b' && s[4] == 'e') adobe = s[11];
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1) return UCFP_IMAGE_NEEDS_HOST;
            const int ns = s[0];
            if (ns != J.ncomp || sl < 1 + 2 * (size_t)ns + 3) return UCFP_IMAGE_NEEDS_HOST;
            for (int c = 0; c < ns; c++) {
                if (s[1 + 2 * c] != cid[c]) return UCFP_IMAGE_NEEDS_HOST;
                J.td[c] = s[2 + 2 * c] >> 4;
                J.ta[c] = s[2 + 2 * c] & 15;
                if (J.td[c] > 3 || J.ta[c] > 3 || !J.dht_off[J.td[c]] || !J.dht_off[4 | J.ta[c]] || !J.dqt_off[tq[c]])
                    return UCFP_IMAGE_NEEDS_HOST;
            }
            if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return UCFP_IMAGE_NEEDS_HOST;
            J.scan_off = (uint32_t)(pos + len);
            break;
        }
        pos += len;
    }
    J.tq0 = tq[0];
    if (J.ncomp == 3) {
        // libjpeg's colour space rule (jdapimin.c): JFIF -> YCbCr; Adobe transform 1 -> YCbCr, else not; otherwise by ids
        bool ycc = true;
        if (jfif) ycc = true;
        else if (adobe >= 0) ycc = adobe == 1;
        else if (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B') ycc = false;
        if (!ycc || J.hs[0] != J.hmax || J.vs[0] != J.vmax) return UCFP_IMAGE_NEEDS_HOST;
    } else {
        J.hmax = J.vmax = 1;          // a one-component scan is not interleaved: one block per MCU (T.81 A.2.2)
        J.hs[0] = J.vs[0] = 1;
    }
    if (w != width || h != height) return UCFP_IMAGE_NEEDS_HOST;      // another geometry than the batch announced
    return 0;
}

// One wave per file.  clean: the file's unstuffed entropy-coded bytes (at the 16-byte rounded file offset of a buffer as
// large as the batch); seg: (max_seg + 2) offsets per file.
__global__ __launch_bounds__(64) void jpeg_scan_kernel(const uint8_t* __restrict__ jpg, const uint64_t* __restrict__ offsets, size_t n,
                                                      const UpItem* __restrict__ items, UpUniform uni,
                                                      uint8_t* __restrict__ clean, uint32_t* __restrict__ seg,
                                                      JpgInfo* __restrict__ info) {
    const size_t ent = blockIdx.x;                    // entry of the batch: info[ent]; the bytes are FILE item.file's
    if (ent >= n) return;
    const UpItem item = up_item(items, uni, ent);
    const size_t img = item.file;
    const uint32_t width = item.w, height = item.h, max_seg = item.max_seg;
    const int lane = threadIdx.x;
    const uint8_t* p = jpg + offsets[img];
    const size_t len = (size_t)(offsets[img + 1] - offsets[img]);
    __shared__ JpgInfo J;
    if (lane == 0) J.status = jpeg_parse(p, len, width, height, J);
    wave_lds_sync();
    int32_t status = J.status;
    uint8_t* out = clean + ((offsets[img] + 15) & ~(uint64_t)15);
    uint32_t* sg = seg + item.seg_off;
    uint32_t o = 0, nseg = 0;       // wave-uniform: clean bytes written, RSTn markers seen
    if (status == 0) {
        if (lane == 0) sg[0] = 0;
        const size_t s0 = J.scan_off;
        uint32_t prev = 0;          // the byte before this step's first byte (never FF at the start of the scan)
        bool ended = false;
        for (size_t base = s0; base < len && !ended && status == 0; base += 64) {
            const size_t i = base + lane;
            const uint32_t b = i < len ? p[i] : 0x100u;        // 0x100: beyond the file
            uint32_t before = (uint32_t)__shfl_up((int)b, 1, 64), after = (uint32_t)__shfl_down((int)b, 1, 64);
            if (lane == 0) before = prev;
            if (lane == 63) after = i + 1 < len ? p[i + 1] : 0x100u;
            const bool second = before == 0xFF;                 // (FF FF is refused, so a byte after FF is never a first FF)
            const bool ff = b == 0xFF && !second;
            const bool rst_next = after >= 0xD0 && after <= 0xD7;
            const bool is_end = (ff && after != 0x00 && !rst_next && after != 0xFF) || b == 0x100u;   // EOI, any other marker, end of file
            const bool is_bad = ff && after == 0xFF;           // fill bytes inside the scan: host
            const bool is_rst = second && b >= 0xD0 && b <= 0xD7;
            const uint64_t endm = __ballot(is_end);
            const uint32_t live = endm ? (uint32_t)__builtin_ctzll(endm) : 64u;     // lanes before the first end take part
            const bool on = (uint32_t)lane < live;
            if (__ballot(on && is_bad)) status = UCFP_IMAGE_NEEDS_HOST;
            const bool keep = on && !(second && (b == 0x00 || is_rst)) && !(ff && rst_next);
            const uint64_t km = __ballot(keep), rm = __ballot(on && is_rst);
            const uint32_t kpos = o + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
            if (keep) out[kpos] = (uint8_t)b;
            if (on && is_rst) {
                const uint32_t k = nseg + __builtin_amdgcn_mbcnt_hi((uint32_t)(rm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rm, 0u));
                if ((b & 7u) != (k & 7u)) status = UCFP_IMAGE_NEEDS_HOST;           // restart markers count modulo 8 (T.81 E.1.4)
                if (k + 1 <= max_seg) sg[k + 1] = kpos;                              // the next segment starts at the next kept byte
            }
            status = __ballot(status != 0) ? UCFP_IMAGE_NEEDS_HOST : 0;
            o += (uint32_t)__popcll(km);
            nseg += (uint32_t)__popcll(rm);
            if (nseg > max_seg) status = UCFP_IMAGE_NEEDS_HOST;
            prev = (uint32_t)__shfl((int)b, 63, 64);
            ended = endm != 0;
        }
        // bit positions are 32-bit (data_bits = clean_len * 8, a segment's (b1 - b0) * 8): a file with 256 MiB or more of
        // entropy-coded data goes to the host's decoder
        if (o >= (1u << 28)) status = UCFP_IMAGE_NEEDS_HOST;
        nseg += 1;                                              // the last segment ends where the data ends
        if (status == 0 && nseg <= max_seg + 1 && lane == 0) sg[nseg] = o;
        if (lane < 16) out[o + lane] = 0;                        // (a reader's look-ahead past the end)
    }
    if (lane == 0) {
        J.status = status;
        J.clean_len = o;
        J.nseg = nseg;
        info[ent] = J;
    }
}

// ---- Huffman decoding ----
// files of ONE segment (no restart markers) are decoded by the speculative kernel below, the others lane-per-segment
__device__ __forceinline__ bool jpeg_takes_spec(const JpgInfo& J);

struct HuffTables {
    uint16_t look[6][512];     // 9 look-ahead bits -> length << 8 | symbol (0: the code is longer)
    int32_t maxcode[6][18];    // canonical form for the long codes (T.81 F.2.2.3); [17] is a sentinel
    int32_t valoff[6][17];     // valptr - mincode
    uint8_t vals[6][256];
    uint16_t qn[64];           // luma quantisation table, natural order
    uint8_t zz[64];            // zigzag -> natural order (per-lane indices: LDS, not the constant cache)
};
struct HuffLds : HuffTables {
    int16_t blk[64][72];       // one coefficient block per lane, natural order (144-byte rows: the lanes' 16-byte reads spread over the banks)
};

// One wave builds a file's code tables in LDS.  Slots: 0..2 = DC of component 0..2, 3..5 = AC (components sharing a table
// build it twice: tiny builds).  Also the zigzag map and the luma quantisation table in natural order (copied to qtab).
__device__ void build_tables(HuffTables& L, const JpgInfo& J, const uint8_t* __restrict__ p, int lane, uint16_t* __restrict__ qtab_img) {
    for (int slot = 0; slot < 2 * (int)J.ncomp; slot++) {
        const int c = slot % J.ncomp, ac = slot / J.ncomp;
        const int sl = ac ? 3 + c : c;
        const uint8_t* t = p + J.dht_off[ac ? (4 | J.ta[c]) : J.td[c]];     // 16 counts, then the symbols
        for (int i = lane; i < 512; i += 64) L.look[sl][i] = 0;
        wave_lds_sync();
        // lane l < 16 owns code length l + 1: first code and first symbol index by a prefix scan over the counts
        const int cnt = lane < 16 ? t[lane] : 0;
        int ksum = cnt;                                 // inclusive symbol count
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const int v = __shfl_up(ksum, off, 64);
            if (lane >= off) ksum += v;
        }
        // mincode[l] = (mincode[l-1] + count[l-1]) << 1, serial over the 16 lengths
        int first = 0;
        {
            int code = 0;
            for (int l = 0; l < 16; l++) {
                const int cl = __shfl(cnt, l, 64);
                if (lane == l) first = code;
                code = (code + cl) << 1;
            }
        }
        const int kfirst = ksum - cnt;
        if (lane < 16) {
            L.maxcode[sl][lane + 1] = cnt ? first + cnt - 1 : -1;
            L.valoff[sl][lane + 1] = kfirst - first;
        }
        if (lane == 16) L.maxcode[sl][17] = 0x7fffffff;
        const int total = __shfl(ksum, 15, 64);
        for (int i = lane; i < total; i += 64) L.vals[sl][i] = t[16 + i];
        wave_lds_sync();
        // look-ahead entries of the codes of <= 9 bits: every code fills 2^(9 - l) entries
        if (lane < 9) {
            const int l = lane + 1;
            for (int j = 0; j < cnt; j++) {
                const int code = first + j, sym = L.vals[sl][kfirst + j];
                const int lo = code << (9 - l), nfill = 1 << (9 - l);
                for (int f = 0; f < nfill; f++) L.look[sl][lo + f] = (uint16_t)(l << 8 | sym);
            }
        }
        wave_lds_sync();
    }
    const uint8_t* q = p + J.dqt_off[J.tq0];
    L.zz[lane] = c_zigzag[lane];
    L.qn[c_zigzag[lane]] = q[lane];
    wave_lds_sync();
    qtab_img[lane] = L.qn[lane];
}

// One wave per file: lane = restart interval (in rounds of 64).
__global__ __launch_bounds__(64) void jpeg_huff_kernel(const uint8_t* __restrict__ jpg, const uint64_t* __restrict__ offsets, size_t n,
                                                      const UpItem* __restrict__ items, UpUniform uni,
                                                      const uint8_t* __restrict__ clean, const uint32_t* __restrict__ seg,
                                                      JpgInfo* __restrict__ info, int16_t* __restrict__ coef,
                                                      uint16_t* __restrict__ qtab) {
    __shared__ HuffLds L;
    const size_t ent = blockIdx.x;
    if (ent >= n) return;
    const int lane = threadIdx.x;
    const JpgInfo J = info[ent];
    if (J.status != 0 || jpeg_takes_spec(J)) return;          // (single-segment files: jpeg_huff_spec_kernel)
    const UpItem item = up_item(items, uni, ent);
    const size_t img = item.file;
    const uint32_t width = item.w, height = item.h, bxp = item.bxp /* luma blocks per row of the plane */;
    const uint8_t* p = jpg + offsets[img];
    build_tables(L, J, p, lane, qtab + ent * 64);
    const uint32_t mcu_w = 8u * J.hmax, mcu_h = 8u * J.vmax;
    const uint32_t mx = (width + mcu_w - 1) / mcu_w, my = (height + mcu_h - 1) / mcu_h, total_mcu = mx * my;
    const uint32_t per = J.restart ? J.restart : total_mcu;
    const uint32_t need = (total_mcu + per - 1) / per;
    bool bad = J.nseg != need;                     // one segment per restart interval, no more, no fewer
    const uint8_t* cl = clean + ((offsets[img] + 15) & ~(uint64_t)15);
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(cl);
    const uint32_t* sg = seg + item.seg_off;
    int16_t* cplane = coef + item.aux_off;
    int16_t* myblk = L.blk[lane];
    for (int i = 0; i < 64; i += 8) *reinterpret_cast<uint4*>(myblk + i) = make_uint4(0, 0, 0, 0);
    const uint32_t nb_c[3] = {(uint32_t)(J.ncomp == 1 ? 1 : J.hs[0] * J.vs[0]), (uint32_t)(J.ncomp == 3 ? J.hs[1] * J.vs[1] : 0),
                              (uint32_t)(J.ncomp == 3 ? J.hs[2] * J.vs[2] : 0)};
    for (uint32_t s0 = 0; s0 < need && !bad; s0 += 64) {
        const uint32_t sgi = s0 + lane;
        const bool mine = sgi < need;
        const uint32_t b0 = mine ? sg[sgi] : 0, b1 = mine ? sg[sgi + 1] : 0;
        const uint32_t avail = (b1 - b0) * 8;
        // bit window: acc holds `cnt` valid bits at its top; wp = the next aligned dword to append; nxt is in flight
        uint32_t wp = b0 >> 2;
        uint64_t acc = 0;
        int cnt = 0;
        uint32_t fed = 0;                           // bits appended so far, counted from the segment's first bit
        {
            const uint32_t w0 = __builtin_bswap32(cw[wp++]);
            const uint32_t skip = (b0 & 3u) * 8;
            acc = (uint64_t)w0 << (32 + skip);
            cnt = 32 - (int)skip;
            fed = (uint32_t)cnt;
        }
        // the dword in flight is kept RAW: swapping its bytes right behind the load would make the wave wait for it there
        // (measured: every refill then cost a full memory round trip -- 45 % of the kernel)
        uint32_t nxt = cw[wp++];
        auto refill = [&]() {                       // keeps cnt >= 32 (a symbol needs <= 16 + 11 bits)
            if (cnt <= 32) {
                acc |= (uint64_t)__builtin_bswap32(nxt) << (32 - cnt);
                cnt += 32;
                fed += 32;
                nxt = cw[wp++];
            }
        };
        auto decode = [&](int sl) -> int {          // one Huffman symbol; -1: no such code
            refill();
            const uint32_t e = L.look[sl][(uint32_t)(acc >> 55)];
            if (e) {
                acc <<= (e >> 8);
                cnt -= (int)(e >> 8);
                return (int)(e & 255u);
            }
            const uint32_t top = (uint32_t)(acc >> 48);
            int l = 10;
            while (l <= 16 && (int)(top >> (16 - l)) > L.maxcode[sl][l]) l++;
            if (l > 16) return -1;
            const int sym = L.vals[sl][(int)(top >> (16 - l)) + L.valoff[sl][l]];
            acc <<= l;
            cnt -= l;
            return sym;
        };
        auto receive = [&](int s) -> int {          // s more bits, sign-extended (T.81 F.2.2.1)
            refill();
            const int v = (int)(acc >> (64 - s));
            acc <<= s;
            cnt -= s;
            return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
        };
        int pred0 = 0, pred1 = 0, pred2 = 0;
        bool err = false;
        if (mine) {
            const uint32_t m1 = (sgi + 1) * per < total_mcu ? (sgi + 1) * per : total_mcu;
            for (uint32_t m = sgi * per; m < m1 && !err; m++) {
                for (int c = 0; c < (int)J.ncomp && !err; c++) {
                    for (uint32_t bi = 0; bi < nb_c[c] && !err; bi++) {
                        const int s = decode(c);
                        if (s < 0 || s > 11) { err = true; break; }
                        const int diff = s ? receive(s) : 0;
                        int dcv;
                        if (c == 0) dcv = (pred0 += diff);
                        else if (c == 1) dcv = (pred1 += diff);
                        else dcv = (pred2 += diff);
                        if (c == 0) myblk[0] = (int16_t)dcv;
                        for (int k = 1; k < 64;) {
                            const int rs = decode(3 + c);
                            if (rs < 0) { err = true; break; }
                            const int r = rs >> 4, sz = rs & 15;
                            if (sz == 0) {
                                if (r != 15) break;
                                k += 16;
                                if (k > 64) err = true;
                                continue;
                            }
                            k += r;
                            if (k > 63 || sz > 10) { err = true; break; }
                            const int v = receive(sz);
                            if (c == 0) myblk[L.zz[k]] = (int16_t)v;
                            k++;
                        }
                        // bits used so far = fed - cnt - (what nxt would add: not yet appended); past the segment: irregular
                        if (fed - (uint32_t)cnt > avail) err = true;
                        if (c == 0 && !err) {
                            const uint32_t bx = (m % mx) * J.hmax + bi % J.hs[0], by = (m / mx) * J.vmax + bi / J.hs[0];
                            uint4* dst = reinterpret_cast<uint4*>(cplane + ((size_t)by * bxp + bx) * 64);
#pragma unroll
                            for (int i = 0; i < 8; i++) {
                                dst[i] = *reinterpret_cast<const uint4*>(myblk + 8 * i);
                                *reinterpret_cast<uint4*>(myblk + 8 * i) = make_uint4(0, 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        if (__ballot(err)) bad = true;
    }
    if (bad && lane == 0) info[ent].status = UCFP_IMAGE_NEEDS_HOST;
}


// ---- files without restart markers: ONE segment, decoded by 64 lanes at once -------------------------------------------
// Huffman decoding is serial in the bit stream; what a wave can add is SPECULATION, as in the PNG front end.  The segment
// is cut into 64 subsequences of S bits.  A decoder's state at a symbol boundary is (bit position, block b of the MCU,
// coefficient index k).  A JPEG stream re-synchronises itself: a parse begun at a wrong bit falls into step with the true
// code boundaries within a few dozen symbols and, at the next end-of-block, with the true coefficient index.  It does NOT
// find the block PHASE b by itself (which of the MCU's B blocks -- hence which Huffman tables -- comes next), and a parse in
// the wrong phase never settles (measured: a chain that only guessed phase 0 needed one round per lane, 3 x slower than the
// serial decode).  So:
//   round 0   lane i parses the TAIL of its subsequence (about kSpecTailBlocks blocks' worth of bits) B times, once per phase h, from (start, h, k = 0):
//             B candidate exit states, one of which is (almost always) the state the true parse leaves the subsequence with;
//   round 1   lane i parses its whole subsequence from EACH candidate exit of lane i - 1;
//   walk      lane 0's entry is known; following the chain -- "my true entry is my predecessor's true exit: look up what I
//             left with when I entered like that" -- is 64 table look-ups; a miss (a true entry nobody guessed) is repaired
//             by one more parse of that lane;
//   decode    every lane decodes its subsequence for real from its true entry: coefficients of luma blocks go straight to
//             the (zeroed) coefficient plane -- a block cut by a subsequence boundary is simply written by two lanes -- with
//             the DC DIFFERENCE in place 0; block numbers come from a prefix sum of the lanes' completed-block counts; a
//             last pass turns the differences into DC values (prefix sum in decoding order).
// About B + B/3 + 1 parses of 1/64 of the stream instead of one of all of it.  The symbol step is written without
// data-dependent branches (DC / AC, run / end-of-block / ZRL are selects): 64 lanes on 64 pieces of the stream would
// otherwise take turns.  The bit stream is read from global memory one dword ahead (no staging buffer), which keeps the
// kernel at 20 KiB of LDS -- eight waves per CU; a first version with a 32 KiB stage and 12-bit tables ran two.
// Where a config-1 file's 3.3 M cycles go (in-kernel cycle stamps, one wave per SIMD): tables 5 %, round 0 35 %, round 1 39 %,
// walk 8 % (one or two repairs), decode 11 %, DC sums < 1 %.  Measured and dropped: 128-entry second-level tables for the
// codes of more than 9 bits instead of the canonical search (+13 % at 1000 files, -13 % at 8000: 26 KiB of LDS are six
// waves per CU); the bit stream fetched eight dwords ahead instead of one (-10 %: the reader does not wait for memory, the
// extra register shuffling costs).
constexpr uint32_t kSpecMaxB = 6;                        // blocks per MCU this path takes (4:2:0 and 4:1:1 have 6)
// What the phase guesses of round 0 parse: the last ~10 blocks' worth of bits of a subsequence (the file's mean bits per block
// x 10, at least 384 bits).  A fixed 768 bits was 10 blocks of a quality-95 4:4:4 file but 16 of a config-1 file: 512 bits
// measured +5 % on 1000 and 8000 config-1 files and -5 % on the quality-95 ones, 384 bits +9 % / +10 % and -14 % (more lanes
// whose true entry nobody guessed: each costs a serial parse).
constexpr uint32_t kSpecTailBlocks = 10, kSpecTailMin = 384;

__device__ __forceinline__ uint32_t jpeg_blocks_per_mcu(const JpgInfo& J) {
    return J.ncomp == 1 ? 1u : (uint32_t)J.hs[0] * J.vs[0] + (uint32_t)J.hs[1] * J.vs[1] + (uint32_t)J.hs[2] * J.vs[2];
}
__device__ __forceinline__ bool jpeg_takes_spec(const JpgInfo& J) {
    return J.nseg == 1 && J.restart == 0 && J.clean_len >= 512 && jpeg_blocks_per_mcu(J) <= kSpecMaxB;
}

struct SpecCand {
    uint32_t pos;
    uint16_t b, k;
    uint32_t done;
};
struct SpecTables : HuffTables {
    uint8_t comp_of[8], ybi_of[8];                       // block of the MCU -> component, index among the luma blocks
};
// NW waves per file = 64 NW subsequences ("lanes" below are the workgroup's threads): 2, 4 or 8 in batches that cannot fill
// the chip with one wave per file, as long as a subsequence keeps >= 256 bits (launch_jpeg_decode): a 1024 x 1024 file is
// 27 ms on one wave whatever the batch, 8.8 on eight.
template <int NW>
struct SpecLds : SpecTables {
    SpecCand c0[64 * NW][kSpecMaxB];                     // round 0: a lane's exit for each guessed block phase
    SpecCand m1[64 * NW][kSpecMaxB];                     // round 1: a lane's exit when entered with its predecessor's j-th exit
    SpecCand tin[64 * NW];                               // the walk: every lane's TRUE entry (done: its blocks completed)
    uint32_t wsum[8];                                    // block scans: the waves' totals
    uint32_t flag;                                       // block-wide "some lane failed"
};

struct SpecState {
    uint32_t pos;       // bit position (from the segment's first bit) of the next symbol
    uint32_t b, k;      // block of the MCU, coefficient index (0: a DC code comes next)
};

// One subsequence: symbols that BEGIN before end_bit, from the state st.  OUT = false: states only (the speculation
// rounds); OUT = true: coefficients of luma blocks are stored as well.
template <bool OUT>
__device__ __forceinline__ void spec_run(const SpecTables& L, const uint32_t* __restrict__ cw, SpecState st, uint32_t end_bit,
                                         uint32_t data_bits, uint32_t B, uint32_t hs0, uint32_t hmax, uint32_t vmax, uint32_t mx,
                                         uint32_t bxp, uint32_t blk_abs, uint32_t blk_total, int16_t* __restrict__ cplane,
                                         SpecState& out, uint32_t& done, bool& err) {
    // bit window: acc holds `cnt` valid bits at its top, wn = the next dword to append, nxt = that dword, in flight (raw:
    // swapping its bytes right behind the load would make the wave wait for it there)
    uint32_t wn = st.pos >> 5;
    uint64_t acc = ((uint64_t)__builtin_bswap32(cw[wn]) << 32 | __builtin_bswap32(cw[wn + 1])) << (st.pos & 31u);
    int cnt = 64 - (int)(st.pos & 31u);
    wn += 2;
    uint32_t nxt = cw[wn];
    uint32_t b = st.b, k = st.k, c = L.comp_of[b];
    done = 0;
    err = false;
    int16_t* dst = nullptr;                 // OUT: the luma block being filled (null: a chroma block)
    auto locate = [&]() {
        if (!OUT) return;
        dst = nullptr;
        if (c == 0) {
            const uint32_t m = blk_abs / B;
            const uint32_t bi = L.ybi_of[b];
            const uint32_t bx = (m % mx) * hmax + bi % hs0, by = (m / mx) * vmax + bi / hs0;
            dst = cplane + ((size_t)by * bxp + bx) * 64;
        }
    };
    locate();
    for (;;) {
        const uint32_t pos = wn * 32u - (uint32_t)cnt;
        if (pos >= end_bit || (OUT && blk_abs >= blk_total)) {
            out.pos = pos;
            break;
        }
        if (cnt <= 32) {                    // afterwards cnt >= 33: a code (<= 16 bits) and its value bits (<= 11) fit
            acc |= (uint64_t)__builtin_bswap32(nxt) << (32 - cnt);
            cnt += 32;
            nxt = cw[++wn];
        }
        const bool isdc = k == 0;
        const uint32_t sl = isdc ? c : 3u + c;
        uint32_t e = L.look[sl][(uint32_t)(acc >> 55)];
        if (e == 0) {                       // a code of more than 9 bits
            const uint32_t top = (uint32_t)(acc >> 48);
            int l = 10;
            while (l <= 16 && (int)(top >> (16 - l)) > L.maxcode[sl][l]) l++;
            if (l > 16) { err = true; break; }
            e = (uint32_t)l << 8 | L.vals[sl][(int)(top >> (16 - l)) + L.valoff[sl][l]];
        }
        const uint32_t nb = e >> 8, sym = e & 255u;
        acc <<= nb;
        const uint32_t sz = isdc ? sym : (sym & 15u), r = isdc ? 0u : sym >> 4;
        int v = 0;
        if (OUT) {
            const int raw = (int)((acc >> 32) >> (32 - sz));          // sz = 0: the shift by 32 of a 32-bit value would be undefined
            v = sz == 0 ? 0 : (raw < (1 << (sz - 1)) ? raw - (1 << sz) + 1 : raw);
        }
        acc <<= sz;
        cnt -= (int)(nb + sz);
        const uint32_t kt = k + r;                                    // where an AC coefficient lands
        const uint32_t kn = isdc ? 1u : sz ? kt + 1u : (r == 15u ? k + 16u : 64u);
        if (kn > 64u || sz > (isdc ? 11u : 10u)) { err = true; break; }
        if (OUT) {
            if (dst && (isdc || sz)) dst[isdc ? 0u : (uint32_t)L.zz[kt]] = (int16_t)v;
            if (wn * 32u - (uint32_t)cnt > data_bits) { err = true; break; }          // the symbol read past the data
        }
        k = kn;
        if (k >= 64u) {
            k = 0;
            b = b + 1 == B ? 0 : b + 1;
            c = L.comp_of[b];
            done++;
            blk_abs++;
            locate();
        }
    }
    out.b = b;
    out.k = k;
    if (err) out.pos = 0xffffffffu;        // (a speculative parse that ran into nonsense: a state no true parse has)
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void jpeg_huff_spec_kernel(const uint8_t* __restrict__ jpg, const uint64_t* __restrict__ offsets, size_t n,
                                                           const UpItem* __restrict__ items, UpUniform uni, const uint8_t* __restrict__ clean,
                                                           JpgInfo* __restrict__ info, int16_t* __restrict__ coef,
                                                           uint16_t* __restrict__ qtab) {
    extern __shared__ __attribute__((aligned(16))) uint8_t spec_lds[];   // (dynamic: 89 KiB at eight waves)
    SpecLds<NW>& L = *reinterpret_cast<SpecLds<NW>*>(spec_lds);
    constexpr int T = 64 * NW;                            // subsequences = threads
    const size_t ent = blockIdx.x;
    if (ent >= n) return;
    const int lane = threadIdx.x;                         // 0 .. T - 1: the subsequence this thread owns
    const int wlane = lane & 63, wave = lane >> 6;
    auto block_sync = [&]() {
        if (NW == 1) wave_lds_sync();
        else __syncthreads();
    };
    // inclusive scan over the workgroup's threads; total = the sum of all
    auto block_scan = [&](uint32_t v, uint32_t& total) -> uint32_t {
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
            if (wlane >= off) incl += o;
        }
        if (NW == 1) {
            total = (uint32_t)__shfl((int)incl, 63, 64);
            return incl;
        }
        if (wlane == 63) L.wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t t = L.wsum[w];
            before += w < wave ? t : 0u;
            all += t;
        }
        __syncthreads();
        total = all;
        return incl + before;
    };
    const JpgInfo J = info[ent];
    if (J.status != 0 || !jpeg_takes_spec(J)) return;
    const UpItem item = up_item(items, uni, ent);
    const size_t img = item.file;
    const uint32_t width = item.w, height = item.h, bxp = item.bxp;
    const size_t coef_stride = (size_t)item.bxp * item.byp * 64;      // the file's coefficient plane, int16
    const uint8_t* p = jpg + offsets[img];
    if (wave == 0) build_tables(L, J, p, wlane, qtab + ent * 64);
    const uint32_t nby = J.ncomp == 1 ? 1u : (uint32_t)J.hs[0] * J.vs[0];
    const uint32_t nb1 = J.ncomp == 3 ? (uint32_t)J.hs[1] * J.vs[1] : 0u;
    const uint32_t B = jpeg_blocks_per_mcu(J);
    if (lane == 0) L.flag = 0;
    if (lane < (int)B) {
        L.comp_of[lane] = (uint32_t)lane < nby ? 0 : (uint32_t)lane < nby + nb1 ? 1 : 2;
        L.ybi_of[lane] = (uint8_t)lane;
    }
    const uint32_t hs0 = J.hs[0], hmax = J.hmax, vmax = J.vmax;
    const uint32_t mx = (width + 8u * hmax - 1) / (8u * hmax), my = (height + 8u * vmax - 1) / (8u * vmax);
    const uint32_t blk_total = mx * my * B;               // (<= 2048 x 2048 MCUs x 6 blocks)
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(clean + ((offsets[img] + 15) & ~(uint64_t)15));
    int16_t* cplane = coef + item.aux_off;
    // the luma blocks' coefficients not written below are zero
    for (size_t i = (size_t)lane * 8; i < coef_stride; i += T * 8) *reinterpret_cast<uint4*>(cplane + i) = make_uint4(0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t data_bits = J.clean_len * 8;
    uint32_t S = (data_bits + T - 1) / T;
    S = (S + 31) & ~31u;
    const uint32_t my0 = (uint32_t)lane * S, my1 = my0 + S < data_bits ? my0 + S : data_bits;
    const bool live = my0 < data_bits;
    const uint32_t last = (data_bits + S - 1) / S - 1;    // the last live lane
    SpecState exit_ = {my1, 0, 0};
    uint32_t done = 0;
    bool err = false;
    auto put = [&](SpecCand& c, const SpecState& e, uint32_t dn) {
        c.pos = e.pos;
        c.b = (uint16_t)e.b;
        c.k = (uint16_t)e.k;
        c.done = dn;
    };
    auto run0 = [&](const SpecState& en) {
        spec_run<false>(L, cw, en, my1, data_bits, B, hs0, hmax, vmax, mx, bxp, 0u, ~0u, nullptr, exit_, done, err);
    };
    block_sync();
    // ---- round 0: every block phase, over the tail of the subsequence (lane 0: its one true entry, the whole of it)
    uint32_t tail_bits = (uint32_t)(((uint64_t)data_bits * kSpecTailBlocks / (blk_total ? blk_total : 1u) + 31u) & ~31ull);
    tail_bits = tail_bits < kSpecTailMin ? kSpecTailMin : tail_bits;
    const uint32_t tail0 = my1 - my0 > tail_bits ? my1 - tail_bits : my0;
    for (uint32_t h = 0; h < B; h++) {
        if (live && (lane > 0 || h == 0)) {
            const SpecState en = {lane == 0 ? 0u : tail0, lane == 0 ? 0u : h, 0u};
            run0(en);
            put(L.c0[lane][h], exit_, done);
        }
    }
    if (lane == 0)
        for (uint32_t h = 1; h < B; h++) L.c0[0][h] = L.c0[0][0];
    block_sync();
    // ---- round 1: the whole subsequence from each candidate exit of the predecessor
    for (uint32_t jx = 0; jx < B; jx++) {
        if (live && lane > 0) {
            const SpecCand pc = L.c0[lane - 1][jx];
            SpecCand res = {0xffffffffu, 0, 0, 0};
            if (pc.pos != 0xffffffffu && (jx == 0 || lane > 1)) {      // (lane 1's candidates are all lane 0's one exit)
                const SpecState en = {pc.pos, pc.b, pc.k};
                run0(en);
                put(res, exit_, done);
            }
            L.m1[lane][jx] = res;
        }
    }
    block_sync();
    // ---- the walk (every lane runs it on the same LDS words: wave-uniform)
    SpecCand t = L.c0[0][0];                                  // lane 0's true exit
    if (lane == 0) L.tin[0] = SpecCand{0u, 0, 0, t.done};
    uint32_t cum = t.done;                                    // blocks completed by the lanes walked so far
    bool bad = t.pos == 0xffffffffu && cum < blk_total;
    for (uint32_t i = 1; i <= last && !bad; i++) {
        if (cum >= blk_total) {
            // the image is complete: what follows (padding bits, stray bytes in front of EOI) is nobody's business -- a
            // sequential decoder stops after the last block too
            if (lane == 0) L.tin[i] = SpecCand{0u, 0, 0, 0u};
            continue;
        }
        // t = true exit of lane i - 1 = true entry of lane i
        int hit = -1;
        for (uint32_t jx = 0; jx < B; jx++) {
            const SpecCand pc = L.c0[i - 1][jx];
            if (hit < 0 && pc.pos == t.pos && pc.b == t.b && pc.k == t.k) hit = (int)jx;
        }
        if (i == 1) hit = 0;
        SpecCand ex;
        if (hit >= 0) {
            ex = L.m1[i][hit];
        } else {
            // nobody guessed lane i's true entry: that lane parses once more
            block_sync();
            if ((uint32_t)lane == i) {
                const SpecState en = {t.pos, t.b, t.k};
                run0(en);
                put(L.m1[i][0], exit_, done);
            }
            block_sync();
            ex = L.m1[i][0];
        }
        if (lane == 0) L.tin[i] = SpecCand{t.pos, t.b, t.k, ex.done};
        cum += ex.done;
        // the TRUE parse of lane i ran into an invalid code before the image was complete: irregular data (J3)
        if (ex.pos == 0xffffffffu && cum < blk_total) bad = true;
        t = ex;                                               // ... and the true entry of lane i + 1
    }
    block_sync();
    uint32_t blk_done = 0;
    if (!bad) {
        const SpecCand mine_in = L.tin[live ? lane : 0];
        const SpecState entry = {mine_in.pos, mine_in.b, mine_in.k};
        // ---- block numbers, then the real decode
        uint32_t total_done;
        const uint32_t incl = block_scan(live ? mine_in.done : 0u, total_done);
        const uint32_t my_abs = incl - (live ? mine_in.done : 0u);
        uint32_t done2 = 0;
        bool err2 = false;
        SpecState ex2 = exit_;
        if (live) spec_run<true>(L, cw, entry, my1, data_bits, B, hs0, hmax, vmax, mx, bxp, my_abs, blk_total, cplane, ex2, done2, err2);
        if (live && err2) L.flag = 1;
        block_sync();
        if (L.flag) bad = true;
        blk_done = total_done;
    }
    if (blk_done < blk_total) bad = true;                 // the data ended before the last block
    if (bad) {
        if (lane == 0) info[ent].status = UCFP_IMAGE_NEEDS_HOST;
        return;
    }
    // ---- DC differences -> DC values: prefix sum over the luma blocks in decoding order (MCU by MCU)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t ny = mx * my * nby, run = (ny + T - 1) / T;
    auto addr_of = [&](uint32_t o) -> int16_t* {
        const uint32_t m = o / nby, bi = o % nby;
        const uint32_t bx = (m % mx) * hmax + bi % hs0, by = (m / mx) * vmax + bi / hs0;
        return cplane + ((size_t)by * bxp + bx) * 64;
    };
    const uint32_t o0 = (uint32_t)lane * run < ny ? (uint32_t)lane * run : ny, o1 = o0 + run < ny ? o0 + run : ny;
    int sum = 0;
    for (uint32_t o = o0; o < o1; o++) sum += addr_of(o)[0];
    uint32_t dc_total;
    int acc_dc = (int)block_scan((uint32_t)sum, dc_total) - sum;      // (two's complement: the wrapping sums are the signed ones)
    for (uint32_t o = o0; o < o1; o++) {
        int16_t* a = addr_of(o);
        acc_dc += a[0];
        a[0] = (int16_t)acc_dc;
    }
}

// ---- inverse DCT: one thread per luma block of the batch ----
#define JD(x, n) (((x) + ((int32_t)1 << ((n)-1))) >> (n))
// in[0..7] -> out[0..7] before descaling (jidctint.c, both passes share this butterfly)
__device__ __forceinline__ void islow_1d(const int32_t in0, const int32_t in1, const int32_t in2, const int32_t in3, const int32_t in4,
                                         const int32_t in5, const int32_t in6, const int32_t in7, int32_t (&o)[8]) {
    int32_t z2 = in2, z3 = in6;
    int32_t z1 = (z2 + z3) * 4433;
    int32_t tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
    int32_t tmp0 = (in0 + in4) * 8192, tmp1 = (in0 - in4) * 8192;
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in7;
    tmp1 = in5;
    tmp2 = in3;
    tmp3 = in1;
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = (z3 + z4) * 9633;
    tmp0 *= 2446;
    tmp1 *= 16819;
    tmp2 *= 25172;
    tmp3 *= 12299;
    z1 *= -7373;
    z2 *= -20995;
    z3 *= -16069;
    z4 *= -3196;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    o[0] = tmp10 + tmp3;
    o[7] = tmp10 - tmp3;
    o[1] = tmp11 + tmp2;
    o[6] = tmp11 - tmp2;
    o[2] = tmp12 + tmp1;
    o[5] = tmp12 - tmp1;
    o[3] = tmp13 + tmp0;
    o[4] = tmp13 - tmp0;
}

// Range guards (DESIGN J4, the same three as the oracle's idct_islow): (a) dequantised coefficients within +-16383, (b) column-pass
// results within +-23000, (c) samples within -512 .. 511 of the centre before the range limit.  Inside them the 32-bit
// butterflies cannot overflow and libjpeg's C table, libjpeg-turbo's 16-bit SIMD form and this code give the same pixels; a
// crafted stream that crosses one makes the file UCFP_IMAGE_NEEDS_HOST (its record is zeroed by the merge kernel).
constexpr int32_t kIdctMaxCoef = 16383, kIdctMaxPass1 = 23000;
// first[e] = the first workgroup of entry e (ceil(blocks of its plane / 256) workgroups each), first[n] = the grid
__global__ __launch_bounds__(256) void jpeg_idct_kernel(JpgInfo* __restrict__ info, size_t n, const UpItem* __restrict__ items, UpUniform uni,
                                                       const uint32_t* __restrict__ first, uint32_t uni_groups,
                                                       const int16_t* __restrict__ coef, const uint16_t* __restrict__ qtab,
                                                       uint8_t* __restrict__ frames) {
    // which entry this workgroup belongs to (wave-uniform)
    size_t img;
    uint32_t g0;
    if (first) {
        size_t lo = 0, hi = n;                       // first[lo] <= blockIdx.x < first[hi]
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) >> 1;
            if (first[mid] <= blockIdx.x) lo = mid;
            else hi = mid;
        }
        img = lo;
        g0 = first[lo];
    } else {
        img = blockIdx.x / uni_groups;
        g0 = (uint32_t)img * uni_groups;
    }
    if (img >= n) return;
    const UpItem item = up_item(items, uni, img);
    const uint32_t width = item.w, height = item.h, bxp = item.bxp, byp = item.byp;
    const size_t row_stride = item.row_stride;
    const uint32_t b = (blockIdx.x - g0) * 256 + threadIdx.x;
    if (b >= bxp * byp) return;
    const uint32_t bx = b % bxp, by = b / bxp;
    if (info[img].status != 0) return;
    if (bx * 8 >= width || by * 8 >= height) return;                 // MCU padding
    const int16_t* c = coef + item.aux_off + (size_t)b * 64;
    const uint16_t* q = qtab + img * 64;
    int32_t ws[64];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint4 cv = *reinterpret_cast<const uint4*>(c + 8 * r);
        const uint4 qv = *reinterpret_cast<const uint4*>(q + 8 * r);
        const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w}, qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            ws[8 * r + 2 * i] = (int32_t)(int16_t)(cw[i] & 0xffffu) * (int32_t)(qw[i] & 0xffffu);
            ws[8 * r + 2 * i + 1] = (int32_t)(int16_t)(cw[i] >> 16) * (int32_t)(qw[i] >> 16);
        }
    }
    {
        int32_t big = 0;
#pragma unroll
        for (int i = 0; i < 64; i++) big = max(big, ws[i] < 0 ? -ws[i] : ws[i]);
        if (big > kIdctMaxCoef) {
            info[img].status = UCFP_IMAGE_NEEDS_HOST;
            return;
        }
    }
    // pass 1: columns
#pragma unroll
    for (int col = 0; col < 8; col++) {
        int32_t o[8];
        islow_1d(ws[col], ws[8 + col], ws[16 + col], ws[24 + col], ws[32 + col], ws[40 + col], ws[48 + col], ws[56 + col], o);
#pragma unroll
        for (int r = 0; r < 8; r++) ws[8 * r + col] = JD(o[r], 13 - 2);
    }
    {
        int32_t big = 0;
#pragma unroll
        for (int i = 0; i < 64; i++) big = max(big, ws[i] < 0 ? -ws[i] : ws[i]);
        if (big > kIdctMaxPass1) {
            info[img].status = UCFP_IMAGE_NEEDS_HOST;
            return;
        }
    }
    bool wild = false;
    // pass 2: rows, range limit, store
    uint8_t* dst = frames + item.frame_off + (size_t)(by * 8) * row_stride + (size_t)bx * 8;
    const bool whole = bx * 8 + 8 <= width && ((reinterpret_cast<uintptr_t>(dst) | row_stride) & 7u) == 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int32_t o[8];
        islow_1d(ws[8 * r], ws[8 * r + 1], ws[8 * r + 2], ws[8 * r + 3], ws[8 * r + 4], ws[8 * r + 5], ws[8 * r + 6], ws[8 * r + 7], o);
        uint32_t px[8];
#pragma unroll
        for (int x = 0; x < 8; x++) {
            const int32_t v = JD(o[x], 13 + 2 + 3) + 128;
            wild = wild || v < -512 + 128 || v > 511 + 128;
            px[x] = (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
        if (by * 8 + r < height) {
            uint8_t* d = dst + (size_t)r * row_stride;
            if (whole) {
                *reinterpret_cast<uint2*>(d) = make_uint2(px[0] | px[1] << 8 | px[2] << 16 | px[3] << 24,
                                                          px[4] | px[5] << 8 | px[6] << 16 | px[7] << 24);
            } else {
#pragma unroll
                for (int x = 0; x < 8; x++)
                    if (bx * 8 + x < width) d[x] = (uint8_t)px[x];
            }
        }
    }
    if (wild) info[img].status = UCFP_IMAGE_NEEDS_HOST;
}

__global__ void jpeg_status_kernel(const JpgInfo* __restrict__ info, const UpItem* __restrict__ items, size_t n,
                                   int32_t* __restrict__ status) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) status[items ? items[i].file : i] = info[i].status;
}

// Records of files that did not decode are zeroed and carry the decoder's status.
__global__ void jpeg_merge_status_kernel(const JpgInfo* __restrict__ info, const UpItem* __restrict__ items, size_t n,
                                         uint8_t* __restrict__ out, uint32_t rec, int32_t* __restrict__ status) {
    const size_t e = (size_t)blockIdx.x * blockDim.x / 64 + threadIdx.x / 64;
    if (e >= n) return;
    const int32_t st = info[e].status;
    if (st == 0) return;
    const size_t i = items ? items[e].file : e;
    const int lane = threadIdx.x & 63;
    for (uint32_t b = lane * 4; b < rec; b += 256) *reinterpret_cast<uint32_t*>(out + i * rec + b) = 0;
    if (lane == 0 && status) status[i] = st;
}

}  // namespace

// the luma plane of a w x h file in blocks, with room for the padding of the largest MCU (up to 3 more blocks each way),
// and the most restart segments its table may have to hold (restart interval = 1 MCU of one block)
void jpeg_plane_geometry(uint32_t w, uint32_t h, uint32_t* bxp, uint32_t* byp, uint32_t* max_seg) {
    *bxp = (w + 7) / 8 + 4;
    *byp = (h + 7) / 8 + 4;
    uint32_t ms = ((w + 7) / 8) * ((h + 7) / 8);
    if (ms > (uint32_t)kMaxSeg) ms = (uint32_t)kMaxSeg;
    *max_seg = ms;
}

// seg_words / coef_words: the batch's totals (uniform: n x per file)
static size_t jpeg_ws_layout(size_t n, size_t jpg_bytes, size_t seg_words, size_t coef_words, JpegWs* l) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    l->jpg_bytes = jpg_bytes;
    l->clean = take(jpg_bytes + 16 + 64 + 512);
    l->info = take(n * sizeof(JpgInfo));
    l->seg = take(seg_words * 4);
    l->qtab = take(n * 64 * 2);
    l->coef = take(coef_words * 2);
    l->total = off;
    return off;
}

size_t jpeg_ws_bytes(size_t n, size_t jpg_bytes, uint32_t w, uint32_t h, JpegWs* ws) {
    JpegWs l;
    jpeg_plane_geometry(w, h, &l.bxp, &l.byp, &l.max_seg);
    l.coef_stride = (size_t)l.bxp * l.byp * 64;
    const size_t off = jpeg_ws_layout(n, jpg_bytes, n * (size_t)(l.max_seg + 2), n * l.coef_stride, &l);
    if (ws) *ws = l;
    return off;
}

size_t jpeg_ragged_ws_bytes(size_t n, size_t jpg_bytes, size_t seg_words, size_t coef_words, JpegWs* ws) {
    JpegWs l;
    const size_t off = jpeg_ws_layout(n, jpg_bytes, seg_words, coef_words, &l);
    if (ws) *ws = l;
    return off;
}

// idct_groups: workgroups of the inverse DCT launch (ragged: d_first[n]; uniform: n x uni_groups)
static int jpeg_decode_launches(const uint8_t* jpg, const uint64_t* offsets, const UpItem* d_items, const UpUniform& uni, size_t n,
                                const uint32_t* d_first, uint32_t uni_groups, size_t idct_groups, uint8_t* ws, const JpegWs& l,
                                uint8_t* frames, int32_t* status, hipStream_t stream) {
    JpgInfo* info = reinterpret_cast<JpgInfo*>(ws + l.info);
    uint32_t* seg = reinterpret_cast<uint32_t*>(ws + l.seg);
    int16_t* coef = reinterpret_cast<int16_t*>(ws + l.coef);
    uint16_t* qtab = reinterpret_cast<uint16_t*>(ws + l.qtab);
    hipLaunchKernelGGL(jpeg_scan_kernel, dim3((unsigned)n), dim3(64), 0, stream, jpg, offsets, n, d_items, uni, ws + l.clean, seg, info);
    hipLaunchKernelGGL(jpeg_huff_kernel, dim3((unsigned)n), dim3(64), 0, stream, jpg, offsets, n, d_items, uni,
                       (const uint8_t*)(ws + l.clean), (const uint32_t*)seg, info, coef, qtab);
    {
        // waves per file of the speculative decoder: as many as keep the batch within the chip's ~2048 wave slots and a
        // subsequence at >= 256 bits (measured on 9 KB config-1 files, images/s at 1 | 2 | 4 waves: 1000 files 458 k | 556 k |
        // 438 k, 500 files 252 k | 310 k | 387 k, 250 files 132 k | 172 k | 212 k)
        const size_t mean_bits = l.jpg_bytes / n * 8;
        int nw = 1;
        while (nw < 8 && n * (size_t)nw * 2 <= 2048 && mean_bits / (64 * (size_t)nw * 2) >= 256) nw *= 2;
        auto go = [&](auto kern, int threads, size_t lds) {
            if (lds > 48 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(threads), lds, stream, jpg, offsets, n, d_items, uni,
                               (const uint8_t*)(ws + l.clean), info, coef, qtab);
        };
        if (nw == 1) go(jpeg_huff_spec_kernel<1>, 64, sizeof(SpecLds<1>));
        else if (nw == 2) go(jpeg_huff_spec_kernel<2>, 128, sizeof(SpecLds<2>));
        else if (nw == 4) go(jpeg_huff_spec_kernel<4>, 256, sizeof(SpecLds<4>));
        else go(jpeg_huff_spec_kernel<8>, 512, sizeof(SpecLds<8>));
    }
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)idct_groups), dim3(256), 0, stream, info, n, d_items, uni, d_first, uni_groups,
                       (const int16_t*)coef, (const uint16_t*)qtab, frames);
    if (status)
        hipLaunchKernelGGL(jpeg_status_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const JpgInfo*)info, d_items, n,
                           status);
    return 0;
}

int launch_jpeg_decode(const uint8_t* jpg, const uint64_t* offsets, size_t n, uint32_t w, uint32_t h, uint8_t* ws,
                       const JpegWs& l, uint8_t* frames, size_t row_stride, size_t frame_stride, int32_t* status,
                       hipStream_t stream) {
    if (n == 0) return 0;
    UpUniform uni;
    uni.w = w;
    uni.h = h;
    uni.pixfmt = UCFP_PIX_GRAY8;
    uni.row_stride = (uint32_t)row_stride;
    uni.bxp = l.bxp;
    uni.byp = l.byp;
    uni.max_seg = l.max_seg;
    uni.frame_stride = frame_stride;
    uni.aux_stride = l.coef_stride;
    const uint32_t groups = (uint32_t)(((size_t)l.bxp * l.byp + 255) / 256);
    return jpeg_decode_launches(jpg, offsets, nullptr, uni, n, nullptr, groups, n * (size_t)groups, ws, l, frames, status, stream);
}

// d_first: n + 1 device words (host table): first workgroup of every entry's inverse DCT, d_first[n] = idct_groups
int launch_jpeg_decode_ragged(const uint8_t* jpg, const uint64_t* offsets, const UpItem* d_items, size_t n, const uint32_t* d_first,
                              size_t idct_groups, uint8_t* ws, const JpegWs& l, uint8_t* frames, int32_t* status, hipStream_t stream) {
    if (n == 0) return 0;
    return jpeg_decode_launches(jpg, offsets, d_items, UpUniform{}, n, d_first, 0, idct_groups, ws, l, frames, status, stream);
}

int launch_jpeg_merge_status(const uint8_t* ws, const JpegWs& l, size_t n, uint8_t* out, uint32_t rec, int32_t* status,
                             hipStream_t stream, const UpItem* d_items) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(jpeg_merge_status_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream,
                       reinterpret_cast<const JpgInfo*>(ws + l.info), d_items, n, out, rec, status);
    return 0;
}

}  // namespace ucfp
