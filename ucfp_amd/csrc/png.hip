// png.hip -- PNG front end on the GPU: compressed uploads in, decoded frames out, so that BASELINE
// config 1 ("?algorithm=phash on 1 k 256x256 PNGs") no longer leaves the GPU waiting on a host decoder.
//
// The reference decodes inside the SDK call (src/modality/image.rs:68-70, :176-179: imgfprint -> image::load_from_memory
// -> png crate).  Here a batch of PNG files takes three stages (round 4: the inflate stage is two kernels, png_huff_kernel with
// 1-4 waves per file and png_lz_kernel -- see "two-pass inflate" below; png_inflate_kernel is the one-kernel form it replaced,
// kept behind UCFP_PNG_TWO_PASS=0 for A/B and for tools/prof_png_phases.py):
//
//   png_scan      walks the chunks (PNG 5.3), checks IHDR against the geometry the batch was announced with, gathers
//                 the IDAT payloads into one contiguous zlib stream.
//   png_inflate   RFC 1950/1951.  Huffman decoding is serial in the bit stream, so the wave SPECULATES: the next
//                 64 x B bits are cut into 64 subsequences, lane j decodes from a guessed bit offset j*B.  Huffman
//                 codes self-synchronise: after a few symbols a wrong parse falls into step with the right one, so
//                 lane j's exit position is usually right even though its start was not.  Lane j+1 restarts from lane
//                 j's exit until the chain of (start == predecessor's exit) reaches from lane 0 (whose start is known)
//                 to the end -- by construction the accepted parse IS the sequential one.  A lane remembers the parses
//                 it has made (the candidate starts are few), so later rounds are lookups.  Then every lane decodes its
//                 subsequence once more into the round buffer (literals directly; matches are listed and resolved in
//                 stream order, 64 at a time where they do not depend on each other).
//   png_unfilter  PNG 9.2 filters.  Average and Paeth chain along the row AND need the row above: lane j takes row
//                 64 k + j one pixel behind lane j-1, so "above" and "above-left" are the neighbour lane's last two
//                 outputs (a diagonal wavefront, 64 rows in flight).
//
// Scope: 8-bit greyscale / grey + alpha / RGB / indexed colour / RGBA, non-interlaced, with or without a tRNS chunk (it only adds
// an alpha channel, and luma takes no alpha) -- anything else gets UCFP_IMAGE_NEEDS_HOST and goes to
// the host's decoder (like non-ASCII text).  Every chunk's CRC-32 and the Adler-32 trailer are computed; a bad CRC on a
// critical chunk is UCFP_E_MODALITY, while the checksum-ONLY failures decoders disagree on (Adler-32 of a stream that
// otherwise inflated to the right length, the CRC of an ancillary chunk) are UCFP_IMAGE_NEEDS_HOST: the host's decoder
// -- the reference's own, in a drop-in -- decides, so the device never rejects a file the reference would fingerprint.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ucfp_hip.h"
#include "common.h"

namespace ucfp {

namespace {

constexpr int kRoot = 10, kDRoot = 8;                  // bits indexed by the first-level tables
// Three shapes of a speculation round.  Wide rounds (256-bit subsequences) resynchronise more often inside a subsequence
// -- 3.3 instead of 4.3 parses per subsequence on the match-heavy config-1 files -- and halve the per-round overhead,
// but their buffers take 30 KiB of LDS per wave instead of 17.5: right for a batch that cannot fill the chip anyway
// (1000 files: +19 %), wrong for a large one (8000 files: -20 %).  launch_png_decode picks by batch size.
template <int BITS>
struct RoundCfg {
    static constexpr int kMaxB = BITS;                                     // bits per subsequence
    static constexpr uint32_t kIterOut = BITS * 32;                        // output bytes one round may add
    static constexpr uint32_t kMatchCap = BITS * 8;                        // matches listed per round
    static constexpr int kStageWords = BITS * 2 + 32;                      // 64 subsequences + overshoot + the fetch window
};
#ifndef UCFP_PNG_LZ_WINDOW
#define UCFP_PNG_LZ_WINDOW 2048
#endif
constexpr uint32_t kHist = 2048;                       // bytes of earlier rounds kept in LDS: a match one image row back (the common distance) never leaves the CU

// table entries: value (literal / length base / distance base) | extra bits << 16 | code length << 20 | kind << 24
constexpr uint32_t kLit = 0, kLen = 1, kEob = 2, kSlow = 3;
constexpr uint32_t kInvalid = kSlow << 24;             // code length 0: resolved (or rejected) by the canonical slow path

// The deflate window is the image itself: the round in flight and the last kHist bytes before it live in LDS, older
// bytes are read back from frame memory (this wave wrote them; a workgroup-scope fence orders the stores before the
// loads).  17.5 KiB per wave in the narrow shape instead of the 48 KiB a 32 KiB window in LDS would take.
template <class C>
struct InflateLds {
    uint32_t lit[1 << kRoot];
    uint32_t dst[1 << kDRoot];
    uint32_t stage[C::kStageWords];
    uint32_t m_dst[C::kMatchCap];
    uint32_t m_ld[C::kMatchCap];                          // len << 16 | (dist - 1)
    uint16_t ll_sorted[288];
    uint16_t d_sorted[32];
    uint16_t ll_count[16];
    uint16_t d_count[16];
    uint8_t lens[320];
    alignas(4) uint8_t rb[kHist + C::kIterOut + 8];       // the last kHist bytes of earlier rounds + this round's output; rb[0] is stream position rb_base
};

__device__ __forceinline__ uint32_t lit_entry(uint32_t sym, uint32_t len) {
    if (sym < 256) return sym | len << 20 | kLit << 24;
    if (sym == 256) return len << 20 | kEob << 24;
    const uint32_t c = sym - 257;
    if (c >= 29) return kInvalid;
    uint32_t ex, base;
    if (c < 8) ex = 0, base = 3 + c;
    else if (c == 28) ex = 0, base = 258;
    else ex = (c - 4) >> 2, base = 3 + ((4 + (c & 3)) << ex);
    return base | ex << 16 | len << 20 | kLen << 24;
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t len) {
    if (sym >= 30) return kInvalid;
    uint32_t ex, base;
    if (sym < 4) ex = 0, base = 1 + sym;
    else ex = (sym - 2) >> 1, base = 1 + ((2 + (sym & 1)) << ex);
    return base | ex << 16 | len << 20 | kLit << 24;
}

// Canonical decode, one bit at a time (codes longer than the root, and the verdict on invalid prefixes).
__device__ uint32_t slow_code(uint64_t buf, const uint16_t* count, const uint16_t* sorted, bool dist) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; l++) {
        code |= (int)(buf & 1);
        buf >>= 1;
        const int cnt = count[l];
        if (code - cnt < first) {
            const uint32_t sym = sorted[index + (code - first)];
            return dist ? dist_entry(sym, l) : lit_entry(sym, l);
        }
        index += cnt;
        first += cnt;
        first <<= 1;
        code <<= 1;
    }
    return kInvalid;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Builds one code from lens[0 .. n): first-level table, canonical arrays for the slow path.  Lane j owns symbols
// [5 j, 5 j + 5).  Returns 0 complete, 1 incomplete, -1 over-subscribed; *used = symbols with a code.
template <bool DIST>
__device__ int build_table(const uint8_t* lens, int n, uint32_t* tab, uint16_t* count, uint16_t* sorted, int lane, int* used) {
    constexpr int R = DIST ? kDRoot : kRoot;
    for (int i = lane; i < (1 << R); i += 64) tab[i] = kInvalid;
    // per-lane histogram of the code lengths 1 .. 15, three 10-bit counters per word
    uint32_t mine[5] = {0, 0, 0, 0, 0};
    uint8_t l5[5];
#pragma unroll
    for (int t = 0; t < 5; t++) {
        const int s = lane * 5 + t;
        const uint32_t l = s < n ? lens[s] : 0;
        l5[t] = (uint8_t)l;
        if (l) {
            const uint32_t w = (l - 1) / 3, sh = ((l - 1) % 3) * 10;
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (w == (uint32_t)k) mine[k] += 1u << sh;
        }
    }
    uint32_t run[5], tot[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t inc = wave_incl_scan(mine[k], lane);
        run[k] = inc - mine[k];
        tot[k] = __shfl(inc, 63, 64);
    }
    int left = 1, code = 0, off = 0, total = 0;
    uint32_t next[16], offs[16];
    next[0] = offs[0] = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) {
        const uint32_t c = (tot[(l - 1) / 3] >> (((l - 1) % 3) * 10)) & 1023u;
        left = (left << 1) - (int)c;
        next[l] = (uint32_t)code;
        offs[l] = (uint32_t)off;
        code = (code + (int)c) << 1;
        off += (int)c;
        total += (int)c;
        if (lane == 0) count[l] = (uint16_t)c;
    }
    if (lane == 0) count[0] = 0;
    *used = total;
    if (left < 0) return -1;
#pragma unroll
    for (int t = 0; t < 5; t++) {
        const uint32_t l = l5[t];
        if (!l) continue;
        const uint32_t s = lane * 5 + t;
        const uint32_t w = (l - 1) / 3, sh = ((l - 1) % 3) * 10;
        uint32_t rank = 0, nx = 0, of = 0;
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (w == (uint32_t)k) {
                rank = (run[k] >> sh) & 1023u;
                run[k] += 1u << sh;
            }
#pragma unroll
        for (int q = 1; q <= 15; q++)
            if (l == (uint32_t)q) nx = next[q], of = offs[q];
        sorted[of + rank] = (uint16_t)s;
        const uint32_t cd = nx + rank;
        const uint32_t rev = __brev(cd) >> (32 - l);          // codes are packed starting from their most significant bit
        if (l <= (uint32_t)R) {
            const uint32_t e = DIST ? dist_entry(s, l) : lit_entry(s, l);
            for (uint32_t i = rev; i < (1u << R); i += 1u << l) tab[i] = e;
        }                                                     // longer codes: the first-level entry stays "slow"
    }
    return left > 0 ? 1 : 0;
}

// 32 bits of the staged stream from bit `pos` on: two adjacent words and one funnel shift.  A lane's only decoder state
// is its bit position -- every VALU instruction of a 64-wide wave costs four cycles, so the symbol loop is written for
// instruction count: no bit buffer to maintain, the bits are fetched again where they are needed.
__device__ __forceinline__ uint32_t bits32(const uint32_t* stage, uint32_t pos) {
    const uint32_t w = pos >> 5;
    return __builtin_amdgcn_alignbit(stage[w + 1], stage[w], pos & 31);
}

// Canonical decode for the rare code longer than the first-level table (and the verdict on invalid prefixes).
__device__ __noinline__ uint32_t slow_code32(uint32_t x, const uint16_t* count, const uint16_t* sorted, bool dist) {
    return slow_code(x, count, sorted, dist);
}

// A lane's parse of its subsequence: the symbols that START in [start, limit).
struct Parse {
    uint32_t start, exit;     // exit: bit position of the first symbol this lane did NOT decode
    uint32_t packed;          // bytes produced (17 bits) | matches << 17 (13 bits) | end-of-block << 30 | invalid code << 31
    __device__ __forceinline__ uint32_t nbytes() const { return packed & 0x1ffffu; }
    __device__ __forceinline__ uint32_t nmatch() const { return (packed >> 17) & 0x1fffu; }
    __device__ __forceinline__ bool eob() const { return (packed >> 30) & 1u; }
    __device__ __forceinline__ bool err() const { return packed >> 31; }
    __device__ __forceinline__ bool stopped() const { return packed >> 30; }
};

// EMIT = false: count.  EMIT = true: literals into the round buffer (rb_base = stream position of L.rb[0]), matches onto
// the list; the err flag of the result then also reports a match that reaches back before the first byte of the image.
// Written straight-line: the loop is uniform (it runs while any lane still has symbols), a lane that is done idles on
// its start position, and literal / length / distance handling are selects, not branches -- nested divergent regions
// cost more in exec-mask bookkeeping and serial LDS waits than the work they skip.  One 64-bit window of the stream per
// symbol (three words, one LDS round trip) serves both the literal/length code and the distance code behind it.
template <bool EMIT, class C>
__device__ __forceinline__ Parse parse_sub(InflateLds<C>& L, uint32_t start, uint32_t limit, uint32_t out_pos, uint32_t rb_base,
                                           uint32_t m_idx) {
    uint32_t pos = start, nb = 0, nm = 0, flags = 0;
    bool act = pos < limit;
    while (__ballot(act)) {
        const uint32_t p = act ? pos : start;
        const uint32_t w = p >> 5, sh = p & 31;
        const uint32_t w0 = L.stage[w], w1 = L.stage[w + 1], w2 = L.stage[w + 2];
        const uint32_t x = __builtin_amdgcn_alignbit(w1, w0, sh), x2 = __builtin_amdgcn_alignbit(w2, w1, sh);
        uint32_t e = L.lit[x & ((1u << kRoot) - 1)];
        if (__ballot(act && (e >> 24) == kSlow)) {
            if ((e >> 24) == kSlow) e = slow_code32(x, L.ll_count, L.ll_sorted, false);
        }
        const uint32_t cl = (e >> 20) & 15u, kind = e >> 24, ex = (e >> 16) & 15u;
        const uint32_t s1 = cl + ex;                                          // <= 20 bits: code + extra bits (0 for literals)
        const bool is_len = kind == kLen;
        const uint32_t len = (e & 0xffffu) + ((x >> cl) & ((1u << ex) - 1u));
        uint32_t adv = s1, dist = 0;
        bool bad = cl == 0;
        if (__ballot(act && is_len)) {
            const uint32_t y = __builtin_amdgcn_alignbit(x2, x, s1);         // the 32 bits behind the length: s1 + 28 <= 48 < 64
            uint32_t d = L.dst[y & ((1u << kDRoot) - 1)];
            if (__ballot(act && is_len && (d >> 24) == kSlow)) {
                if (is_len && (d >> 24) == kSlow) d = slow_code32(y, L.d_count, L.d_sorted, true);
            }
            const uint32_t dl = (d >> 20) & 15u, dex = (d >> 16) & 15u;
            dist = (d & 0xffffu) + ((y >> dl) & ((1u << dex) - 1u));
            adv = is_len ? s1 + dl + dex : adv;
            bad = bad || (is_len && dl == 0);
        }
        const bool lit = kind == kLit, eob = kind == kEob;
        const bool go = act && !bad;
        if (EMIT) {
            if (go && lit) L.rb[out_pos + nb - rb_base] = (uint8_t)e;
            if (go && is_len) {
                if (dist > out_pos + nb) flags |= 2u;
                const uint32_t mi = m_idx + nm;
                if (mi < C::kMatchCap) {
                    L.m_dst[mi] = out_pos + nb;
                    L.m_ld[mi] = len << 16 | (dist - 1);
                }
            }
        }
        flags |= (act && bad) ? 2u : 0u;
        flags |= (go && eob) ? 1u : 0u;
        pos += go ? adv : 0u;
        nb += (go && lit) ? 1u : (go && is_len) ? len : 0u;
        nm += (go && is_len) ? 1u : 0u;
        act = go && !eob && pos < limit;
    }
    return Parse{start, pos, nb | nm << 17 | flags << 30};     // a 512-bit subsequence holds < 256 matches, < 66 KiB of output
}

// The parses a lane has already made in this round, by start position: the candidates for a lane's start are few (wrong
// parses fall into step with one another too), so after two or three passes the chain resolves by lookup alone.
struct ParseCache {
    static constexpr int kWays = 4;
    Parse way[kWays];
    int next;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < kWays; i++) way[i] = Parse{0xffffffffu, 0, 0};
        next = 0;
    }
    __device__ __forceinline__ bool find(uint32_t start, Parse* out) const {
        bool hit = false;
#pragma unroll
        for (int i = 0; i < kWays; i++)
            if (way[i].start == start) {
                *out = way[i];
                hit = true;
            }
        return hit;
    }
    __device__ __forceinline__ void put(const Parse& p) {
#pragma unroll
        for (int i = 0; i < kWays; i++)
            if (next == i) way[i] = p;
        next = (next + 1) & (kWays - 1);
    }
};

// Uniform (whole-wave) bit reader over the stage for block headers: every lane computes the same values.
__device__ __forceinline__ uint32_t peek_u(const uint32_t* stage, uint32_t pos, uint32_t n) {
    const uint32_t w = pos >> 5;
    const uint64_t two = (uint64_t)stage[w] | (uint64_t)stage[w + 1] << 32;
    return (uint32_t)(two >> (pos & 31)) & ((1u << n) - 1u);
}

// Stages kStageWords words of the stream starting at the word that holds bit `bp`; returns that word's bit offset.
template <class C>
__device__ __forceinline__ uint32_t stage_load(InflateLds<C>& L, const uint8_t* z, uint32_t zwords, uint32_t bp, int lane) {
    const uint32_t w0 = bp >> 5;
    const uint32_t* zw = reinterpret_cast<const uint32_t*>(z);
    for (int i = lane; i < C::kStageWords; i += 64) L.stage[i] = (w0 + i < zwords) ? zw[w0 + i] : 0u;
    wave_lds_sync();
    return w0 << 5;
}

// Round buffer -> frame memory for [from, to); rb[0] is position `base` (a multiple of 4), so whole words line up.
template <class LT>
__device__ __forceinline__ void flush_out(const LT& L, uint8_t* out, uint32_t from, uint32_t to, uint32_t base, int lane) {
    uint32_t p = from;
    const uint32_t head = (4 - (p & 3)) & 3;
    if (lane < (int)head && p + lane < to) out[p + lane] = L.rb[p + lane - base];
    p += head;
    if (p >= to) return;
    const uint32_t words = (to - p) >> 2;
    for (uint32_t i = lane; i < words; i += 64)
        *reinterpret_cast<uint32_t*>(out + p + 4 * i) = *reinterpret_cast<const uint32_t*>(&L.rb[p + 4 * i - base]);
    p += words * 4;
    if (lane < (int)(to - p)) out[p + lane] = L.rb[p + lane - base];
}

// A byte of the stream: from the round buffer (this round and the kHist bytes before it), or from frame memory.
template <class LT>
__device__ __forceinline__ uint8_t window_byte(const LT& L, const uint8_t* out, uint32_t p, uint32_t rb_base) {
    return p >= rb_base ? L.rb[p - rb_base] : out[p];
}

// Resolves the listed matches in stream order.  All literals of the round are already in the round buffer.
//
// A group of 64 entries (one per lane) is resolved in steps: every entry whose source ends at or below the first unresolved
// destination moves together.  What the steps cost was measured piece by piece (round 4, png_lz_kernel, a level-1 photograph:
// 950 three-byte matches per round, 38 steps): a source older than the LDS window is a round trip to frame memory for the whole
// wave, and nearly every step had one; sixteen byte-wide LDS instructions at random addresses are ~6-way bank conflicts each.
// So: (1) sources below the window never depend on a pending match -- they are fetched for the WHOLE group when it is loaded,
// one group ahead of the one being stepped through, as aligned words; (2) window sources come as two or three aligned words
// shifted into place; (3) a step whose ready matches are all <= 4 bytes issues four byte writes; (4) places a lane does not
// own are written to a spare byte behind the window, so the moves are straight-line.
template <class LT>
__device__ void resolve_matches(LT& L, const uint8_t* out, uint32_t rb_base, uint32_t total, int lane,
                                unsigned long long* rounds = nullptr, unsigned long long* coop = nullptr) {
    constexpr uint32_t kSpare = sizeof(L.rb) - 4;
    const uint32_t* rw = reinterpret_cast<const uint32_t*>(L.rb);
    struct Group {
        uint32_t dst, len, dist;
        uint32_t g[3];         // a far source's bytes [srcp & ~3, + 12), requested when the group was loaded
        bool have, far;
    };
    auto load_group = [&](uint32_t g0) {
        Group G;
        const uint32_t mi = g0 + lane;
        G.have = mi < total;
        const uint32_t d = G.have ? L.m_dst[mi] : 0, ld = G.have ? L.m_ld[mi] : 0;
        G.dst = d;
        G.len = ld >> 16;
        G.dist = (ld & 0xffffu) + 1;
        const uint32_t srcp = d - G.dist;
        G.far = G.have && G.len <= 8 && G.dist >= G.len && srcp < rb_base;     // (short entries only: the others go byte by byte)
        G.g[0] = G.g[1] = G.g[2] = 0;
        if (G.far) {
            // whole words around the source: it ends below the round, inside the file's own area (whose first byte is 16-aligned)
            const uint32_t* gw = reinterpret_cast<const uint32_t*>(out + (srcp & ~3u));
            G.g[0] = gw[0];
            G.g[1] = gw[1];
            G.g[2] = gw[2];
        }
        return G;
    };
    Group N = load_group(0);
    for (uint32_t g0 = 0; g0 < total; g0 += 64) {
        const Group G = N;
        if (g0 + 64 < total) N = load_group(g0 + 64);
        const uint32_t dst = G.dst, len = G.len, dist = G.dist;
        const uint32_t srcp = dst - dist;
        uint64_t pending = __ballot(G.have);
        while (pending) {
            const int f = __builtin_ctzll(pending);
            const uint32_t f_dst = __builtin_amdgcn_readlane(dst, f), f_len = __builtin_amdgcn_readlane(len, f),
                           f_dist = __builtin_amdgcn_readlane(dist, f);
            if (rounds) (*rounds)++;
            if (f_len > 8 || f_dist < f_len) {
                if (coop) (*coop)++;
                // a long or self-overlapping match: the whole wave copies it (byte k comes from k mod dist)
                for (uint32_t k0 = 0; k0 < f_len; k0 += 64) {
                    const uint32_t k = k0 + lane;
                    uint8_t v = 0;
                    if (k < f_len) v = window_byte(L, out, f_dst - f_dist + (k % f_dist), rb_base);
                    if (k < f_len) L.rb[f_dst + k - rb_base] = v;
                }
                wave_lds_fence();      // (the DS unit runs one wave's instructions in order: the next step's reads see these writes without a drain)
                pending &= ~(1ull << f);
                continue;
            }
            // short matches whose source lies entirely below the first unresolved destination: one per lane, together
            const bool ready = ((pending >> lane) & 1) && len <= 8 && dist >= len && srcp + len <= f_dst;
            const uint32_t sa = (ready && !G.far) ? srcp - rb_base : 0u, da = ready ? dst - rb_base : 0u, ln = ready ? len : 0u;
            const uint32_t sh = srcp & 3u;          // (rb_base is a multiple of 4: the same shift in the window and in frame memory)
            const bool need8 = __ballot(ln > 4u) != 0;
            uint32_t w0 = rw[sa >> 2], w1 = rw[(sa >> 2) + 1], w2 = need8 ? rw[(sa >> 2) + 2] : 0u;
            w0 = G.far ? G.g[0] : w0;
            w1 = G.far ? G.g[1] : w1;
            w2 = G.far ? G.g[2] : w2;
            const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
#pragma unroll
            for (int k = 0; k < 4; k++) L.rb[(uint32_t)k < ln ? da + k : kSpare] = (uint8_t)(lo >> (8 * k));
            if (need8) {
                const uint32_t hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
#pragma unroll
                for (int k = 4; k < 8; k++) L.rb[(uint32_t)k < ln ? da + k : kSpare] = (uint8_t)(hi >> (8 * (k - 4)));
            }
            wave_lds_fence();
            pending &= ~__ballot(ready);
        }
    }
}

// Adler-32 (RFC 1950) over the stream as it is produced: (a, b) after m more bytes d_0 .. d_{m-1} is
// a + S1, b + m a + S2 with S1 = sum d_j, S2 = sum (m - j) d_j; the lanes take every 64th byte, m <= 4096 keeps S2 below 2^32.
struct Adler {
    uint32_t a = 1, b = 0;
    template <class F>
    __device__ __forceinline__ void add(uint32_t m, F&& byte_at, int lane) {
        uint32_t s1 = 0, s2 = 0;
        for (uint32_t j = lane; j < m; j += 64) {
            const uint32_t d = byte_at(j);
            s1 += d;
            s2 += (m - j) * d;
        }
        uint64_t t1 = s1, t2 = s2;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            t1 += __shfl_xor(t1, o, 64);
            t2 += __shfl_xor(t2, o, 64);
        }
        b = (uint32_t)((b + (uint64_t)m * a + t2) % 65521u);
        a = (uint32_t)((a + t1) % 65521u);
    }
    __device__ __forceinline__ uint32_t value() const { return b << 16 | a; }
};

struct PngInfo {
    uint32_t zlen;      // bytes of the gathered zlib stream
    int32_t status;
    uint32_t raw_n;     // filtered bytes the stream must inflate to: height x (1 + width x bytes per pixel IN THE FILE)
    uint8_t layout;     // kLayoutPlain: the file's pixels are the announced format; kLayoutPalette: 8-bit indices into
                        // PLTE -> RGB8; kLayoutGreyAlpha: 8-bit grey + alpha -> GRAY8 (the alpha byte is dropped, as the
                        // host path does: luma takes no alpha, DESIGN I1)
    uint8_t fbpp;       // bytes per pixel IN THE FILE (1, 2, 3, 4)
    uint16_t plte_n;    // palette entries
    uint32_t plte_off;  // offset of the PLTE data inside the file
};
constexpr uint8_t kLayoutPlain = 0, kLayoutPalette = 1, kLayoutGreyAlpha = 2;

__device__ __forceinline__ uint32_t be32(const uint8_t* p) {
    return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
}

// ---- chunk CRCs (PNG 5.5: CRC-32 of chunk type + data), verified like the reference's decoder does ----
// A chunk is cut into 64 slices, one per lane (byte-wise table CRC from LDS); the slices' CRCs combine linearly:
// crc(S0 | S1 | ...) = xor_j crc(S_j) * x^(8 * bytes after S_j) mod P (zlib's crc32_combine identity, reflected polynomial).
constexpr uint32_t kCrcPoly = 0xedb88320u;
__device__ __forceinline__ uint32_t crc_multmodp(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        p ^= ((a >> (31 - i)) & 1u) ? b : 0u;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}
// x^(8 n) mod P; x2n[k] = x^(2^k) mod P
__device__ __forceinline__ uint32_t crc_x8n(uint32_t n, const uint32_t* x2n) {
    uint32_t p = 1u << 31;
    for (uint32_t k = 3; __ballot(n != 0); k++, n >>= 1)
        p = (n & 1u) ? crc_multmodp(x2n[k & 31], p) : p;
    return p;
}
// tab: four 256-entry tables (slicing by 4: a word of input is four independent lookups instead of a chain of four)
__device__ uint32_t chunk_crc(const uint8_t* base, uint32_t total, const uint32_t* tab, const uint32_t* x2n, int lane) {
    const uint32_t per = (total + 63) / 64;
    const uint32_t lo = lane * per < total ? lane * per : total, hi = lo + per < total ? lo + per : total;
    uint32_t crc = 0xffffffffu;
    uint32_t b = lo;
    for (; b < hi && ((reinterpret_cast<uintptr_t>(base) + b) & 3u); b++) crc = tab[(crc ^ base[b]) & 255u] ^ (crc >> 8);
    auto word = [&](uint32_t w) {
        crc ^= w;
        crc = tab[768 + (crc & 255u)] ^ tab[512 + ((crc >> 8) & 255u)] ^ tab[256 + ((crc >> 16) & 255u)] ^ tab[crc >> 24];
    };
    for (; b + 16 <= hi; b += 16) {                    // four words requested together
        typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));     // word-aligned only
        const u32x4a4 v = *reinterpret_cast<const u32x4a4*>(base + b);
        word(v[0]), word(v[1]), word(v[2]), word(v[3]);
    }
    for (; b + 4 <= hi; b += 4) word(*reinterpret_cast<const uint32_t*>(base + b));
    for (; b < hi; b++) crc = tab[(crc ^ base[b]) & 255u] ^ (crc >> 8);
    const uint32_t c = hi > lo ? ~crc : 0u;
    uint32_t t = crc_multmodp(crc_x8n(total - hi, x2n), c);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) t ^= __shfl_xor(t, o, 64);
    return t;
}

// One wave per file: validate, gather IDAT payloads to zbuf + align16(offsets[i]).
__global__ __launch_bounds__(64) void png_scan_kernel(const uint8_t* __restrict__ png, const uint64_t* __restrict__ offsets,
                                                     size_t n, const UpItem* __restrict__ items, UpUniform uni,
                                                     uint8_t* __restrict__ zbuf, PngInfo* __restrict__ info) {
    __shared__ uint32_t crc_tab[1024];
    __shared__ uint32_t x2n[32];
    const size_t k = blockIdx.x;
    if (k >= n) return;
    const UpItem item = up_item(items, uni, k);
    const size_t img = item.file;                    // (the gather area is addressed by the FILE's offset, info by the entry)
    const uint32_t width = item.w, height = item.h;
    const int pixfmt = item.pixfmt;
    const int lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) {
        uint32_t c = i;
#pragma unroll
        for (int k = 0; k < 8; k++) c = (c & 1u) ? kCrcPoly ^ (c >> 1) : c >> 1;
        crc_tab[i] = c;
    }
    wave_lds_sync();
    for (int t = 1; t < 4; t++) {                      // T_t[i] = T_{t-1}[i] advanced by one zero byte
        for (uint32_t i = lane; i < 256; i += 64) {
            const uint32_t v = crc_tab[(t - 1) * 256 + i];
            crc_tab[t * 256 + i] = (v >> 8) ^ crc_tab[v & 255u];
        }
        wave_lds_sync();
    }
    {
        uint32_t v = 1u << 30;            // x^1
        for (int k = 0; k < 32; k++) {
            if (lane == 0) x2n[k] = v;
            v = crc_multmodp(v, v);
        }
    }
    wave_lds_sync();
    const uint8_t* p = png + offsets[img];
    const size_t len = (size_t)(offsets[img + 1] - offsets[img]);
    uint8_t* z = zbuf + ((offsets[img] + 15) & ~(uint64_t)15);
    int32_t status = 0;
    uint32_t zn = 0, fbpp = 1, plte_off = 0, plte_n = 0;
    uint8_t layout = kLayoutPlain;
    if (len < 8 + 25 + 12 || p[0] != 137 || p[1] != 80 || p[2] != 78 || p[3] != 71 || p[4] != 13 || p[5] != 10 || p[6] != 26 ||
        p[7] != 10 || be32(p + 8) != 13 || p[12] != 'I' || p[13] != 'H' || p[14] != 'D' || p[15] != 'R') {
        status = UCFP_E_MODALITY;
    } else {
        const uint32_t w = be32(p + 16), h = be32(p + 20);
        const int depth = p[24], ctype = p[25], comp = p[26], filt = p[27], lace = p[28];
        // what the file decodes to: palette -> RGB8 (through PLTE), grey + alpha -> GRAY8 (alpha dropped)
        const int fmt = ctype == 0 ? UCFP_PIX_GRAY8 : ctype == 2 ? UCFP_PIX_RGB8 : ctype == 6 ? UCFP_PIX_RGBA8
                      : ctype == 3 ? UCFP_PIX_RGB8 : ctype == 4 ? UCFP_PIX_GRAY8 : -1;
        layout = ctype == 3 ? kLayoutPalette : ctype == 4 ? kLayoutGreyAlpha : kLayoutPlain;
        fbpp = ctype == 0 || ctype == 3 ? 1u : ctype == 4 ? 2u : ctype == 2 ? 3u : 4u;
        if (w == 0 || h == 0 || comp != 0 || filt != 0 || lace > 1) status = UCFP_E_MODALITY;
        else if (depth != 8 || lace != 0 || fmt < 0 || fmt != pixfmt || w != width || h != height) status = UCFP_IMAGE_NEEDS_HOST;
    }
    if (status != UCFP_E_MODALITY && chunk_crc(p + 12, 4 + 13, crc_tab, x2n, lane) != be32(p + 29)) status = UCFP_E_MODALITY;   // IHDR
    if (status == 0) {
        size_t pos = 8 + 25;
        bool seen_idat = false, idat_done = false, seen_end = false;
        while (pos + 12 <= len) {
            const uint32_t cl = be32(p + pos);
            const uint8_t t0 = p[pos + 4], t1 = p[pos + 5], t2 = p[pos + 6], t3 = p[pos + 7];
            if (cl > 0x7fffffffu || pos + 12 + (size_t)cl > len) {
                status = UCFP_E_MODALITY;
                break;
            }
            if (chunk_crc(p + pos + 4, 4 + cl, crc_tab, x2n, lane) != be32(p + pos + 8 + cl)) {
                // a critical chunk with a bad CRC is damage; an ANCILLARY one (tEXt, pHYs ...) is a checksum-only
                // failure of data the pixels do not depend on: the host's decoder decides (P5)
                status = (t0 & 0x20) ? UCFP_IMAGE_NEEDS_HOST : UCFP_E_MODALITY;
                break;
            }
            if (t0 == 'I' && t1 == 'D' && t2 == 'A' && t3 == 'T') {
                if (idat_done) {
                    status = UCFP_E_MODALITY;
                    break;
                }
                seen_idat = true;
                const uint8_t* src = p + pos + 8;
                for (uint32_t i = lane; i < cl; i += 64) z[zn + i] = src[i];
                zn += cl;
            } else {
                if (seen_idat) idat_done = true;
                if (t0 == 'I' && t1 == 'E' && t2 == 'N' && t3 == 'D') {
                    seen_end = true;
                    break;
                }
                if (t0 == 't' && t1 == 'R' && t2 == 'N' && t3 == 'S') {
                    // PNG 11.3.2.1: simple transparency for grey (one 16-bit sample), RGB (three) and indexed colour (an alpha
                    // per palette entry, after PLTE), in front of the first IDAT.  It adds an alpha channel and changes no
                    // colour sample -- and luma takes no alpha (DESIGN I1; the host path's conversion drops it the same way):
                    // a well-formed one is skipped like any ancillary chunk, anything else is the host decoder's to judge
                    const uint32_t ct = p[25];
                    const bool fine = !seen_idat && ((ct == 0 && cl == 2) || (ct == 2 && cl == 6) || (ct == 3 && plte_n && cl >= 1 && cl <= plte_n));
                    if (!fine) status = UCFP_IMAGE_NEEDS_HOST;
                } else if (t0 == 'P' && t1 == 'L' && t2 == 'T' && t3 == 'E') {
                    // PNG 11.2.3: 1 .. 256 entries of 3 bytes, before the first IDAT, once
                    if (cl == 0 || cl % 3 != 0 || cl > 768 || seen_idat || plte_n) status = UCFP_E_MODALITY;
                    plte_off = (uint32_t)(pos + 8);
                    plte_n = cl / 3;
                } else if (!(t0 & 0x20)) status = UCFP_E_MODALITY;   // unknown critical chunk
                if (status == UCFP_E_MODALITY) break;
            }
            pos += 12 + (size_t)cl;
        }
        if (status == 0 && (!seen_idat || !seen_end)) status = UCFP_E_MODALITY;
        if (status == 0 && layout == kLayoutPalette && plte_n == 0) status = UCFP_E_MODALITY;   // indexed colour needs PLTE
    }
    // zero the tail word so that a staged partial word holds no stale bytes -- only for a file that will be inflated:
    // a rejected file of < 16 bytes shares its (16-byte aligned) gather address with the NEXT file, whose wave is
    // writing its zlib header there in this same launch (a valid file's zn + 4 stays inside its own byte range).
    if (status == 0 && lane < 4) z[zn + lane] = 0;
    if (lane == 0)
        info[k] = PngInfo{zn, status, height * (1u + width * fbpp), layout, (uint8_t)fbpp, (uint16_t)plte_n, plte_off};
}

#ifdef PNG_PROF
__device__ unsigned long long g_png_prof[16];
#define PROF_T0() long long _t = clock64()
#define PROF_ADD(slot) do { const long long _n = clock64(); _acc[slot] += (unsigned long long)(_n - _t); _t = _n; } while (0)
#define PROF_CNT(slot, v) (_acc[slot] += (unsigned long long)(v))
#else
#define PROF_T0()
#define PROF_ADD(slot)
#define PROF_CNT(slot, v)
#endif
#ifdef PNG_DEBUG
#define PNG_BAD(code) (bad = true, (lane == 0 ? printf("png img %d bad %d bp %u outpos %u B %d\n", (int)img, code, bp, outpos, B) : 0))
#else
#define PNG_BAD(code) (bad = true)
#endif
// One wave per image: zlib stream -> filtered scanlines (raw_n bytes expected).
template <class C>
__global__ __launch_bounds__(64) void png_inflate_kernel(const uint8_t* __restrict__ zbuf, const uint64_t* __restrict__ offsets,
                                                        size_t n, const UpItem* __restrict__ items, UpUniform uni,
                                                        PngInfo* __restrict__ info, uint8_t* __restrict__ raw) {
    __shared__ InflateLds<C> L;
    const size_t img = blockIdx.x;                   // entry of the batch: info[img]
    if (img >= n) return;
    const int lane = threadIdx.x;
    if (info[img].status != 0) return;
    const UpItem item = up_item(items, uni, img);
    const uint32_t raw_n = info[img].raw_n;
    const uint8_t* z = zbuf + ((offsets[item.file] + 15) & ~(uint64_t)15);
    const uint32_t zlen = info[img].zlen;
    const uint32_t zwords = (zlen + 3) / 4, total_bits = zlen * 8;
    uint8_t* out = raw + item.aux_off;
    bool bad = zlen < 6;
    if (!bad) {
        const uint32_t cmf = z[0], flg = z[1];
        bad = (cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8 | flg) % 31) != 0 || (flg & 0x20);
    }
    uint32_t bp = 16, outpos = 0;
    bool last = false, checksum_only = false;
    int B = C::kMaxB;
    Adler adler;
#ifdef PNG_PROF
    unsigned long long _acc[16] = {0};
#endif
    PROF_T0();
    while (!bad && !last) {
        uint32_t s0 = stage_load(L, z, zwords, bp, lane);
        uint32_t rel = bp - s0;
        last = peek_u(L.stage, rel, 1);
        const uint32_t type = peek_u(L.stage, rel + 1, 2);
        rel += 3;
        if (type == 3 || bp + 3 > total_bits) {
            PNG_BAD(1);
            break;
        }
        if (type == 0) {
            // stored: to the byte boundary, LEN, NLEN, then LEN bytes through the window
            uint32_t byte = (s0 + rel + 7) >> 3;
            if (byte + 4 > zlen) {
                PNG_BAD(2);
                break;
            }
            const uint32_t len = z[byte] | (uint32_t)z[byte + 1] << 8, nlen = z[byte + 2] | (uint32_t)z[byte + 3] << 8;
            byte += 4;
            if ((len ^ 0xffffu) != nlen || byte + len > zlen || outpos + len > raw_n) {
                PNG_BAD(3);
                break;
            }
            for (uint32_t k = lane; k < len; k += 64) out[outpos + k] = z[byte + k];
            for (uint32_t p0 = 0; p0 < len; p0 += 4096) {
                const uint32_t m = len - p0 < 4096 ? len - p0 : 4096;
                adler.add(m, [&](uint32_t j) { return (uint32_t)z[byte + p0 + j]; }, lane);
            }
            __threadfence_block();
            outpos += len;
            {
                // the LDS copy of the last kHist bytes, re-read from what was just written
                const uint32_t nb2 = outpos > kHist ? (outpos - kHist) & ~3u : 0u;
                for (uint32_t k = nb2 + lane; k < outpos; k += 64) L.rb[k - nb2] = out[k];
                wave_lds_sync();
            }
            bp = (byte + len) * 8;
            continue;
        }
        // ---- the block's two codes ----
        int used = 0;
        if (type == 1) {
            for (int s = lane; s < 320; s += 64) L.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5;
            wave_lds_sync();
            build_table<false>(L.lens, 288, L.lit, L.ll_count, L.ll_sorted, lane, &used);
            build_table<true>(L.lens + 288, 30, L.dst, L.d_count, L.d_sorted, lane, &used);
        } else {
            const uint32_t nlen = peek_u(L.stage, rel, 5) + 257, ndist = peek_u(L.stage, rel + 5, 5) + 1,
                           ncode = peek_u(L.stage, rel + 10, 4) + 4;
            rel += 14;
            if (nlen > 286 || ndist > 30) {
                PNG_BAD(4);
                break;
            }
            // the code-length code: 19 symbols in the permuted order of 3.2.7, decoded through the distance-table slots
            if (lane < 19) L.lens[lane] = 0;
            wave_lds_sync();
            if (lane < (int)ncode) {
                const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                L.lens[order[lane]] = (uint8_t)peek_u(L.stage, rel + 3 * lane, 3);
            }
            rel += 3 * ncode;
            wave_lds_sync();
            // a 7-bit table in L.dst (entries: symbol | code length << 20)
            {
                for (int i = lane; i < 128; i += 64) L.dst[i] = 0;
                const uint32_t l = lane < 19 ? L.lens[lane] : 0;
                uint32_t cnt[8], next[8];
                int left = 1, code = 0;
                cnt[0] = next[0] = 0;
#pragma unroll
                for (int q = 1; q <= 7; q++) {
                    cnt[q] = (uint32_t)__popcll(__ballot(l == (uint32_t)q));
                    left = (left << 1) - (int)cnt[q];
                    next[q] = (uint32_t)code;
                    code = (code + (int)cnt[q]) << 1;
                }
                if (left != 0) {   // the code-length code must be complete
                    PNG_BAD(5);
                    break;
                }
                uint32_t rank = 0, nx = 0;
#pragma unroll
                for (int q = 1; q <= 7; q++)
                    if (l == (uint32_t)q) {
                        rank = (uint32_t)__popcll(__ballot(l == (uint32_t)q) & ((1ull << lane) - 1));
                        nx = next[q];
                    }
                wave_lds_sync();
                if (l) {
                    const uint32_t rev = __brev(nx + rank) >> (32 - l);
                    for (uint32_t i = rev; i < 128; i += 1u << l) L.dst[i] = (uint32_t)lane | l << 20;
                }
                wave_lds_sync();
            }
            // the nlen + ndist code lengths, run-length coded (uniform: every lane walks the same bits)
            uint32_t idx = 0, prev = 0;
            const uint32_t want = nlen + ndist;
            while (idx < want) {
                if (rel + 14 > (uint32_t)(C::kStageWords - 2) * 32) {   // a header is at most ~4.5 kbit: cannot happen in a valid stream
                    PNG_BAD(6);
                    break;
                }
                const uint32_t e = L.dst[peek_u(L.stage, rel, 7)];
                const uint32_t cl = e >> 20, sym = e & 31u;
                if (cl == 0) {
                    PNG_BAD(7);
                    break;
                }
                rel += cl;
                if (sym < 16) {
                    if (lane == 0) L.lens[idx] = (uint8_t)sym;
                    prev = sym;
                    idx++;
                } else {
                    uint32_t rep, v = 0;
                    if (sym == 16) {
                        if (idx == 0) {
                            PNG_BAD(8);
                            break;
                        }
                        v = prev;
                        rep = 3 + peek_u(L.stage, rel, 2);
                        rel += 2;
                    } else if (sym == 17) {
                        rep = 3 + peek_u(L.stage, rel, 3);
                        rel += 3;
                        prev = 0;
                    } else {
                        rep = 11 + peek_u(L.stage, rel, 7);
                        rel += 7;
                        prev = 0;
                    }
                    if (idx + rep > want) {
                        PNG_BAD(9);
                        break;
                    }
                    if (lane < (int)rep) L.lens[idx + lane] = (uint8_t)v;
                    if (lane + 64 < (int)rep) L.lens[idx + lane + 64] = (uint8_t)v;
                    if (lane + 128 < (int)rep) L.lens[idx + lane + 128] = (uint8_t)v;
                    idx += rep;
                }
            }
            if (bad) break;
            wave_lds_sync();
            if (L.lens[256] == 0) {
                PNG_BAD(10);
                break;
            }
            // distance lengths move to their own 32-aligned place: lens[288 ..)
            const uint32_t dl = lane < (int)ndist ? L.lens[nlen + lane] : 0;
            wave_lds_sync();
            if (lane < 32) L.lens[288 + lane] = (uint8_t)dl;
            for (int s = nlen + lane; s < 288; s += 64) L.lens[s] = 0;
            wave_lds_sync();
            int r = build_table<false>(L.lens, 288, L.lit, L.ll_count, L.ll_sorted, lane, &used);
            if (r < 0 || (r > 0 && used != 1)) {
                PNG_BAD(11);
                break;
            }
            r = build_table<true>(L.lens + 288, 30, L.dst, L.d_count, L.d_sorted, lane, &used);
            if (r < 0 || (r > 0 && used > 1)) {
                PNG_BAD(12);
                break;
            }
        }
        wave_lds_sync();
        bp = s0 + rel;
        PROF_ADD(0);      // block header + tables
        PROF_CNT(8, 1);
        // ---- the block's symbols, one speculation round after the other ----
        bool eob = false;
        while (!eob && !bad) {
            s0 = stage_load(L, z, zwords, bp, lane);
            const uint32_t r0 = bp - s0;
            const uint32_t vbase = r0 + (uint32_t)(lane * B), limit = vbase + (uint32_t)B;
            ParseCache cache;
            cache.clear();
            Parse P{vbase, vbase, 0};
            uint32_t start = vbase;
            int nvalid = 0;
            PROF_ADD(1);  // stage
            PROF_CNT(9, 1);
#ifndef UCFP_PNG_NO_WARMUP
            // Warm-up (round 4).  A lane's guess j B is almost never a code boundary, so its first parse is wrong at the start and
            // only its EXIT is (usually) right; the lane behind it then needs a second parse from that exit, and a third when its
            // predecessor had not fallen into step within B bits: 4.3 parses per subsequence at level 1.  Instead every lane first
            // runs through its PREDECESSOR's subsequence, from that one's guess: by the time it reaches its own first bit it is in
            // step (or the chain below repairs it), and its counted parse starts on a true boundary -- the chain then mostly
            // confirms what is there.  One uncounted parse of B bits buys back two counted ones.
            {
                const Parse W = parse_sub<false, C>(L, lane > 0 ? vbase - (uint32_t)B : vbase, vbase, 0, 0, 0);   // (lane 0 knows its start)
                if (lane > 0 && !W.stopped()) start = W.exit;
            }
#endif
            for (;;) {
                // the lane before stopped short of its limit (an end-of-block or an invalid code, real or in a parse
                // from a wrong start): nothing to continue here
                const bool skip = start < vbase;
                bool miss = !skip && !cache.find(start, &P);
                if (skip) P = Parse{start, start, 2u << 30};
                if (__ballot(miss)) {
                    PROF_CNT(10, 1);
                    if (miss) {
                        P = parse_sub<false, C>(L, start, limit, 0, 0, 0);
                        cache.put(P);
                    }
                }
                const uint32_t prev_exit = __shfl_up(P.exit, 1, 64);
                const bool dirty = lane > 0 && prev_exit != P.start;
                if (dirty) start = prev_exit;
                const uint64_t dm = __ballot(dirty);
                const int f = dm ? __builtin_ctzll(dm) : 64;                 // lanes below f continue lane 0's parse
                const uint64_t stop = __ballot(P.stopped()) & (f == 64 ? ~0ull : ((1ull << f) - 1));
                if (stop) {
                    nvalid = __builtin_ctzll(stop) + 1;
                    break;
                }
                if (f == 64) {
                    nvalid = 64;
                    break;
                }
            }
            PROF_ADD(2);  // sync rounds
            // how many of the confirmed lanes fit this round's buffer and match list
            const bool in = lane < nvalid;
            const uint32_t cb = wave_incl_scan(in ? P.nbytes() : 0, lane), cm = wave_incl_scan(in ? P.nmatch() : 0, lane);
            const bool fits = in && cb <= C::kIterOut && cm <= C::kMatchCap && outpos + cb <= raw_n;
            const int take = __popcll(__ballot(fits));                        // a prefix: the sums are monotonic
            if (take == 0) {
                const uint32_t b0 = __shfl(P.nbytes(), 0, 64);
                if (outpos + b0 > raw_n || B <= 8) {                          // more output than the image has rows for
                    PNG_BAD(13);
                    break;
                }
                B = B / 4 < 8 ? 8 : B / 4;                                    // extremely dense matches: shorter subsequences
                continue;
            }
            const uint32_t rb_base = outpos > kHist ? (outpos - kHist) & ~3u : 0u;
            bool far = false;
            PROF_ADD(3);  // scans / cut
            if (lane < take) far = parse_sub<true, C>(L, P.start, limit, outpos + cb - P.nbytes(), rb_base, cm - P.nmatch()).err() && !P.err();
            wave_lds_sync();
            PROF_ADD(4);  // emit
            PROF_CNT(11, take);
            if (__ballot(far)) {
                PNG_BAD(14);
                break;
            }
            const uint32_t add = __shfl(cb, take - 1, 64), nm = __shfl(cm, take - 1, 64);
#ifdef PNG_PROF
            resolve_matches(L, out, rb_base, nm, lane, &_acc[13], &_acc[14]);
#else
            resolve_matches(L, out, rb_base, nm, lane);
#endif
            PROF_ADD(5);  // matches
            PROF_CNT(12, nm);
            flush_out(L, out, outpos, outpos + add, rb_base, lane);
            adler.add(add, [&](uint32_t j) { return (uint32_t)L.rb[outpos + j - rb_base]; }, lane);
            __threadfence_block();                                           // later rounds read these bytes back
            outpos += add;
            {
                // slide the window: the last kHist bytes stay in LDS at the base the next round will use
                const uint32_t nb2 = outpos > kHist ? (outpos - kHist) & ~3u : 0u;
                const uint32_t shift = nb2 - rb_base, keep = (outpos - nb2 + 3) / 4;      // words; shift is a multiple of 4
                if (shift) {
                    uint32_t wv[(kHist + 4) / 4 / 64 + 1];
#pragma unroll
                    for (int i = 0; i < (int)((kHist + 4) / 4 / 64 + 1); i++) {
                        const uint32_t wi = lane + 64 * i;
                        wv[i] = wi < keep ? *reinterpret_cast<const uint32_t*>(&L.rb[shift + 4 * wi]) : 0u;
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int i = 0; i < (int)((kHist + 4) / 4 / 64 + 1); i++) {
                        const uint32_t wi = lane + 64 * i;
                        if (wi < keep) *reinterpret_cast<uint32_t*>(&L.rb[4 * wi]) = wv[i];
                    }
                    wave_lds_sync();
                }
            }
            PROF_ADD(6);  // flush
            const bool t_eob = __shfl((int)P.eob(), take - 1, 64), t_err = __shfl((int)P.err(), take - 1, 64);
            bp = s0 + __shfl(P.exit, take - 1, 64);
            if (t_err || bp > total_bits) PNG_BAD(15);
            eob = t_eob;
            if (take == 64 && add < C::kIterOut / 4 && B < C::kMaxB) B *= 2;
        }
    }
    if (!bad && outpos != raw_n) PNG_BAD(16);
    if (!bad) {
        // the Adler-32 of the output follows the deflate data at the next byte boundary, most significant byte first
        const uint32_t e = (bp + 7) / 8;
        // A stream that inflated to exactly the announced length but whose CHECKSUM is absent or wrong is not rejected
        // here: decoders differ on it (the reference's `png` crate can be configured either way, and its defaults
        // have changed between releases), so the file goes to the host's decoder, which decides (P5).
        if (e + 4 > zlen ||
            ((uint32_t)z[e] << 24 | (uint32_t)z[e + 1] << 16 | (uint32_t)z[e + 2] << 8 | z[e + 3]) != adler.value())
            checksum_only = true;
    }
#ifdef PNG_PROF
    if (lane == 0)
        for (int i = 0; i < 16; i++) atomicAdd(&g_png_prof[i], _acc[i]);
#endif
    if (lane == 0) {
        if (bad) info[img].status = UCFP_E_MODALITY;
        else if (checksum_only) info[img].status = UCFP_IMAGE_NEEDS_HOST;
    }
}

// ================= two-pass inflate (round 4): Huffman decoding by several waves per file, then LZ77 ================
// png_inflate_kernel above is bound by its own latencies: a parse step is ~1000 cycles of dependent LDS round trips, one wave
// per file means one wave per SIMD at a thousand files, and its 22-62 KB of LDS per wave (tables + stage + match list +
// window) rule out a second wave.  The split:
//   png_huff_kernel<BITS, W>   W waves per file: the same speculation rounds over W x 64 subsequences (the chain of
//                              "start == predecessor's exit" crosses waves through LDS), and the accepted parse is written as
//                              TOKENS to global memory -- 16-bit words: a literal is its byte; a match is 0x8000 | (len - 3)
//                              followed by dist - 1 (a word is a distance iff the word before it has bit 15 set: distances
//                              are < 2^15, so the stream can be cut anywhere).  LDS: tables + stage only.
//   png_lz_kernel              one wave per file: 1024 token words per round -> literals and the match list in LDS, matches
//                              resolved as before (resolve_matches), bytes flushed, Adler-32 checked.
struct PngTok {
    uint32_t ntok;      // 16-bit token words written
    uint32_t end_bit;   // bit position behind the last block (the Adler-32 follows at the next byte boundary)
};

struct Parse2 {
    uint32_t start, exit, packed, ntok;      // packed as Parse; ntok: token words (1 per literal, 2 per match)
    __device__ __forceinline__ uint32_t nbytes() const { return packed & 0x1ffffu; }
    __device__ __forceinline__ bool eob() const { return (packed >> 30) & 1u; }
    __device__ __forceinline__ bool err() const { return packed >> 31; }
    __device__ __forceinline__ bool stopped() const { return packed >> 30; }
};

template <int BITS, int W>
struct HuffLds {
    static constexpr int kStageWords = W * 64 * BITS / 32 + 64;      // W x 64 subsequences + overshoot + the fetch window
    uint32_t lit[1 << kRoot];
    uint32_t dst[1 << kDRoot];
    uint32_t stage[kStageWords];
    uint16_t ll_sorted[288];
    uint16_t d_sorted[32];
    uint16_t ll_count[16];
    uint16_t d_count[16];
    uint8_t lens[320];
    uint32_t xexit[2][W], xfirst[2][W], xstop[2][W];   // per chain iteration (parity): last lane's exit, first dirty / stopped lane
    alignas(8) uint32_t wsum[2][W];                     // workgroup scans (W 64-bit sums)
    uint32_t bc[12];                                    // broadcasts of uniform decisions
};

// parse_sub for the token kernel: EMIT = false counts (bytes, matches, token words), EMIT = true writes the tokens.
template <bool EMIT, class LT>
__device__ __forceinline__ Parse2 parse_tok(LT& L, uint32_t start, uint32_t limit, uint32_t out_pos, uint16_t* __restrict__ tok) {
    // A divergent loop, lanes leaving as they finish, and plain divergent branches for the rare and the per-kind work: written
    // with wave-uniform `if (__ballot(...))` guards around predicated bodies the step was ~45 vector + ~35 scalar instructions,
    // a third of them materialising and re-testing predicates; four waves per SIMD are bound by issue, not by the LDS round trips.
    uint32_t pos = start, nb = 0, nm = 0, nt = 0, flags = 0;
    while (pos < limit) {
        const uint32_t w = pos >> 5, sh = pos & 31;
        const uint32_t w0 = L.stage[w], w1 = L.stage[w + 1], w2 = L.stage[w + 2];
        const uint32_t x = __builtin_amdgcn_alignbit(w1, w0, sh), x2 = __builtin_amdgcn_alignbit(w2, w1, sh);
        uint32_t e = L.lit[x & ((1u << kRoot) - 1)];
        if ((e >> 24) == kSlow) e = slow_code32(x, L.ll_count, L.ll_sorted, false);
        const uint32_t cl = (e >> 20) & 15u, kind = e >> 24, ex = (e >> 16) & 15u;
        const uint32_t s1 = cl + ex;
        if (cl == 0) {
            flags |= 2u;
            break;
        }
        if (kind == kLen) {
            const uint32_t len = (e & 0xffffu) + ((x >> cl) & ((1u << ex) - 1u));
            const uint32_t y = __builtin_amdgcn_alignbit(x2, x, s1);
            uint32_t d = L.dst[y & ((1u << kDRoot) - 1)];
            if ((d >> 24) == kSlow) d = slow_code32(y, L.d_count, L.d_sorted, true);
            const uint32_t dl = (d >> 20) & 15u, dex = (d >> 16) & 15u;
            if (dl == 0) {
                flags |= 2u;
                break;
            }
            const uint32_t dist = (d & 0xffffu) + ((y >> dl) & ((1u << dex) - 1u));
            if (EMIT) {
                if (dist > out_pos + nb) flags |= 2u;          // reaches back before the first byte of the image
                tok[nt] = (uint16_t)(0x8000u | (len - 3u));
                tok[nt + 1] = (uint16_t)(dist - 1u);
            }
            pos += s1 + dl + dex;
            nb += len;
            nm += 1u;
            nt += 2u;
        } else if (kind == kLit) {
            if (EMIT) tok[nt] = (uint16_t)(e & 0xffu);
            pos += s1;
            nb += 1u;
            nt += 1u;
        } else {           // end of block (any other kind: not a code)
            flags |= kind == kEob ? 1u : 2u;
            pos += kind == kEob ? s1 : 0u;
            break;
        }
    }
    return Parse2{start, pos, nb | nm << 17 | flags << 30, nt};
}

// The parses a lane has made in this round, by start position (see ParseCache).  Four register vectors: as an array of structs
// indexed by `next` (or as scalars / structs under selects) the compiler keeps it in scratch memory -- 80-112 bytes per lane and a
// round trip to memory at every find / put.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct ParseCache2 {
    u32x4 s, e, p, n;
    uint32_t next;
    __device__ __forceinline__ void clear() {
        s = u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        e = p = n = u32x4{0, 0, 0, 0};
        next = 0;
    }
    __device__ __forceinline__ bool find(uint32_t start, Parse2* out) const {
        const bool h0 = s.x == start, h1 = s.y == start, h2 = s.z == start, h3 = s.w == start;
        const bool hit = h0 || h1 || h2 || h3;
        if (hit) {
            out->start = start;
            out->exit = h0 ? e.x : h1 ? e.y : h2 ? e.z : e.w;
            out->packed = h0 ? p.x : h1 ? p.y : h2 ? p.z : p.w;
            out->ntok = h0 ? n.x : h1 ? n.y : h2 ? n.z : n.w;
        }
        return hit;
    }
    __device__ __forceinline__ void put(const Parse2& q) {
        s[next] = q.start;
        e[next] = q.exit;
        p[next] = q.packed;
        n[next] = q.ntok;
        next = (next + 1) & 3u;
    }
};

// W waves per file: zlib stream -> token words.
template <int BITS, int W>
__global__ __launch_bounds__(64 * W) void png_huff_kernel(const uint8_t* __restrict__ zbuf, const uint64_t* __restrict__ offsets, size_t n,
                                                         const UpItem* __restrict__ items, UpUniform uni, PngInfo* __restrict__ info,
                                                         uint16_t* __restrict__ tokens, PngTok* __restrict__ tinfo, uint32_t warm,
                                                         uint32_t max_iter) {
    using LT = HuffLds<BITS, W>;
    __shared__ LT L;
    constexpr int T = 64 * W;
    const size_t img = blockIdx.x;
    if (img >= n) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (info[img].status != 0) return;
    const UpItem item = up_item(items, uni, img);
    const uint32_t raw_n = info[img].raw_n;
    const uint8_t* z = zbuf + ((offsets[item.file] + 15) & ~(uint64_t)15);
    const uint32_t zlen = info[img].zlen;
    const uint32_t zwords = (zlen + 3) / 4, total_bits = zlen * 8;
    uint16_t* tok = tokens + item.aux_off;      // (2 bytes per filtered byte at most: the file's token area is its raw area, doubled)
    auto wg_sync = [&]() {
        if (W == 1) wave_lds_sync();
        else __syncthreads();
    };
    auto stage_fill = [&](uint32_t bp, uint32_t words) {
        const uint32_t w0 = bp >> 5;
        const uint32_t* zw = reinterpret_cast<const uint32_t*>(z);
        for (uint32_t i = tid; i < words; i += T) L.stage[i] = (w0 + i < zwords) ? zw[w0 + i] : 0u;
        wg_sync();
        return w0 << 5;
    };
    // inclusive scan of a 64-bit pair (bytes in the low word, token words in the high) over the workgroup; total = the sum of all
    auto wg_scan = [&](uint64_t v, uint64_t& total, int slot) -> uint64_t {
        uint64_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint64_t o = (uint64_t)__shfl_up((long long)incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (W == 1) {
            total = (uint64_t)__shfl((long long)incl, 63, 64);
            return incl;
        }
        uint64_t* ws = reinterpret_cast<uint64_t*>(&L.wsum[0][0]) + 0;      // (wsum[2][W] u32 = W u64)
        (void)slot;
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        uint64_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < W; w++) {
            const uint64_t t = ws[w];
            before += w < wave ? t : 0ull;
            all += t;
        }
        __syncthreads();
        total = all;
        return incl + before;
    };
    bool bad = zlen < 6;
    if (!bad) {
        const uint32_t cmf = z[0], flg = z[1];
        bad = (cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8 | flg) % 31) != 0 || (flg & 0x20);
    }
    uint32_t bp = 16, outpos = 0, tpos = 0;
    bool last = false;
#ifdef PNG_PROF
    unsigned long long _acc[16] = {0};
#endif
    PROF_T0();
    while (!bad && !last) {
        // ---- block header and tables: wave 0, from the first 192 words behind bp; the others wait
        const uint32_t hs0 = stage_fill(bp, 192 < LT::kStageWords ? 192 : LT::kStageWords);
        if (wave == 0) {
            uint32_t rel = bp - hs0;
            uint32_t hbad = 0, stored_byte = 0, stored_len = 0;
            const uint32_t is_last = peek_u(L.stage, rel, 1), type = peek_u(L.stage, rel + 1, 2);
            rel += 3;
            int used = 0;
            if (type == 3 || bp + 3 > total_bits) {
                hbad = 1;
            } else if (type == 0) {
                uint32_t byte = (hs0 + rel + 7) >> 3;
                if (byte + 4 > zlen) {
                    hbad = 2;
                } else {
                    const uint32_t len = z[byte] | (uint32_t)z[byte + 1] << 8, nlen = z[byte + 2] | (uint32_t)z[byte + 3] << 8;
                    byte += 4;
                    if ((len ^ 0xffffu) != nlen || byte + len > zlen || outpos + len > raw_n) hbad = 3;
                    stored_byte = byte;
                    stored_len = len;
                }
            } else if (type == 1) {
                for (int s = lane; s < 320; s += 64) L.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5;
                wave_lds_sync();
                build_table<false>(L.lens, 288, L.lit, L.ll_count, L.ll_sorted, lane, &used);
                build_table<true>(L.lens + 288, 30, L.dst, L.d_count, L.d_sorted, lane, &used);
            } else {
                const uint32_t nlen = peek_u(L.stage, rel, 5) + 257, ndist = peek_u(L.stage, rel + 5, 5) + 1,
                               ncode = peek_u(L.stage, rel + 10, 4) + 4;
                rel += 14;
                if (nlen > 286 || ndist > 30) hbad = 4;
                if (!hbad) {
                    if (lane < 19) L.lens[lane] = 0;
                    wave_lds_sync();
                    if (lane < (int)ncode) {
                        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                        L.lens[order[lane]] = (uint8_t)peek_u(L.stage, rel + 3 * lane, 3);
                    }
                    rel += 3 * ncode;
                    wave_lds_sync();
                    for (int i = lane; i < 128; i += 64) L.dst[i] = 0;
                    const uint32_t l = lane < 19 ? L.lens[lane] : 0;
                    uint32_t cnt[8], next[8];
                    int left = 1, code = 0;
                    cnt[0] = next[0] = 0;
#pragma unroll
                    for (int q = 1; q <= 7; q++) {
                        cnt[q] = (uint32_t)__popcll(__ballot(l == (uint32_t)q));
                        left = (left << 1) - (int)cnt[q];
                        next[q] = (uint32_t)code;
                        code = (code + (int)cnt[q]) << 1;
                    }
                    if (left != 0) hbad = 5;
                    uint32_t rank = 0, nx = 0;
#pragma unroll
                    for (int q = 1; q <= 7; q++)
                        if (l == (uint32_t)q) {
                            rank = (uint32_t)__popcll(__ballot(l == (uint32_t)q) & ((1ull << lane) - 1));
                            nx = next[q];
                        }
                    wave_lds_sync();
                    if (!hbad && l) {
                        const uint32_t rev = __brev(nx + rank) >> (32 - l);
                        for (uint32_t i = rev; i < 128; i += 1u << l) L.dst[i] = (uint32_t)lane | l << 20;
                    }
                    wave_lds_sync();
                }
                uint32_t idx = 0, prev = 0;
                const uint32_t want = nlen + ndist;
                while (!hbad && idx < want) {
                    if (rel + 14 > (uint32_t)(192 - 2) * 32) {      // a header is at most ~4.5 kbit
                        hbad = 6;
                        break;
                    }
                    const uint32_t e = L.dst[peek_u(L.stage, rel, 7)];
                    const uint32_t cl = e >> 20, sym = e & 31u;
                    if (cl == 0) {
                        hbad = 7;
                        break;
                    }
                    rel += cl;
                    if (sym < 16) {
                        if (lane == 0) L.lens[idx] = (uint8_t)sym;
                        prev = sym;
                        idx++;
                    } else {
                        uint32_t rep, v = 0;
                        if (sym == 16) {
                            if (idx == 0) {
                                hbad = 8;
                                break;
                            }
                            v = prev;
                            rep = 3 + peek_u(L.stage, rel, 2);
                            rel += 2;
                        } else if (sym == 17) {
                            rep = 3 + peek_u(L.stage, rel, 3);
                            rel += 3;
                            prev = 0;
                        } else {
                            rep = 11 + peek_u(L.stage, rel, 7);
                            rel += 7;
                            prev = 0;
                        }
                        if (idx + rep > want) {
                            hbad = 9;
                            break;
                        }
                        if (lane < (int)rep) L.lens[idx + lane] = (uint8_t)v;
                        if (lane + 64 < (int)rep) L.lens[idx + lane + 64] = (uint8_t)v;
                        if (lane + 128 < (int)rep) L.lens[idx + lane + 128] = (uint8_t)v;
                        idx += rep;
                    }
                }
                if (!hbad) {
                    wave_lds_sync();
                    if (L.lens[256] == 0) hbad = 10;
                }
                if (!hbad) {
                    const uint32_t dl = lane < (int)ndist ? L.lens[nlen + lane] : 0;
                    wave_lds_sync();
                    if (lane < 32) L.lens[288 + lane] = (uint8_t)dl;
                    for (int s = nlen + lane; s < 288; s += 64) L.lens[s] = 0;
                    wave_lds_sync();
                    int r = build_table<false>(L.lens, 288, L.lit, L.ll_count, L.ll_sorted, lane, &used);
                    if (r < 0 || (r > 0 && used != 1)) hbad = 11;
                    if (!hbad) {
                        r = build_table<true>(L.lens + 288, 30, L.dst, L.d_count, L.d_sorted, lane, &used);
                        if (r < 0 || (r > 0 && used > 1)) hbad = 12;
                    }
                }
            }
            wave_lds_sync();
            if (lane == 0) {
                L.bc[0] = hbad;
                L.bc[1] = is_last;
                L.bc[2] = type;
                L.bc[3] = hs0 + rel;
                L.bc[4] = stored_byte;
                L.bc[5] = stored_len;
            }
        }
        wg_sync();
        const uint32_t hbad = L.bc[0], type = L.bc[2];
        last = L.bc[1] != 0;
        const uint32_t bp_sym = L.bc[3], st_byte = L.bc[4], st_len = L.bc[5];
        wg_sync();                                   // (bc is rewritten below)
        if (hbad) {
            bad = true;
            break;
        }
        if (type == 0) {
            // stored: its bytes become literal tokens
            for (uint32_t k = tid; k < st_len; k += T) tok[tpos + k] = (uint16_t)z[st_byte + k];
            tpos += st_len;
            outpos += st_len;
            bp = (st_byte + st_len) * 8;
            continue;
        }
        bp = bp_sym;
        PROF_ADD(0);      // header + tables
        // ---- the block's symbols, one speculation round of T subsequences after the other ----
        bool eob = false;
        while (!eob && !bad) {
            PROF_CNT(9, 1);
            if (tid == 0) L.bc[6] = L.bc[7] = 0;     // (read for the last time two barriers ago; the stage fill's barrier publishes it)
            const uint32_t s0 = stage_fill(bp, LT::kStageWords);
            const uint32_t r0 = bp - s0;
            const uint32_t vbase = r0 + (uint32_t)tid * BITS, limit = vbase + BITS;
            ParseCache2 cache;
            cache.clear();
            Parse2 P{vbase, vbase, 0, 0};
            uint32_t start = vbase;
            PROF_ADD(1);  // stage
            if (warm) {   // warm-up through the predecessor's subsequence (see png_inflate_kernel)
                // (warm = 2 / 4: only the last half / quarter of it -- 1000 files x 4 waves 186 k / 176 k images/s against 217 k; warm = 3: the
                // two subsequences before mine)
                const uint32_t back = warm == 3 ? (tid > 1 ? 2u * BITS : (uint32_t)tid * BITS) : (tid > 0 ? BITS / warm : 0u);
                const Parse2 Wm = parse_tok<false>(L, vbase - back, vbase, 0, nullptr);
                if (tid > 0 && !Wm.stopped()) start = Wm.exit;
            }
            uint32_t nvalid = 0;
            PROF_ADD(2);  // warm-up
            for (uint32_t it = 0;; it++) {
                PROF_CNT(10, 1);
                const uint32_t par = it & 1u;
                const bool skip = start < vbase;
                bool miss = !skip && !cache.find(start, &P);
                if (skip) P = Parse2{start, start, 2u << 30, 0};
                if (__ballot(miss)) {
                    if (miss) {
                        P = parse_tok<false>(L, start, limit, 0, nullptr);
                        cache.put(P);
                    }
                }
                if (W > 1) {
                    if (lane == 63) L.xexit[par][wave] = P.exit;
                    __syncthreads();
                }
                uint32_t prev_exit = __shfl_up(P.exit, 1, 64);
                if (W > 1 && lane == 0 && wave > 0) prev_exit = L.xexit[par][wave - 1];
                const bool dirty = tid > 0 && prev_exit != P.start;
                if (dirty) start = prev_exit;
                const uint64_t dm = __ballot(dirty), sm = __ballot(P.stopped());
                uint32_t f, sfirst;
                if (W == 1) {
                    f = dm ? (uint32_t)__builtin_ctzll(dm) : (uint32_t)T;
                    sfirst = sm ? (uint32_t)__builtin_ctzll(sm) : (uint32_t)T;
                } else {
                    if (lane == 0) {
                        L.xfirst[par][wave] = dm ? (uint32_t)__builtin_ctzll(dm) : 64u;
                        L.xstop[par][wave] = sm ? (uint32_t)__builtin_ctzll(sm) : 64u;
                    }
                    __syncthreads();
                    f = T;
                    sfirst = T;
#pragma unroll
                    for (int w = W - 1; w >= 0; w--) {
                        const uint32_t a = L.xfirst[par][w], b = L.xstop[par][w];
                        if (a < 64u) f = (uint32_t)w * 64u + a;
                        if (b < 64u) sfirst = (uint32_t)w * 64u + b;
                    }
                }
                if (sfirst < f) {            // the first stopped subsequence lies in the confirmed prefix
                    nvalid = sfirst + 1;
                    break;
                }
                if (f == (uint32_t)T) {
                    nvalid = T;
                    break;
                }
                if (it + 1 >= max_iter && f >= 64u) {      // enough chasing: the confirmed prefix goes out, the rest is next round's
                    nvalid = f;
                    break;
                }
            }
            PROF_ADD(3);  // chain iterations
            PROF_CNT(11, nvalid);
            // ---- the confirmed prefix: positions of its bytes and token words
            const bool in = (uint32_t)tid < nvalid;
            uint64_t tot;
            const uint64_t inc = wg_scan(in ? ((uint64_t)P.ntok << 32 | P.nbytes()) : 0ull, tot, 0);
            const uint32_t cb = (uint32_t)inc, ct = (uint32_t)(inc >> 32);
            const bool fits = in && outpos + cb <= raw_n;
            PROF_ADD(4);  // scan
            // (measured and dropped: counted parses that also record their tokens in per-thread scratch, the accepted parse then
            // copied instead of decoded once more -- the recording's stores cost more than the decode they save: 1000 files 216 k ->
            // 202 k images/s, 8000 files 300 k -> 282 k)
            bool far = false;
            if (fits) far = parse_tok<true>(L, P.start, limit, outpos + cb - P.nbytes(), tok + tpos + (ct - P.ntok)).err() && !P.err();
            PROF_ADD(5);  // emit
            // the last subsequence taken decides how the round ends
            const uint64_t fm = __ballot(fits);
            const bool is_end = fits && (lane == 63 ? true : !((fm >> (lane + 1)) & 1));      // last fitting lane of this wave
            if (__ballot(far) && lane == 0) L.bc[7] = 1;
            wg_sync();
            // the overall last fitting lane: the highest wave that has one (fits is a prefix)
            if (is_end) atomicMax(&L.bc[6], (uint32_t)tid + 1u);
            wg_sync();
            const uint32_t take = L.bc[6];
            if (take == 0) {                  // more output than the image has rows for
                bad = true;
                break;
            }
            if ((uint32_t)tid == take - 1u) {
                L.bc[8] = cb;
                L.bc[9] = ct;
                L.bc[10] = P.exit;
                L.bc[11] = (P.eob() ? 1u : 0u) | (P.err() ? 2u : 0u);
            }
            wg_sync();
            const uint32_t add = L.bc[8], ntk = L.bc[9], ex_bits = L.bc[10], endf = L.bc[11], farf = L.bc[7];
            wg_sync();
            if (farf) {
                bad = true;
                break;
            }
            outpos += add;
            tpos += ntk;
            bp = s0 + ex_bits;
            if ((endf & 2u) || bp > total_bits) bad = true;
            eob = endf & 1u;
            PROF_ADD(6);  // round end
        }
    }
    if (!bad && outpos != raw_n) bad = true;
#if defined(PNG_PROF) && defined(PNG_PROF_HUFF)      // (the LZ pass reports into the same slots: one of the two per build)
    if (tid == 0)
        for (int i = 0; i < 12; i++) atomicAdd(&g_png_prof[i], _acc[i]);
#endif
    if (tid == 0) {
        if (bad) info[img].status = UCFP_E_MODALITY;
        tinfo[img].ntok = tpos;
        tinfo[img].end_bit = bp;
    }
}

// One wave per file: token words -> filtered scanlines.  A lane takes 16 consecutive words of the round's 1024; a match
// belongs to the lane that holds its first word.
// round buffer + match list only (26 KB: six waves per CU where InflateLds' tables and stage left room for four)
struct LzLds {
    static constexpr uint32_t kIterOut = 8192, kMatchCap = 2048;
    // earlier output kept in LDS; sources older than that are fetched from frame memory a group of 64 matches ahead
    // (resolve_matches).  2 KB and 8 KB measured the same once that fetch was off the steps' critical path.
    static constexpr uint32_t kWin = UCFP_PNG_LZ_WINDOW;
    uint32_t m_dst[kMatchCap + 4];                        // (+ a spare slot for the places that list nothing)
    uint32_t m_ld[kMatchCap + 4];                         // len << 16 | (dist - 1)
    alignas(4) uint8_t rb[kWin + kIterOut + 8];
};

template <class C>
__global__ __launch_bounds__(64) void png_lz_kernel(const uint8_t* __restrict__ zbuf, const uint64_t* __restrict__ offsets, size_t n,
                                                   const UpItem* __restrict__ items, UpUniform uni, PngInfo* __restrict__ info,
                                                   const uint16_t* __restrict__ tokens, const PngTok* __restrict__ tinfo,
                                                   uint8_t* __restrict__ raw) {
    __shared__ C L;
    constexpr uint32_t kTPL = 32;      // token words per lane and round: 2048 words ~ 5 KB of output, ~1400 matches at level 1
    const size_t img = blockIdx.x;
    if (img >= n) return;
    const int lane = threadIdx.x;
    if (info[img].status != 0) return;
    const UpItem item = up_item(items, uni, img);
    const uint32_t raw_n = info[img].raw_n, ntok = tinfo[img].ntok;
    const uint16_t* tok = tokens + item.aux_off;      // (16-byte aligned: aux_off is a multiple of 16)
    uint8_t* out = raw + item.aux_off;
    uint32_t tpos = 0, outpos = 0;
    bool bad = false;
    Adler adler;
#ifdef PNG_PROF
    unsigned long long _acc[16] = {0};
#endif
    // a lane's words of a round: kTPL / 8 16-byte pieces (the token area is padded: reading past ntok stays inside it), the word
    // before them and the word behind them.  The NEXT round's are requested as soon as this round knows how many lanes it takes:
    // their round trip (5 k of a round's 54 k cycles) runs under the match resolution.
    struct Words {
        uint4 v[kTPL / 8];
        uint32_t before, after;
    };
    auto fetch_words = [&](uint32_t tp) {
        Words Wd;
        const uint32_t w0 = tp + (uint32_t)lane * kTPL;
        Wd.before = w0 > 0 && w0 <= ntok ? tok[w0 - 1] : 0u;
        const uint4* t4 = reinterpret_cast<const uint4*>(tok + w0);
#pragma unroll
        for (uint32_t q = 0; q < kTPL / 8; q++) {
            Wd.v[q] = make_uint4(0, 0, 0, 0);
            if (w0 + 8 * q < ntok) Wd.v[q] = t4[q];
        }
        Wd.after = w0 + kTPL < ntok ? tok[w0 + kTPL] : 0u;
        return Wd;
    };
    PROF_T0();
    Words Nx = fetch_words(0);
    while (tpos < ntok && !bad) {
        PROF_CNT(9, 1);
        // my words, and whether the first of them is the second half of a match that began before them
        const uint32_t w0 = tpos + (uint32_t)lane * kTPL;
        uint32_t wd[kTPL + 1];
        const Words Cur = Nx;
        const uint32_t before = Cur.before;
#pragma unroll
        for (uint32_t q = 0; q < kTPL / 8; q++) {
            const uint32_t x[4] = {Cur.v[q].x, Cur.v[q].y, Cur.v[q].z, Cur.v[q].w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                wd[8 * q + 2 * c] = w0 + 8 * q + 2 * c < ntok ? (x[c] & 0xffffu) : 0u;
                wd[8 * q + 2 * c + 1] = w0 + 8 * q + 2 * c + 1 < ntok ? (x[c] >> 16) : 0u;
            }
        }
        wd[kTPL] = Cur.after;
        // Matches that follow one another with the SAME distance are one longer match (byte by byte the copy is the same):
        // level-1 streams of photographs are chains of 3-byte matches one pixel back, each depending on the one before --
        // as separate list entries they resolve a handful per step, merged they are periodic runs the whole wave copies.
        uint32_t nb = 0, nm = 0, run_dist = 0;           // run_dist: distance word of the match the lane is extending (0: none)
        bool dist_next = (before & 0x8000u) != 0;
#pragma unroll
        for (uint32_t j = 0; j < kTPL; j++) {
            const bool have = w0 + j < ntok;
            const bool head = have && !dist_next && (wd[j] & 0x8000u);
            const bool lit = have && !dist_next && !(wd[j] & 0x8000u);
            const uint32_t dw = wd[j + 1] + 1u;           // (a head's distance word, + 1 so that 0 means "no run")
            nb += lit ? 1u : head ? (wd[j] & 0xffu) + 3u : 0u;
            nm += (head && dw != run_dist) ? 1u : 0u;
            run_dist = head ? dw : lit ? 0u : run_dist;
            dist_next = head;
        }
        const uint32_t cb = wave_incl_scan(nb, lane), cm = wave_incl_scan(nm, lane);
        const bool fits = cb <= C::kIterOut && cm <= C::kMatchCap && outpos + cb <= raw_n;
        const int take = __popcll(__ballot(fits));
        PROF_ADD(0);      // token words, counts, scans
        if (take == 0) {                      // (a lane's 16 words are at most 16 x 258 bytes: only the image's size can refuse them)
            bad = true;
            break;
        }
        Nx = fetch_words(tpos + (uint32_t)take * kTPL);
        const uint32_t rb_base = outpos > C::kWin ? (outpos - C::kWin) & ~3u : 0u;
        bool far = false;
        {
            // straight-line: every lane runs the 32 places; a place that writes nothing writes to a spare byte / spare list slot
            const bool act = lane < take;
            constexpr uint32_t kSpareB = sizeof(L.rb) - 4, kSpareM = C::kMatchCap;
            uint32_t p = outpos + cb - nb, mi = cm - nm, rd = 0, run_len = 0;
            bool dn = (before & 0x8000u) != 0;
#pragma unroll
            for (uint32_t j = 0; j < kTPL; j++) {
                const bool have = act && w0 + j < ntok;
                const bool head = have && !dn && (wd[j] & 0x8000u);
                const bool lit = have && !dn && !(wd[j] & 0x8000u);
                const uint32_t len = (wd[j] & 0xffu) + 3u, dw = wd[j + 1] + 1u, dist = (wd[j + 1] & 0x7fffu) + 1u;
                const bool isnew = head && dw != rd;            // a new entry (the run before it, if any, is complete)
                L.rb[lit ? p - rb_base : kSpareB] = (uint8_t)wd[j];
                far = far || (isnew && dist > p);
                L.m_dst[isnew ? mi : kSpareM] = p;
                mi += isnew ? 1u : 0u;
                run_len = isnew ? len : run_len + (head ? len : 0u);
                L.m_ld[head ? mi - 1u : kSpareM] = run_len << 16 | (dist - 1u);      // (a 32-word lane holds at most 16 x 258 bytes: 16 bits)
                rd = head ? dw : lit ? 0u : rd;
                p += head ? len : lit ? 1u : 0u;
                dn = head;
            }
        }
        wave_lds_sync();
        if (__ballot(far)) {
            bad = true;
            break;
        }
        const uint32_t add = __shfl(cb, take - 1, 64), nmt = __shfl(cm, take - 1, 64);
        PROF_ADD(1);      // literals and match list
        PROF_CNT(12, nmt);
#ifdef PNG_PROF
        resolve_matches(L, out, rb_base, nmt, lane, &_acc[13], &_acc[14]);
#else
        resolve_matches(L, out, rb_base, nmt, lane);
#endif
        PROF_ADD(2);      // matches      // (four groups of 64 entries per step measured slower: 1000 files 217 k -> 144 k images/s --
                                                           // the chains are real dependencies: 98 % of a level-1 photograph's bytes come from matches, so a source
                                                           // inside the round lies in an earlier match's destination)
        flush_out(L, out, outpos, outpos + add, rb_base, lane);
        adler.add(add, [&](uint32_t j) { return (uint32_t)L.rb[outpos + j - rb_base]; }, lane);
        __threadfence_block();
        outpos += add;
        PROF_ADD(3);      // flush + Adler
        {
            const uint32_t nb2 = outpos > C::kWin ? (outpos - C::kWin) & ~3u : 0u;
            const uint32_t shift = nb2 - rb_base, keep = (outpos - nb2 + 3) / 4;
            if (shift) {
                uint32_t wv[(C::kWin + 4) / 4 / 64 + 1];
#pragma unroll
                for (int i = 0; i < (int)((C::kWin + 4) / 4 / 64 + 1); i++) {
                    const uint32_t wi = lane + 64 * i;
                    wv[i] = wi < keep ? *reinterpret_cast<const uint32_t*>(&L.rb[shift + 4 * wi]) : 0u;
                }
                wave_lds_sync();
#pragma unroll
                for (int i = 0; i < (int)((C::kWin + 4) / 4 / 64 + 1); i++) {
                    const uint32_t wi = lane + 64 * i;
                    if (wi < keep) *reinterpret_cast<uint32_t*>(&L.rb[4 * wi]) = wv[i];
                }
                wave_lds_sync();
            }
        }
        PROF_ADD(4);      // window slide
        // words consumed: the taken lanes' (a match head in a lane's last word takes its distance from the next lane's first)
        tpos += (uint32_t)take * kTPL;
    }
    if (!bad && outpos != raw_n) bad = true;
    bool checksum_only = false;
    if (!bad) {
        const uint8_t* z = zbuf + ((offsets[item.file] + 15) & ~(uint64_t)15);
        const uint32_t zlen = info[img].zlen, e = (tinfo[img].end_bit + 7) / 8;
        if (e + 4 > zlen ||
            ((uint32_t)z[e] << 24 | (uint32_t)z[e + 1] << 16 | (uint32_t)z[e + 2] << 8 | z[e + 3]) != adler.value())
            checksum_only = true;
    }
#if defined(PNG_PROF) && !defined(PNG_PROF_HUFF)
    if (lane == 0)
        for (int i = 0; i < 16; i++) atomicAdd(&g_png_prof[i], _acc[i]);
#endif
    if (lane == 0) {
        if (bad) info[img].status = UCFP_E_MODALITY;
        else if (checksum_only) info[img].status = UCFP_IMAGE_NEEDS_HOST;
    }
}

// One wave per image: PNG 9.2 reconstruction.  Lane j takes rows j, j + 64, ...; it works one pixel behind lane j - 1
// (lane 0 one pixel behind lane 63's previous row, kept in LDS), so the pixel above is the neighbour's last output.
// A lane walks its row in blocks of 64 pixels, and the step loop only touches LDS:
//   in   the filtered bytes of a block (and, in front of block 0, the row's filter-type byte) are fetched as aligned
//        16-byte pieces TWO blocks ahead into registers and parked in the lane's LDS slot ONE block ahead, byte-shifted so
//        that pixel data starts on a word -- all lanes at the same steps, so the one wait per 64 steps is for loads
//        issued 64 steps earlier;
//   out  a reconstructed byte overwrites the filtered byte it came from; two blocks later (every lane is done with the
//        block by then) the 64 row segments leave as whole words, 256 contiguous bytes per store instruction.
template <int BPP>
struct UnfilterCfg {
    static constexpr int kPieces = (3 + 1 + 64 * BPP + 15 + 15) / 16;   // 3 pad + the byte before + 64 pixels + misalignment
    static constexpr int kWords = kPieces * 4;
    static constexpr int kSlot = kPieces * 16;
};
// BPP = bytes per pixel IN THE FILE; LAYOUT says what leaves: kLayoutPlain the same bytes, kLayoutPalette (BPP 1) three
// bytes per index through the file's PLTE, kLayoutGreyAlpha (BPP 2) the grey byte.  One launch per layout a batch may
// hold; a launch skips the files of the other layouts (and only the plain launch reports the status of rejected files).
template <int BPP, int LAYOUT>
__global__ __launch_bounds__(64) void png_unfilter_kernel(const uint8_t* __restrict__ raw, const UpItem* __restrict__ items,
                                                         UpUniform uni, uint32_t uprow_bytes /* widest row of the launch, 16-byte rounded */,
                                                         PngInfo* __restrict__ info, size_t n, uint8_t* __restrict__ frames,
                                                         int32_t* __restrict__ status, const uint8_t* __restrict__ png,
                                                         const uint64_t* __restrict__ offsets) {
    using Cfg = UnfilterCfg<BPP>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint8_t* slots = lds_raw;                               // [2][64 lanes][kSlot]; byte 3 = the byte before the block, byte 4.. = its pixels
    uint8_t* uprow = lds_raw + 2 * 64 * Cfg::kSlot;         // w * BPP bytes: the last row lane 63 finished
    uint8_t* plte = uprow + uprow_bytes;                    // kLayoutPalette: 256 x 3 bytes, black beyond the file's entries
    const size_t ent = blockIdx.x;
    if (ent >= n) return;
    const int lane = threadIdx.x;
    const UpItem item = up_item(items, uni, ent);
    const size_t img = item.file;
    const uint32_t w = item.w, h = item.h;
    const size_t row_stride = item.row_stride;
    const int32_t st = info[ent].status;
    if (st != 0) {
        if (LAYOUT == kLayoutPlain && lane == 0 && status) status[img] = st;
        return;
    }
    // a launch takes the files of its own layout and bytes per pixel (a ragged batch holds every kind)
    if (info[ent].layout != LAYOUT || info[ent].fbpp != BPP) return;
    if (LAYOUT == kLayoutPalette) {
        const uint8_t* pp = png + offsets[img] + info[ent].plte_off;
        const uint32_t pn = (uint32_t)info[ent].plte_n * 3;
        for (uint32_t i = lane; i < 768; i += 64) plte[i] = i < pn ? pp[i] : 0;
        wave_lds_fence();
    }
    const uint8_t* src = raw + item.aux_off;
    uint8_t* dst = frames + item.frame_off;
    const uint32_t rowb = w * BPP;
    const uint32_t W = ((w > 64 ? w : 64) + 63) & ~63u;       // steps per row: whole blocks, and lane 63 is done with x before lane 0 needs it
    const uint32_t bpr = W / 64;                              // blocks per row
    const uint32_t rounds = (h + 63) / 64;
    const uint32_t nblocks = rounds * bpr;                    // lane-local blocks; the stagger adds one global block
    bool bad = false;
    uint32_t a[BPP], b[BPP], prev_out[BPP];                    // left, above (= the next step's above-left), this lane's last output
#pragma unroll
    for (int ch = 0; ch < BPP; ch++) a[ch] = b[ch] = prev_out[ch] = 0;
    uint32_t ft = 0;
    uint32_t pre[Cfg::kWords + 1];                             // the block fetched ahead (+ one zero word for the funnel shift)
    uint32_t pre_mis = 0;
    auto fetch = [&](uint32_t bl) {                            // lane-local block bl -> registers
#pragma unroll
        for (int i = 0; i <= Cfg::kWords; i++) pre[i] = 0;
        pre_mis = 0;
        if (bl >= nblocks) return;
        const uint32_t row = (bl / bpr) * 64 + lane, x0 = (bl % bpr) * 64;
        if (row >= h || x0 >= w) return;
        const uint8_t* rs = src + (size_t)row * (rowb + 1);
        const uintptr_t addr = reinterpret_cast<uintptr_t>(rs) + (size_t)x0 * BPP - 3;  // lands the byte before the block at slot byte 3
        const uintptr_t base = addr & ~(uintptr_t)15, end = reinterpret_cast<uintptr_t>(rs) + rowb + 1;
        pre_mis = (uint32_t)(addr & 15);
#pragma unroll
        for (int p = 0; p < Cfg::kPieces; p++)
            if (base + 16 * p < end) {                         // (the first piece of row 0 may start in the 16 bytes before the image: workspace)
                const uint4 v = *reinterpret_cast<const uint4*>(base + 16 * p);
                pre[4 * p] = v.x, pre[4 * p + 1] = v.y, pre[4 * p + 2] = v.z, pre[4 * p + 3] = v.w;
            }
    };
    const bool words_ok = (row_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0;
    auto flush_block = [&](uint32_t bl) {
        const uint32_t k = bl / bpr, x0 = (bl % bpr) * 64;
        if (x0 >= w) return;
        const uint32_t px = w - x0 < 64 ? w - x0 : 64, len = px * BPP;         // bytes per row segment
        const uint8_t* t0 = slots + (size_t)(bl & 1) * 64 * Cfg::kSlot + 4;
        if (LAYOUT != kLayoutPlain) {
            // lane = pixel of the 64-pixel segment: a palette index becomes its three PLTE bytes, a grey + alpha pair its grey
            const uint32_t rows = h - k * 64 < 64 ? h - k * 64 : 64;
            for (uint32_t r = 0; r < rows; r++) {
                const uint8_t* t = t0 + (size_t)r * Cfg::kSlot;
                if ((uint32_t)lane < px) {
                    if (LAYOUT == kLayoutPalette) {
                        uint8_t* d = dst + (size_t)(k * 64 + r) * row_stride + (size_t)(x0 + lane) * 3;
                        const uint32_t e = (uint32_t)t[lane] * 3;
                        d[0] = plte[e], d[1] = plte[e + 1], d[2] = plte[e + 2];
                    } else {
                        dst[(size_t)(k * 64 + r) * row_stride + x0 + lane] = t[2 * lane];
                    }
                }
            }
            return;
        }
        const uint32_t wpr = words_ok ? len / 4 : 0;                           // <= 64: one word per lane and row
        const uint32_t rows = h - k * 64 < 64 ? h - k * 64 : 64;
        for (uint32_t r = 0; r < rows; r++) {
            uint8_t* d = dst + (size_t)(k * 64 + r) * row_stride + (size_t)x0 * BPP;
            const uint8_t* t = t0 + (size_t)r * Cfg::kSlot;
            if ((uint32_t)lane < wpr) *reinterpret_cast<uint32_t*>(d + 4 * lane) = *reinterpret_cast<const uint32_t*>(t + 4 * lane);
            for (uint32_t bi = 4 * wpr + lane; bi < len; bi += 64) d[bi] = t[bi];   // bytes the words did not cover
        }
    };
    // the lane's own position (it starts `lane` steps late): block in its row, row group, slot parity
    uint32_t xb = 0, kk = 0, par = 0, done_blocks = 0;
    fetch(0);
    for (uint32_t m = 0; m <= nblocks; m++) {
        // block m - 2 is complete in every lane (lane 63 finished it 2 steps ago); its slot is the one block m parks in
        if (m >= 2) flush_block(m - 2);
        {
            uint32_t* slot = reinterpret_cast<uint32_t*>(slots + ((m & 1) * 64 + lane) * Cfg::kSlot);
            const uint32_t bsh = pre_mis & 3, dsh = pre_mis >> 2;
#pragma unroll
            for (int i = 0; i < Cfg::kWords; i++) {
                const uint32_t v = __builtin_amdgcn_alignbyte(pre[i + 1], pre[i], bsh);
                if ((uint32_t)i >= dsh) slot[i - dsh] = v;
            }
            fetch(m + 1);
        }
        wave_lds_fence();
        for (uint32_t s = 0; s < 64; s++) {
            const uint32_t t = m * 64 + s;
            // what the lane above produced in the previous step (lane 0: lane 63's stored row)
            uint32_t up[BPP];
#pragma unroll
            for (int ch = 0; ch < BPP; ch++)      // wave_shr:1 on the vector pipe (__shfl_up is a ds_bpermute: an LDS round trip, and the compiler waits for each)
                up[ch] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)prev_out[ch], 0x138, 0xf, 0xf, false);
            const int tt = (int)t - lane;
            const bool live = tt >= 0;
            const uint32_t xi = (uint32_t)tt & 63u;
            const uint32_t x = xb * 64 + xi;
            const uint32_t row = kk * 64 + lane;
            const bool on = live && done_blocks < nblocks && row < h && x < w;
            // Straight-line from here: every LDS read of the step is issued at once (idle lanes read their own slot's
            // start), the four predictors are computed side by side and picked by the filter type with selects.  The
            // branchy version spent the step in exec-mask bookkeeping and six serial LDS round trips.
            uint8_t* sp = slots + (par * 64 + lane) * Cfg::kSlot + 3;               // sp[0]: the byte before the block
            const uint32_t xr = on ? xi : 0, xu = on ? x : 0;
            const uint32_t ftb = sp[0];
            uint32_t v[BPP], ur[BPP];
#pragma unroll
            for (int ch = 0; ch < BPP; ch++) {
                v[ch] = sp[1 + xr * BPP + ch];
                ur[ch] = uprow[xu * BPP + ch];
            }
            const bool first = x == 0;
            ft = (on && first) ? ftb : ft;
            if (on && first && ftb > 4) bad = true;
            uint32_t o[BPP];
#pragma unroll
            for (int ch = 0; ch < BPP; ch++) {
                const int A = first ? 0 : (int)a[ch];
                const int C = first ? 0 : (int)b[ch];                               // last step's "above" is this step's "above-left"
                const int Bv = row == 0 ? 0 : (lane == 0 ? (int)ur[ch] : (int)up[ch]);
                const int pp = A + Bv - C;
                const int pa = pp > A ? pp - A : A - pp, pb = pp > Bv ? pp - Bv : Bv - pp, pc = pp > C ? pp - C : C - pp;
                const int paeth = (pa <= pb && pa <= pc) ? A : (pb <= pc ? Bv : C);
                int pred = 0;
                pred = ft == 1 ? A : pred;
                pred = ft == 2 ? Bv : pred;
                pred = ft == 3 ? (A + Bv) >> 1 : pred;
                pred = ft == 4 ? paeth : pred;
                o[ch] = ((uint32_t)v[ch] + (uint32_t)pred) & 255u;
                if (on) {
                    b[ch] = (uint32_t)Bv;
                    a[ch] = prev_out[ch] = o[ch];
                }
            }
            if (on) {
#pragma unroll
                for (int ch = 0; ch < BPP; ch++) sp[1 + xi * BPP + ch] = (uint8_t)o[ch];   // in place: the filtered byte is spent
                if (lane == 63) {
#pragma unroll
                    for (int ch = 0; ch < BPP; ch++) uprow[x * BPP + ch] = (uint8_t)o[ch];
                }
            }
            if (live && xi == 63) {                               // on to the lane's next block
                done_blocks++;
                par ^= 1;
                if (++xb == bpr) xb = 0, kk++;
            }
            wave_lds_fence();
        }
    }
    if (nblocks >= 1) flush_block(nblocks - 1);               // the loop flushed blocks 0 .. nblocks - 2
    const bool any_bad = __ballot(bad) != 0;
    if (lane == 0) {
        if (any_bad) info[ent].status = UCFP_E_MODALITY;
        if (status) status[img] = any_bad ? UCFP_E_MODALITY : 0;
    }
}

// Records of files that did not decode are zeroed and carry the decoder's status.
__global__ void png_merge_status_kernel(const PngInfo* __restrict__ info, const UpItem* __restrict__ items, size_t n,
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

int launch_png_merge_status(const uint8_t* ws, const PngWs& l, size_t n, uint8_t* out, uint32_t rec, int32_t* status,
                            hipStream_t stream, const UpItem* d_items) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(png_merge_status_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream,
                       reinterpret_cast<const PngInfo*>(ws + l.info), d_items, n, out, rec, status);
    return 0;
}

#ifdef PNG_PROF
extern "C" int ucfp_debug_png_prof(unsigned long long* out16, int reset) {
    if (out16) (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_png_prof), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_png_prof), z, sizeof z);
    }
    return 0;
}
#endif

size_t png_raw_bytes(uint32_t w, uint32_t h, int pixfmt) {
    const size_t bpp = pixfmt == UCFP_PIX_GRAY8 ? 2 /* a grey + alpha file */ : pixfmt == UCFP_PIX_RGB8 ? 3 : 4;
    const size_t raw_n = (size_t)h * ((size_t)w * bpp + 1);
    return (raw_n + 15 + 64) & ~(size_t)15;
}

static size_t png_ws_layout(size_t n, size_t png_bytes, size_t raw_total, PngWs* l) {
    size_t off = 0;
    l->zbuf = off;
    off += (png_bytes + 16 + 64 + 255) & ~(size_t)255;
    l->info = off;
    off += (n * sizeof(PngInfo) + 255) & ~(size_t)255;
    l->raw = off;
    off += raw_total;
    l->tok = off;
    off += 2 * raw_total;
    l->tinfo = off;
    off += (n * sizeof(PngTok) + 255) & ~(size_t)255;
    l->total = off;
    return off;
}

size_t png_ws_bytes(size_t n, size_t png_bytes, uint32_t w, uint32_t h, int pixfmt, PngWs* ws) {
    PngWs l;
    l.raw_stride = png_raw_bytes(w, h, pixfmt);
    l.raw_n = l.raw_stride;
    const size_t off = png_ws_layout(n, png_bytes, n * l.raw_stride, &l);
    if (ws) *ws = l;
    return off;
}

size_t png_ragged_ws_bytes(size_t n, size_t png_bytes, size_t raw_total, PngWs* ws) {
    PngWs l;
    const size_t off = png_ws_layout(n, png_bytes, raw_total, &l);
    if (ws) *ws = l;
    return off;
}

// max_w[f]: widest frame (pixels) announced as pixel format f among the launch's files (0: no such file)
static int png_decode_launches(const uint8_t* png, const uint64_t* offsets, const UpItem* d_items, const UpUniform& uni, size_t n,
                               const uint32_t max_w[3], uint8_t* ws, const PngWs& l, uint8_t* frames, int32_t* status,
                               hipStream_t stream) {
    PngInfo* info = reinterpret_cast<PngInfo*>(ws + l.info);
    hipLaunchKernelGGL(png_scan_kernel, dim3((unsigned)n), dim3(64), 0, stream, png, offsets, n, d_items, uni, ws + l.zbuf, info);
    // Two-pass inflate (png_huff_kernel + png_lz_kernel): waves per file by how many files there are to fill the chip with
    // measured, 256-bit subsequences, images/s (one-kernel inflate | 1 | 2 | 4 | 8 waves per file):  64 files 12.7 k | - | 13.8 | 16.2 | 11.9;
    // 256: 49 | - | 55 | 62 | 45;  500: 92 | - | 100 | 112 | 78;  1000: 160 | - | 187 | 206 | 135;  2000: 163 | - | 237 | 218 | 161;
    // 3000: 191 | - | 228 | 227 | 174;  8000: 213 | 263 | 255 | 241 | 194 (128-bit subsequences need more re-parses: 151 k at
    // 1000 x 4; 512-bit ones are level with 256)
    static const char* two = getenv("UCFP_PNG_TWO_PASS");      // (A/B: 0 = the one-kernel inflate; 2 / 4 / 8 = that many waves per file)
    if (!two || atoi(two) != 0) {
        uint16_t* tokens = reinterpret_cast<uint16_t*>(ws + l.tok);
        PngTok* tinfo = reinterpret_cast<PngTok*>(ws + l.tinfo);
        const int force_w = two && atoi(two) > 1 ? atoi(two) : 0;
        const int w = force_w ? force_w : n <= 1200 ? 4 : n <= 3000 ? 2 : 1;
        auto huff = [&](auto kern, int waves) {
            static const char* wm = getenv("UCFP_PNG_HUFF_WARM");          // (tuning)
            static const char* mi = getenv("UCFP_PNG_HUFF_MAX_ITER");
            hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(64 * waves), 0, stream, ws + l.zbuf, offsets, n, d_items, uni, info, tokens, tinfo,
                               wm ? (uint32_t)atoi(wm) : 1u, mi ? (uint32_t)atoi(mi) : 3u);      // (1000 files x 4 waves: 150 k images/s without the
                               // warm-up parse, 210 k with; chain iterations capped at 2 / 3 / 4 / none: 200 / 217 / 215 / 210 k)
        };
        static const char* hb = getenv("UCFP_PNG_HUFF_BITS");      // (tuning)
        const int bits = hb ? atoi(hb) : 256;
        if (bits >= 512) {
            if (w >= 8) huff(png_huff_kernel<512, 8>, 8);
            else if (w >= 4) huff(png_huff_kernel<512, 4>, 4);
            else if (w >= 2) huff(png_huff_kernel<512, 2>, 2);
            else huff(png_huff_kernel<512, 1>, 1);
        } else if (bits >= 256) {
            if (w >= 8) huff(png_huff_kernel<256, 8>, 8);
            else if (w >= 4) huff(png_huff_kernel<256, 4>, 4);
            else if (w >= 2) huff(png_huff_kernel<256, 2>, 2);
            else huff(png_huff_kernel<256, 1>, 1);
        } else {
            if (w >= 8) huff(png_huff_kernel<128, 8>, 8);
            else if (w >= 4) huff(png_huff_kernel<128, 4>, 4);
            else if (w >= 2) huff(png_huff_kernel<128, 2>, 2);
            else huff(png_huff_kernel<128, 1>, 1);
        }
        hipLaunchKernelGGL(png_lz_kernel<LzLds>, dim3((unsigned)n), dim3(64), 0, stream, ws + l.zbuf, offsets, n, d_items, uni, info,
                           (const uint16_t*)tokens, (const PngTok*)tinfo, ws + l.raw);
    } else
    // wide rounds while the batch leaves most of the chip's wave slots empty anyway (RoundCfg)
    // measured (files/s, narrow | wide): 600: 83 k | 98 k, 1000: 130 k | 157 k, 1400: 162 k | 118 k, 2000: 141 k | 160 k,
    // 3000: 189 k | 164 k -- the chip holds about 1024 wide or 1536 narrow waves at a time
    if (n <= 512)           // two waves per CU hold such a batch: 512-bit rounds, 10 % less latency than 256-bit ones
        hipLaunchKernelGGL(png_inflate_kernel<RoundCfg<512>>, dim3((unsigned)n), dim3(64), 0, stream, ws + l.zbuf, offsets, n, d_items, uni,
                           info, ws + l.raw);
    else if (n <= 1024 || (n > 1536 && n <= 2048))
        hipLaunchKernelGGL(png_inflate_kernel<RoundCfg<256>>, dim3((unsigned)n), dim3(64), 0, stream, ws + l.zbuf, offsets, n, d_items, uni,
                           info, ws + l.raw);
    else
        hipLaunchKernelGGL(png_inflate_kernel<RoundCfg<128>>, dim3((unsigned)n), dim3(64), 0, stream, ws + l.zbuf, offsets, n, d_items, uni,
                           info, ws + l.raw);
    // one unfilter launch per file layout the announced formats admit: GRAY8 <- grey | grey + alpha, RGB8 <- RGB | palette
    auto go = [&](auto kern, size_t fbpp, bool palette, uint32_t wmax) {
        const uint32_t uprow = (uint32_t)(((size_t)wmax * fbpp + 15) & ~(size_t)15);
        const size_t lds = uprow + (palette ? 768 : 0) + 2 * 64 * (size_t)((3 + 1 + 64 * fbpp + 15 + 15) / 16) * 16;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(64), lds, stream, ws + l.raw, d_items, uni, uprow, info, n, frames, status, png,
                           offsets);
    };
    if (max_w[UCFP_PIX_GRAY8]) {
        go(png_unfilter_kernel<1, kLayoutPlain>, 1, false, max_w[UCFP_PIX_GRAY8]);
        go(png_unfilter_kernel<2, kLayoutGreyAlpha>, 2, false, max_w[UCFP_PIX_GRAY8]);
    }
    if (max_w[UCFP_PIX_RGB8]) {
        go(png_unfilter_kernel<3, kLayoutPlain>, 3, false, max_w[UCFP_PIX_RGB8]);
        go(png_unfilter_kernel<1, kLayoutPalette>, 1, true, max_w[UCFP_PIX_RGB8]);
    }
    if (max_w[UCFP_PIX_RGBA8]) go(png_unfilter_kernel<4, kLayoutPlain>, 4, false, max_w[UCFP_PIX_RGBA8]);
    return 0;
}

int launch_png_decode(const uint8_t* png, const uint64_t* offsets, size_t n, uint32_t w, uint32_t h, int pixfmt, uint8_t* ws,
                      const PngWs& l, uint8_t* frames, size_t row_stride, size_t frame_stride, int32_t* status,
                      hipStream_t stream, hipStream_t side, hipEvent_t fork, hipEvent_t join) {
    if (n == 0) return 0;
    UpUniform uni;
    uni.w = w;
    uni.h = h;
    uni.pixfmt = pixfmt;
    uni.row_stride = (uint32_t)row_stride;
    uni.frame_stride = frame_stride;
    uni.aux_stride = l.raw_stride;
    uint32_t max_w[3] = {0, 0, 0};
    max_w[pixfmt] = w;
    static const bool no_split = getenv("UCFP_PNG_NO_SPLIT") != nullptr;      // (A/B)
    if (side && fork && join && n >= 1600 && !no_split) {      // (measured: 256 / 500 / 1000 files 8 / 3 / 4 % slower in halves, 2000 / 8000 4-5 % faster)
        // two halves side by side: file i of the second half is file h + i of every per-file array (the gathered streams are
        // addressed by the files' own byte offsets, so that area is shared as it is)
        const size_t hn = n / 2;
        PngWs l2 = l;
        l2.info += hn * sizeof(PngInfo);
        l2.raw += hn * l.raw_stride;
        l2.tok += 2 * hn * l.raw_stride;
        l2.tinfo += hn * sizeof(PngTok);
        (void)hipEventRecord(fork, stream);
        (void)hipStreamWaitEvent(side, fork, 0);
        png_decode_launches(png, offsets, nullptr, uni, hn, max_w, ws, l, frames, status, stream);
        png_decode_launches(png, offsets + hn, nullptr, uni, n - hn, max_w, ws, l2, frames + hn * frame_stride, status ? status + hn : nullptr,
                            side);
        (void)hipEventRecord(join, side);
        (void)hipStreamWaitEvent(stream, join, 0);
        return 0;
    }
    return png_decode_launches(png, offsets, nullptr, uni, n, max_w, ws, l, frames, status, stream);
}

int launch_png_decode_ragged(const uint8_t* png, const uint64_t* offsets, const UpItem* d_items, size_t n, const uint32_t max_w[3],
                             uint8_t* ws, const PngWs& l, uint8_t* frames, int32_t* status, hipStream_t stream) {
    if (n == 0) return 0;
    return png_decode_launches(png, offsets, d_items, UpUniform{}, n, max_w, ws, l, frames, status, stream);
}

}  // namespace ucfp
