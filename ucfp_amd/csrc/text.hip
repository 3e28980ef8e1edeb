// text.hip -- batched MinHash-128 and SimHash-64 for gfx950.
//
// Replaces the arithmetic behind text::fingerprint_minhash_with::<128> (src/modality/text.rs:182-236)
// and simhash_dispatch (text.rs:366-421) of the reference, i.e. txtfp's MinHashFingerprinter /
// SimHashFingerprinter, for documents that are ASCII (canonicalisation = lower-casing, UAX#29 word
// segmentation restricted to ASCII, both done here on the GPU) or that the host has already
// canonicalised and tokenised (PRETOKENIZED: tokens separated by single spaces).  Spec: DESIGN.md
// "Text spec" T1..T6; CPU statement: oracle/ (text).
//
// One 256-thread workgroup per document, tiles of <= 2047 bytes:
//   A  load the tile (16 B/lane, coalesced) into LDS, classify bytes; "byte is inside a word" is a
//      function of (prev, cur, next) only (WB5-13 on ASCII), so it is embarrassingly parallel
//   B  two block-wide exclusive scans (word bytes, token starts) give every word byte its place in
//      the CANONICAL STREAM  tok0 ' ' tok1 ' ' tok2 ...  ; a k-shingle is then one contiguous
//      byte range of that stream
//   C  thread = shingle (MinHash) or token (SimHash): XXH3_64 of its byte range, straight from LDS
//   D  MinHash: lane = 2 of the 128 slots; every shingle's (h1, h2) is broadcast from LDS and each
//      lane keeps running minima of h1 + i*h2 for its slots i and i + 64 (a wave = all 128
//      permutations; the 4 waves split the shingles and are min-combined at the end).
//      SimHash: lane = output bit; each token hash is broadcast and lane b counts bit b.
// Long documents loop over tiles cut at token boundaries; MinHash tiles overlap by k-1 tokens (min
// is idempotent), SimHash tiles do not.
// ALU-bound: ~17 integer ops per shingle per lane in D; HBM traffic is 4 KiB + 1 KiB per document.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ucfp_xxh3.h"
#include "common.h"

namespace ucfp {

namespace {

constexpr int kTile = 2048;           // LDS bytes per tile (last byte is look-ahead)
constexpr int kMaxTok = kTile / 2 + 8;
constexpr int kPer = kTile / 256;      // bytes per thread per tile

struct TextLds {
    uint8_t raw[kTile + 16];
    uint8_t canon[kTile + 16];
    uint16_t cstart[kMaxTok];
    uint16_t cend[kMaxTok];
    uint64_t h1[kMaxTok];
    uint64_t h2[kMaxTok];
    uint32_t scan_in[8], scan_st[8];   // per-wave totals for the block scans
    uint64_t wave_min[4][128];
    uint32_t wave_ones[4][64];
    int32_t cut;
    uint32_t next_raw;
    uint32_t flags;                    // bit0: non-ASCII byte seen, bit1: token longer than a tile
};

enum { C_O = 0, C_L = 1, C_N = 2, C_ML = 3, C_MNL = 4, C_MN = 5 };

__device__ __forceinline__ int cls(uint8_t c) {
    if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_') return C_L;
    if (c >= '0' && c <= '9') return C_N;
    if (c == ':') return C_ML;
    if (c == '.' || c == '\'') return C_MNL;
    if (c == ',' || c == ';') return C_MN;
    return C_O;
}

// prev / next are raw bytes (0 outside the tile)
__device__ __forceinline__ bool inword(uint8_t p, uint8_t c, uint8_t q, bool pretok) {
    if (pretok) return c != ' ' && c != 0;
    const int cc = cls(c);
    if (cc == C_L || cc == C_N) return true;
    if (cc == C_O) return false;
    const int pc = cls(p), qc = cls(q);
    if (pc == C_L && qc == C_L && (cc == C_ML || cc == C_MNL)) return true;
    if (pc == C_N && qc == C_N && (cc == C_MN || cc == C_MNL)) return true;
    return false;
}

#define UCFP_RD8_LDS(p, i) ((p)[(i)])
UCFP_XXH3_DEFINE(xxh3_lds, const uint8_t*, UCFP_RD8_LDS)

__device__ __forceinline__ uint64_t mix_h2(uint64_t h1) {
    uint64_t z = h1 + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (z ^ (z >> 31)) | 1ull;
}

// block-wide exclusive scan of two counters at once (256 threads)
__device__ __forceinline__ void block_scan2(TextLds& L, uint32_t a, uint32_t b, uint32_t& ea, uint32_t& eb,
                                            uint32_t& ta, uint32_t& tb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t ia = a, ib = b;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t oa = __shfl_up(ia, off, 64), ob = __shfl_up(ib, off, 64);
        if (lane >= off) {
            ia += oa;
            ib += ob;
        }
    }
    if (lane == 63) {
        L.scan_in[wave] = ia;
        L.scan_st[wave] = ib;
    }
    __syncthreads();
    uint32_t ba = 0, bb = 0;
    ta = 0;
    tb = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t xa = L.scan_in[w], xb = L.scan_st[w];
        if (w < wave) {
            ba += xa;
            bb += xb;
        }
        ta += xa;
        tb += xb;
    }
    ea = ba + ia - a;
    eb = bb + ib - b;
    __syncthreads();
}

}  // namespace

// MODE_SIM = false: MinHash (out 1032 B/doc); true: SimHash (out 8 B/doc)
template <bool MODE_SIM>
__global__ __launch_bounds__(256, 4) void text_hash_kernel(const uint8_t* __restrict__ utf8,
                                                        const uint64_t* __restrict__ offsets, size_t n,
                                                        int pretok_i, uint32_t k,
                                                        uint8_t* __restrict__ out, int32_t* __restrict__ status) {
    __shared__ TextLds L;
    const size_t doc = blockIdx.x;
    if (doc >= n) return;
    const bool pretok = pretok_i != 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t* __restrict__ text = utf8 + offsets[doc];
    const size_t len = (size_t)(offsets[doc + 1] - offsets[doc]);

    uint64_t m0 = ~0ull, m1 = ~0ull;  // MinHash running minima: slots lane, lane + 64
    uint32_t ones = 0;                // SimHash: count of bit `lane`
    uint32_t total_tok = 0;
    bool first_tile = true;
    if (tid == 0) L.flags = 0;
    __syncthreads();

    size_t pos = 0;
    while (pos < len) {
        const size_t remain = len - pos;
        const bool last = remain <= (size_t)(kTile - 1);
        const int tl = last ? (int)remain : kTile - 1;
        // ---- A: load tl (+1 look-ahead) bytes ----
        {
            const int nload = last ? tl : tl + 1;
            const uint8_t* src = text + pos;
            const int b0 = tid * kPer;
            if (((reinterpret_cast<uintptr_t>(src) & 7u) == 0) && b0 + kPer <= nload) {
                *reinterpret_cast<uint2*>(&L.raw[b0]) = *reinterpret_cast<const uint2*>(src + b0);
            } else {
#pragma unroll
                for (int j = 0; j < kPer; j++) L.raw[b0 + j] = (b0 + j < nload) ? src[b0 + j] : 0;
            }
            if (tid == 0) {
                L.cut = -1;
                L.next_raw = 0;
            }
        }
        __syncthreads();
        uint8_t c[kPer + 2];  // c[0] = byte before my kPer, c[kPer + 1] = byte after
        {
            const int b0 = tid * kPer;
            c[0] = b0 > 0 ? L.raw[b0 - 1] : 0;
#pragma unroll
            for (int j = 0; j < kPer; j++) c[j + 1] = L.raw[b0 + j];
            c[kPer + 1] = (b0 + kPer < kTile) ? L.raw[b0 + kPer] : 0;
        }
        uint32_t inw = 0;  // bit j: byte b0 + j is a word byte
        bool nonascii = false;
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int i = tid * kPer + j;
            if (i < tl) {
                if (!pretok && c[j + 1] >= 0x80) nonascii = true;
                if (inword(c[j], c[j + 1], c[j + 2], pretok)) inw |= 1u << j;
            }
        }
        if (nonascii) atomicOr(&L.flags, 1u);
        // ---- cut the tile at its last non-word byte when more text follows ----
        int limit = tl;
        if (!last) {
            int mycut = -1;
#pragma unroll
            for (int j = 0; j < kPer; j++) {
                const int i = tid * kPer + j;
                if (i < tl && !((inw >> j) & 1u)) mycut = i;
            }
            if (mycut >= 0) atomicMax(&L.cut, mycut);
            __syncthreads();
            limit = L.cut;
            if (limit < 0) {  // one token fills the whole tile
                if (tid == 0) atomicOr(&L.flags, 2u);
                __syncthreads();
                break;
            }
        }
        // word bytes at or beyond the cut belong to the next tile
        uint32_t st = 0;  // bit j: token starts at byte b0 + j
        uint32_t nin = 0, nst = 0;
        bool prev_in = false;
        {
            const int b0 = tid * kPer;
            // previous byte's status (recomputed: needs the byte before it)
            if (b0 > 0 && b0 - 1 < limit) {
                const uint8_t pp = b0 > 1 ? L.raw[b0 - 2] : 0;
                prev_in = inword(pp, c[0], c[1], pretok);
            }
        }
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int i = tid * kPer + j;
            bool w = ((inw >> j) & 1u) && i < limit;
            if (!w) inw &= ~(1u << j);
            if (w) {
                nin++;
                if (!prev_in) {
                    st |= 1u << j;
                    nst++;
                }
            }
            prev_in = w;
        }
        uint32_t base_in, base_st, tot_in, nt;
        block_scan2(L, nin, nst, base_in, base_st, tot_in, nt);
        // ---- B: canonical stream ----
        {
            uint32_t li = 0, ls = 0;
#pragma unroll
            for (int j = 0; j < kPer; j++) {
                if (!((inw >> j) & 1u)) continue;
                const int i = tid * kPer + j;
                if ((st >> j) & 1u) ls++;
                const uint32_t tok = base_st + ls - 1;
                const uint32_t cpos = base_in + li + tok;
                uint8_t ch = c[j + 1];
                if (!pretok && ch >= 'A' && ch <= 'Z') ch = (uint8_t)(ch + 32);
                L.canon[cpos] = ch;
                if ((st >> j) & 1u) {
                    L.cstart[tok] = (uint16_t)cpos;
                    if (cpos > 0) L.canon[cpos - 1] = ' ';
                    if (!MODE_SIM && !last && nt >= k && tok == nt - (k - 1)) L.next_raw = (uint32_t)i;
                }
                // token ends here if the next byte is not a word byte (or lies beyond the limit)
                bool next_in;
                if (j < kPer - 1) next_in = (inw >> (j + 1)) & 1u;
                else next_in = (i + 1 < limit) && inword(c[kPer], c[kPer + 1], (i + 2 < kTile) ? L.raw[i + 2] : 0, pretok);
                if (!next_in) L.cend[tok] = (uint16_t)(cpos + 1);
                li++;
            }
        }
        __syncthreads();
        // ---- C: hash shingles / tokens ----
        uint32_t nitems;
        if (MODE_SIM) nitems = nt;
        else nitems = nt >= k ? nt - k + 1 : ((first_tile && last && nt > 0) ? 1u : 0u);
        for (uint32_t s = tid; s < nitems; s += 256) {
            uint32_t e;
            if (MODE_SIM) e = s;
            else e = nt >= k ? s + k - 1 : nt - 1;
            const uint32_t a = L.cstart[s], b = L.cend[e];
            const uint64_t h = xxh3_lds(L.canon + a, (size_t)(b - a));
            L.h1[s] = h;
            if (!MODE_SIM) L.h2[s] = mix_h2(h);
        }
        __syncthreads();
        // ---- D: slot minima / bit counts ----
        if (MODE_SIM) {
            for (uint32_t s = wave; s < nitems; s += 4) ones += (uint32_t)((L.h1[s] >> lane) & 1ull);
        } else {
            for (uint32_t s = wave; s < nitems; s += 4) {
                const uint64_t h = L.h1[s], g = L.h2[s];
                const uint64_t v0 = h + (uint64_t)lane * g;
                const uint64_t v1 = v0 + (g << 6);
                m0 = v0 < m0 ? v0 : m0;
                m1 = v1 < m1 ? v1 : m1;
            }
        }
        total_tok += nt;
        first_tile = false;
        if (last) break;
        // ---- advance ----
        if (MODE_SIM) {
            pos += (size_t)limit + 1;
        } else {
            if (nt < k) {  // cannot carry k-1 tokens into the next tile
                if (tid == 0) atomicOr(&L.flags, 2u);
                __syncthreads();
                break;
            }
            pos += L.next_raw;
            total_tok -= (k - 1);  // the overlap is counted again by the next tile
        }
        __syncthreads();
    }
    // ---- combine the four waves and emit ----
    if (MODE_SIM) L.wave_ones[wave][lane] = ones;
    else {
        L.wave_min[wave][lane] = m0;
        L.wave_min[wave][lane + 64] = m1;
    }
    __syncthreads();
    const uint32_t flags = L.flags;
    int32_t stv = 0;
    if (flags & 1u) stv = 1;             // non-ASCII in raw mode: host must pre-tokenise
    else if (flags & 2u) stv = -2;       // UCFP_E_UNSUPPORTED: token / token run longer than a tile
    else if (total_tok == 0 || (!MODE_SIM && k == 0)) stv = -1;  // UCFP_E_MODALITY: no tokens
    if (MODE_SIM) {
        if (wave == 0) {
            const uint32_t o = L.wave_ones[0][lane] + L.wave_ones[1][lane] + L.wave_ones[2][lane] +
                               L.wave_ones[3][lane];
            const uint64_t bits = __ballot(2u * o > total_tok);
            if (lane == 0) {
                const uint64_t v = stv == 0 ? bits : 0ull;
                uint8_t* o8 = out + doc * 8;
                for (int b = 0; b < 8; b++) o8[b] = (uint8_t)(v >> (8 * b));
            }
        }
    } else {
        uint8_t* rec = out + doc * 1032;
        if (tid < 128) {
            uint64_t v = L.wave_min[0][tid];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                const uint64_t x = L.wave_min[w][tid];
                v = x < v ? x : v;
            }
            if (stv != 0) v = 0;
            // 1032-byte records are only 8-byte aligned when the base is: write two dwords
            uint32_t* o32 = reinterpret_cast<uint32_t*>(rec + 8 + 8 * tid);
            o32[0] = (uint32_t)v;
            o32[1] = (uint32_t)(v >> 32);
        } else if (tid == 128) {
            uint32_t* o32 = reinterpret_cast<uint32_t*>(rec);
            o32[0] = stv == 0 ? 1u : 0u;  // schema: u16 = 1, pad
            o32[1] = 0;
        }
    }
    if (status && tid == 0) status[doc] = stv;
}

int launch_text_minhash(const uint8_t* utf8, const uint64_t* offsets, size_t n, int mode, uint32_t k,
                        uint8_t* out, int32_t* status, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(text_hash_kernel<false>, dim3((unsigned)n), dim3(256), 0, stream, utf8, offsets, n, mode,
                       k, out, status);
    return 0;
}

int launch_text_simhash(const uint8_t* utf8, const uint64_t* offsets, size_t n, int mode, uint8_t* out,
                        int32_t* status, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(text_hash_kernel<true>, dim3((unsigned)n), dim3(256), 0, stream, utf8, offsets, n, mode,
                       1u, out, status);
    return 0;
}

}  // namespace ucfp
