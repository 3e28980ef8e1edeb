// text.hip -- batched MinHash-128 and SimHash-64 for gfx950.
//
// Replaces the arithmetic behind text::fingerprint_minhash_with::<128> (src/modality/text.rs:182-236)
// and simhash_dispatch (text.rs:366-421) of the reference, i.e. txtfp's MinHashFingerprinter /
// SimHashFingerprinter, for documents that are ASCII (canonicalisation = lower-casing, UAX#29 word
// segmentation restricted to ASCII, both done here on the GPU) or that the host has already
// canonicalised and tokenised (PRETOKENIZED: tokens separated by single spaces).  Spec: DESIGN.md
// "Text spec" T1..T6; CPU statement: oracle/ (text).
//
// ONE WAVE PER DOCUMENT, no workgroup barrier anywhere.  The wave is the tokenizer:
//   A  64 bytes per step, lane = byte.  "Byte is inside a word" is a function of (prev, cur, next)
//      only (WB5-13 on ASCII), so one __ballot gives the 64-bit word mask of the step; token starts,
//      token ends, a byte's rank among word bytes and its token index are shifts, ANDs and
//      popcounts (v_mbcnt) of that mask -- no scan, no LDS traffic besides the output itself.
//   B  every word byte is written (lower-cased) to its place in the CANONICAL STREAM
//      tok0 ' ' tok1 ' ' ...  in LDS; a k-shingle is one contiguous byte range of that stream.
//   C  when the LDS batch fills (256 tokens / 1.5 KiB) or the document ends: lane = shingle (MinHash)
//      or token (SimHash) hashes its byte range with XXH3_64 straight from LDS,
//   D  then lane = 2 of the 128 slots: every shingle's (h1, h2) is broadcast from LDS and each lane
//      keeps running minima of h1 + i*h2 for its slots i and i + 64 (a wave = all 128 permutations);
//      SimHash: lane = output bit, each token hash is broadcast and lane b counts bit b.
//      The last k-1 complete tokens (and an unfinished one) are carried to the front of the batch.
// ALU-bound: ~16 integer ops per shingle per lane in D; HBM traffic is 4 KiB + 1 KiB per document.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ucfp_xxh3.h"
#include "common.h"

namespace ucfp {

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kTokCap = 256;        // tokens per LDS batch
constexpr int kCanonCap = 1536;     // canonical bytes per LDS batch
constexpr int kStepTok = 33;        // a 64-byte step can open at most 32 (+1 carried) tokens

struct WaveLds {
    uint8_t stage[256 + 8];
    uint8_t canon[kCanonCap + 72];
    uint16_t cstart[kTokCap + 8];
    uint16_t cend[kTokCap + 8];
    uint64_t h1[kTokCap];
    uint64_t h2[kTokCap];
};

enum { C_O = 0, C_L = 1, C_N = 2, C_ML = 3, C_MNL = 4, C_MN = 5 };

__device__ __forceinline__ int cls(uint32_t c) {
    const uint32_t lc = c | 0x20u;
    int r = C_O;
    r = (lc - 'a' <= 25u || c == '_') ? C_L : r;
    r = (c - '0' <= 9u) ? C_N : r;
    r = (c == ':') ? C_ML : r;
    r = (c == '.' || c == '\'') ? C_MNL : r;
    r = (c == ',' || c == ';') ? C_MN : r;
    return r;
}

__device__ __forceinline__ bool inword(uint32_t p, uint32_t c, uint32_t q, bool pretok) {
    if (pretok) return c != ' ' && c != 0;
    const int cc = cls(c), pc = cls(p), qc = cls(q);
    const bool mid_l = (cc == C_ML || cc == C_MNL) && pc == C_L && qc == C_L;   // WB6/7
    const bool mid_n = (cc == C_MN || cc == C_MNL) && pc == C_N && qc == C_N;   // WB11/12
    return cc == C_L || cc == C_N || mid_l || mid_n;
}

// XXH3 over the canonical token stream in LDS.  Unaligned 8 / 4-byte words come from ALIGNED dwords and
// v_alignbyte (3 + 2 or 2 + 1 instructions instead of 8 / 4 byte loads and a shift/or ladder); the stream has
// 72 bytes of slack behind it, so the dword past the end is readable.  The hash body is force-inlined, which
// also keeps the pointer in the LDS address space (ds_read, not flat loads).
#define UCFP_RD8_LDS(p, i) ((p)[(i)])
__device__ __forceinline__ uint64_t xxh3_lds_rd64(const uint8_t* src, size_t o) {
    const uint8_t* q = src + o;
    const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(q) & 3u;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(q - sh);   // pointer arithmetic keeps the LDS address space
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    return (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32);
}
__device__ __forceinline__ uint32_t xxh3_lds_rd32(const uint8_t* src, size_t o) {
    const uint8_t* q = src + o;
    const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(q) & 3u;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(q - sh);
    return __builtin_amdgcn_alignbyte(p[1], p[0], sh);
}
UCFP_XXH3_DEFINE_BODY(xxh3_lds, const uint8_t*, UCFP_RD8_LDS)

__device__ __forceinline__ uint64_t mix_h2(uint64_t h1) {
    uint64_t z = h1 + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (z ^ (z >> 31)) | 1ull;
}

__device__ __forceinline__ uint32_t popc_below(uint64_t m, int lane) {  // bits of m below `lane`
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    (void)lane;
}

__device__ __forceinline__ void wave_sync() { wave_lds_sync(); }

}  // namespace

// MODE_SIM = false: MinHash (out 1032 B/doc); true: SimHash (out 8 B/doc)
template <bool MODE_SIM>
__global__ __launch_bounds__(64 * kWavesPerBlock) void text_hash_kernel(
    const uint8_t* __restrict__ utf8, const uint64_t* __restrict__ offsets, size_t n, int pretok_i, uint32_t k,
    uint8_t* __restrict__ out, int32_t* __restrict__ status) {
    __shared__ WaveLds lds[kWavesPerBlock];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t doc = (size_t)blockIdx.x * kWavesPerBlock + wave;
    if (doc >= n) return;  // whole wave
    WaveLds& L = lds[wave];
    const bool pretok = pretok_i != 0;
    const uint8_t* __restrict__ text = utf8 + offsets[doc];
    const size_t len = (size_t)(offsets[doc + 1] - offsets[doc]);
    const bool aligned4 = (reinterpret_cast<uintptr_t>(text) & 3u) == 0;

    uint64_t m0 = ~0ull, m1 = ~0ull;  // MinHash running minima: slots lane, lane + 64
    uint32_t ones = 0;                // SimHash: count of bit `lane`
    uint32_t total_tok = 0;           // complete tokens consumed by flushes (net of carried ones)
    bool any_shingle = false, nonascii = false, too_long = false;

    // wave-uniform tokenizer state of the current LDS batch
    uint32_t ntok = 0;        // tokens opened in this batch (the last one may be unfinished)
    uint32_t cbase = 0;       // word bytes written in this batch
    bool carry = false;       // the byte just before the current step was a word byte
    uint32_t prev_last = 0;   // that byte

    // consume the batch: hash complete items, fold them in, carry the tail to the front
    auto flush = [&](bool final) {
        wave_sync();
        const uint32_t ncomplete = ntok - (carry && !final ? 1u : 0u);
        uint32_t nitems, keep_from;
        if (MODE_SIM) {
            nitems = ncomplete;
            keep_from = ncomplete;
        } else if (ncomplete >= k) {
            nitems = ncomplete - k + 1;
            keep_from = ncomplete - (k - 1);
        } else if (final && !any_shingle && ncomplete > 0) {
            nitems = 1;  // fewer than k tokens in the whole document: one shingle of all of them
            keep_from = ncomplete;
        } else {
            nitems = 0;
            keep_from = 0;
        }
        for (uint32_t s0 = 0; s0 < nitems; s0 += 64) {
            const uint32_t s = s0 + lane;
            if (s < nitems) {
                uint32_t e;
                if (MODE_SIM) e = s;
                else e = ncomplete >= k ? s + k - 1 : ncomplete - 1;
                const uint32_t a = L.cstart[s], b = L.cend[e];
                const uint64_t h = xxh3_lds(L.canon + a, (size_t)(b - a));
                L.h1[s] = h;
                if (!MODE_SIM) L.h2[s] = mix_h2(h);
            }
        }
        wave_sync();
        if (MODE_SIM) {
            for (uint32_t s = 0; s < nitems; s++) ones += (uint32_t)((L.h1[s] >> lane) & 1ull);
        } else {
#pragma unroll 4
            for (uint32_t s = 0; s < nitems; s++) {
                const uint64_t h = L.h1[s], g = L.h2[s];
                const uint64_t v0 = h + (uint64_t)lane * g;
                const uint64_t v1 = v0 + (g << 6);
                m0 = v0 < m0 ? v0 : m0;
                m1 = v1 < m1 ? v1 : m1;
            }
        }
        if (nitems) any_shingle = true;
        total_tok += keep_from;
        if (final) return;
        // carry tokens [keep_from, ntok) to the front
        if (keep_from == 0) return;  // nothing consumed (fewer than k complete tokens): the caller re-checks room
        const uint32_t src0 = keep_from < ntok ? L.cstart[keep_from] : cbase + ntok - 1 + (carry ? 1u : 0u);
        const uint32_t used = cbase + (ntok ? ntok - 1 : 0);   // bytes of canon in use
        const uint32_t nkeep = ntok - keep_from;
        wave_sync();
        uint16_t ks = 0, ke = 0;
        if ((uint32_t)lane < nkeep) {   // nkeep <= k <= 64
            ks = (uint16_t)(L.cstart[keep_from + lane] - src0);
            ke = (uint16_t)(L.cend[keep_from + lane] - src0);
        }
        for (uint32_t o = 0; src0 + o < used; o += 64) {
            const uint32_t i = src0 + o + lane;
            const uint8_t v = i < used ? L.canon[i] : 0;
            wave_sync();
            if (i < used) L.canon[o + lane] = v;
            wave_sync();
        }
        if ((uint32_t)lane < nkeep) {
            L.cstart[lane] = ks;
            L.cend[lane] = ke;
        }
        // word bytes kept = total kept bytes minus the separators between kept tokens
        const uint32_t kept_bytes = used > src0 ? used - src0 : 0;
        ntok = nkeep;
        cbase = kept_bytes - (nkeep ? nkeep - 1 : 0);
        wave_sync();
    };

    // ---- stream the document, 256 bytes per outer iteration, 64 per step ----
    auto load_chunk = [&](size_t base) -> uint32_t {  // this lane's 4 bytes of [base, base + 256)
        const size_t o = base + 4 * (size_t)lane;
        if (o >= len) return 0u;
        if (aligned4 && o + 4 <= len) return *reinterpret_cast<const uint32_t*>(text + o);
        uint32_t v = 0;
        for (int j = 0; j < 4; j++)
            if (o + j < len) v |= (uint32_t)text[o + j] << (8 * j);
        return v;
    };
    uint32_t cur = load_chunk(0);
    for (size_t base = 0; base < len && !too_long; base += 256) {
        const uint32_t nxt = load_chunk(base + 256);
        wave_sync();
        *reinterpret_cast<uint32_t*>(&L.stage[4 * lane]) = cur;
        if (lane == 0) *reinterpret_cast<uint32_t*>(&L.stage[256]) = __builtin_amdgcn_readfirstlane(nxt);
        wave_sync();
        if (!pretok) nonascii |= (cur & 0x80808080u) != 0;
#pragma unroll 1
        for (int sub = 0; sub < 4; sub++) {
            const size_t pos = base + 64 * sub + lane;
            if (base + 64 * sub >= len) break;
            // make room: a step opens at most 32 tokens and writes at most 64 + 32 bytes
            if (ntok + kStepTok > (uint32_t)kTokCap || cbase + ntok + 130 > (uint32_t)kCanonCap) {
                flush(false);
                if (ntok + kStepTok > (uint32_t)kTokCap || cbase + ntok + 130 > (uint32_t)kCanonCap) too_long = true;
                if (too_long) break;
            }
            const uint32_t c = L.stage[64 * sub + lane];
            const uint32_t q = L.stage[64 * sub + lane + 1];
            uint32_t p = __shfl_up(c, 1, 64);
            if (lane == 0) p = prev_last;
            const bool w = pos < len && inword(p, c, q, pretok);
            const uint64_t inw = __ballot(w);
            const uint64_t prev = (inw << 1) | (carry ? 1ull : 0ull);
            const uint64_t starts = inw & ~prev;
            const uint64_t endmark = ~inw & prev;   // first non-word byte after a token
            const uint32_t nin_before = popc_below(inw, lane);
            const uint32_t nst_before = popc_below(starts, lane);
            const bool is_start = (starts >> lane) & 1ull;
            if (w) {
                const uint32_t tok = ntok + nst_before + (is_start ? 1u : 0u) - 1u;
                const uint32_t cpos = cbase + nin_before + tok;
                uint32_t ch = c;
                if (!pretok && ch - 'A' <= 25u) ch += 32;
                L.canon[cpos] = (uint8_t)ch;
                if (is_start) {
                    L.cstart[tok] = (uint16_t)cpos;
                    if (cpos > 0) L.canon[cpos - 1] = ' ';
                }
            }
            if ((endmark >> lane) & 1ull) {
                const uint32_t tok = ntok + nst_before - 1u;   // starts strictly before this byte
                L.cend[tok] = (uint16_t)(cbase + nin_before + tok);
            }
            ntok += (uint32_t)__popcll(starts);
            cbase += (uint32_t)__popcll(inw);
            carry = (inw >> 63) & 1ull;
            prev_last = __shfl(c, 63, 64);
        }
        cur = nxt;
    }
    // close a token that runs to the end of the document, then the final flush
    if (carry && ntok > 0 && lane == 0) L.cend[ntok - 1] = (uint16_t)(cbase + ntok - 1);
    if (!too_long) flush(true);

    // ---- emit ----
    const uint64_t na = __ballot(nonascii);
    int32_t stv = 0;
    if (na) stv = 1;                          // non-ASCII in raw mode: host must pre-tokenise
    else if (too_long) stv = -2;              // UCFP_E_UNSUPPORTED: a token / k-token run exceeds the LDS batch
    else if (total_tok == 0 || (!MODE_SIM && !any_shingle)) stv = -1;   // UCFP_E_MODALITY: no tokens
    if (MODE_SIM) {
        const uint64_t bits = __ballot(2u * ones > total_tok);
        if (lane == 0) {
            const uint64_t v = stv == 0 ? bits : 0ull;
            uint8_t* o8 = out + doc * 8;
            for (int b = 0; b < 8; b++) o8[b] = (uint8_t)(v >> (8 * b));
        }
    } else {
        uint8_t* rec = out + doc * 1032;
        const uint64_t a = stv == 0 ? m0 : 0ull, b = stv == 0 ? m1 : 0ull;
        // 1032-byte records are only 8-byte aligned when the base is: write dwords
        uint32_t* o0 = reinterpret_cast<uint32_t*>(rec + 8 + 8 * lane);
        uint32_t* o1 = reinterpret_cast<uint32_t*>(rec + 8 + 8 * (lane + 64));
        o0[0] = (uint32_t)a;
        o0[1] = (uint32_t)(a >> 32);
        o1[0] = (uint32_t)b;
        o1[1] = (uint32_t)(b >> 32);
        if (lane == 0) {
            uint32_t* o32 = reinterpret_cast<uint32_t*>(rec);
            o32[0] = stv == 0 ? 1u : 0u;  // schema: u16 = 1, pad
            o32[1] = 0;
        }
    }
    if (status && lane == 0) status[doc] = stv;
}

int launch_text_minhash(const uint8_t* utf8, const uint64_t* offsets, size_t n, int mode, uint32_t k,
                        uint8_t* out, int32_t* status, hipStream_t stream) {
    if (n == 0) return 0;
    const unsigned grid = (unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(text_hash_kernel<false>, dim3(grid), dim3(64 * kWavesPerBlock), 0, stream, utf8, offsets, n,
                       mode, k, out, status);
    return 0;
}

int launch_text_simhash(const uint8_t* utf8, const uint64_t* offsets, size_t n, int mode, uint8_t* out,
                        int32_t* status, hipStream_t stream) {
    if (n == 0) return 0;
    const unsigned grid = (unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(text_hash_kernel<true>, dim3(grid), dim3(64 * kWavesPerBlock), 0, stream, utf8, offsets, n,
                       mode, 1u, out, status);
    return 0;
}

}  // namespace ucfp
