// hamming.hip -- exact brute-force Hamming top-k over 64-bit fingerprints for gfx950.
//
// The reference has no Hamming search (/v1/query is cosine-only,
// src/index/embedded/mod.rs:268-360); this is the new capability BASELINE config 5 puts behind
// that route.  Semantics (DESIGN.md "Hamming spec"): distance d = popcount(q ^ x); results
// ordered by (d ascending, record_id ascending); at most k per query.
//
// Three scans share one structure (an upper bound tau[q] on the final k-th distance turns the search into a
// filter; bounds come from the matrix cores' own bound pass (hamming_bound_mfma: k-th smallest of 256 group minima) or, for
// the lane and robust tiers, from a 32k-code sample, then from the candidates found so far, over ranges growing 4x.  Batches
// of 9 .. 256 queries with k <= 16 -- what the search micro-batcher sends -- run a SHORT chain: bound pass over 2^20 codes,
// ONE filter stage over the whole shard whose scan derives its thresholds itself, rescan, selection: 6 launches):
//   many queries   hamming_scan_mfma   the pair distance as a +-1 x 0/1 contraction on the matrix cores (FP4
//                                      operands, exact); suspect blocks are logged and re-evaluated exactly by hamming_rescan
//   few queries    hamming_scan_lanes  lane = code, queries in SGPRs: a pure HBM stream
//   robust tier    hamming_scan        lane = query, corpus code wave-uniform (s_load_dwordx16), per-pair work
//                                      2 v_xor + 2 v_bcnt + 1 compare, lane-private LDS candidate lists pruned
//                                      wave-synchronously: small corpora, and -- gated device-side by a flag --
//                                      any batch whose lists or logs overflowed.  Results never depend on the
//                                      heuristics of the other two.
// Kernels in pipeline order:
//   hamming_bound_mfma (+ hamming_bound_tau above 256 queries)   first thresholds of matrix-core batches
//   hamming_sample_hist[_lanes] / hist_reduce / tau0   k-th smallest sampled distance per query (lane and robust tiers)
//   per stage: hamming_scan_mfma + hamming_rescan  (or hamming_scan_lanes), hamming_list_tau
//   hamming_final_select                               top-k of the candidate lists by (d, id)
//   hamming_scan + topk_merge_u32                      robust tier / fallback
// topk_merge_u32 is also the last step after the multi-GPU all-gather.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace ucfp {

constexpr int kWave = 64;
// A query's candidate list is kSub sub-lists (slots [s cap / kSub, (s + 1) cap / kSub) of its cap slots, counter
// cand_cnt[q kSub + s]); an append takes the sub-list of its workgroup's number.  ~80 appends per query in a first stage
// are 80 atomics with return on ONE address otherwise, a dependent chain in the memory-side atomic unit (19 of the first
// rescan's 38 us at 12.5 M x 4096).  (A 128-byte line per counter, by itself, measured no change.)
constexpr uint32_t kSub = 8;

__device__ __forceinline__ uint32_t hamming64(uint64_t q, uint64_t x) {
    const uint32_t lo = (uint32_t)q ^ (uint32_t)x;
    const uint32_t hi = (uint32_t)(q >> 32) ^ (uint32_t)(x >> 32);
    return __builtin_popcount(hi) + __builtin_popcount(lo);
}

// ---- sample pre-pass --------------------------------------------------------------------
// hist layout: [q][65] u32 (global, zeroed by the launcher).
// grid (parts, qgroups), block 256: four waves scan a quarter of the part each for the same 64 queries and
// share one LDS histogram (more waves in flight to hide the scalar-load latency, one flush per block).
__global__ __launch_bounds__(256) void hamming_sample_hist(const uint64_t* __restrict__ codes,
                                                           size_t sample_n, size_t per_part,
                                                           const uint64_t* __restrict__ queries,
                                                           uint32_t nq, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[65 * kWave];  // [bin][query lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.y * kWave + lane;
    for (int b = threadIdx.x; b < 65 * kWave; b += 256) h[b] = 0;
    __syncthreads();
    const uint64_t qv = queries[q < nq ? q : nq - 1];
    const size_t p0 = (size_t)blockIdx.x * per_part;
    const size_t p1 = p0 + per_part < sample_n ? p0 + per_part : sample_n;
    const size_t quarter = (p1 - p0 + 3) / 4;
    const size_t s0 = p0 + wave * quarter < p1 ? p0 + wave * quarter : p1;
    const size_t s1 = s0 + quarter < p1 ? s0 + quarter : p1;
    size_t i = s0;
    for (; i + 16 <= s1; i += 16) {   // 16 codes per trip: two s_load_dwordx16, one scalar-load latency per trip
        uint64_t x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = codes[i + j];   // wave-uniform address -> scalar loads
#pragma unroll
        for (int j = 0; j < 16; j++)
            atomicAdd(&h[hamming64(qv, x[j]) * kWave + lane], 1u);   // ds_add without return: no RMW latency chain
    }
    for (; i < s1; i++) atomicAdd(&h[hamming64(qv, codes[i]) * kWave + lane], 1u);
    __syncthreads();
    // partial histogram of this part: hist[part][bin][query], plain coalesced stores (no atomics, no memset)
    const size_t nqp = (size_t)gridDim.y * kWave;
    for (int e = threadIdx.x; e < 65 * kWave; e += 256)
        hist[((size_t)blockIdx.x * 65 + (e >> 6)) * nqp + blockIdx.y * kWave + (e & 63)] = h[e];
}

// sum of the partial histograms [part][bin][padded q] into part 0, one thread per (bin, q): 32 independent loads
__global__ void hamming_hist_reduce(uint32_t* __restrict__ hist, uint32_t parts, uint32_t nq_padded) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)65 * nq_padded) return;
    uint32_t c = 0;
    for (uint32_t p = 0; p < parts; p++) c += hist[(size_t)p * 65 * nq_padded + i];
    hist[i] = c;
}

// stride_b / stride_q: hist[b * stride_b + q * stride_q]  ([q][65] on the few-query path, [bin][padded q] otherwise)
// also empties the candidate lists and the overflow flag of the search that starts here (cand_cnt may be null)
__global__ void hamming_tau0(const uint32_t* __restrict__ hist, uint32_t nq, uint32_t k, uint32_t stride_b,
                             uint32_t stride_q, uint32_t* __restrict__ tau0, uint32_t* __restrict__ cand_cnt,
                             uint32_t* __restrict__ overflow) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    if (cand_cnt) {
        for (uint32_t u = 0; u < kSub; u++) cand_cnt[(size_t)q * kSub + u] = 0;
        if (q == 0) *overflow = 0;
    }
    uint32_t c[65];
#pragma unroll
    for (int b = 0; b < 65; b++) c[b] = hist[(size_t)b * stride_b + (size_t)q * stride_q];   // 65 loads in flight
    uint32_t cum = 0, t = 64;
    bool done = false;
#pragma unroll
    for (int b = 0; b < 65; b++) {
        cum += c[b];
        if (!done && cum >= k) {
            t = b;
            done = true;
        }
    }
    tau0[q] = t;  // 64 when the sample holds fewer than k codes: accept everything
}

// ---- lane-private candidate lists in LDS ---------------------------------------------------
// CAP entries per lane, layout [entry][lane] so that a wave-wide access to entry e is one
// conflict-free row. Ordering key: (d, id) ascending.
template <int CAP>
struct CandLists {
    uint64_t id[CAP * kWave];
    uint32_t d[CAP * kWave];
};

__device__ __forceinline__ bool key_less(uint32_t d1, uint64_t i1, uint32_t d2, uint64_t i2) {
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

// Wave-synchronous partial selection sort: afterwards entries [0, min(cnt,k)) are the smallest
// by (d, id), ascending. Every lane runs the same trip counts (bounded by the wave max of cnt).
template <int CAP>
__device__ __forceinline__ void prune(CandLists<CAP>& L, int lane, uint32_t& cnt, uint32_t k) {
    uint32_t maxcnt = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(maxcnt, off, kWave);
        maxcnt = o > maxcnt ? o : maxcnt;
    }
    const uint32_t kk = k < maxcnt ? k : maxcnt;
    for (uint32_t p = 0; p < kk; p++) {
        uint32_t bd = 0xffffffffu;
        uint64_t bi = ~0ull;
        uint32_t be = p;
        for (uint32_t e = p; e < maxcnt; e++) {
            const uint32_t dd = e < cnt ? L.d[e * kWave + lane] : 0xffffffffu;
            const uint64_t ii = e < cnt ? L.id[e * kWave + lane] : ~0ull;
            if (key_less(dd, ii, bd, bi)) {
                bd = dd;
                bi = ii;
                be = e;
            }
        }
        if (p < cnt && be != p) {
            const uint32_t td = L.d[p * kWave + lane];
            const uint64_t ti = L.id[p * kWave + lane];
            L.d[p * kWave + lane] = bd;
            L.id[p * kWave + lane] = bi;
            L.d[be * kWave + lane] = td;
            L.id[be * kWave + lane] = ti;
        }
    }
    if (cnt > k) cnt = k;
}

// ---- main scan ---------------------------------------------------------------------------
// grid (slices, qgroups), block 64 (one independent wave).
// partial layout: [slice][q][k] for ids / dist, and [slice][q] for counts.
template <int CAP>
__global__ __launch_bounds__(64) void hamming_scan(
    const uint64_t* __restrict__ codes, const uint64_t* __restrict__ ids, size_t n, size_t per_slice,
    const uint64_t* __restrict__ queries, uint32_t nq, uint32_t k, const uint32_t* __restrict__ tau0,
    uint64_t* __restrict__ part_ids, uint32_t* __restrict__ part_d, uint32_t* __restrict__ part_cnt,
    const uint32_t* __restrict__ run_flag) {
    __shared__ CandLists<CAP> L;
    if (run_flag && *run_flag == 0) return;  // fallback tier: only when the fast tier overflowed
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.y * kWave + lane;
    const bool live = q < nq;
    const uint64_t qv = queries[live ? q : nq - 1];
    const uint32_t qlo = (uint32_t)qv, qhi = (uint32_t)(qv >> 32);
    // dead lanes never accept: tau = -1 as signed
    int32_t tau = live ? (int32_t)tau0[q] : -1;
    uint32_t cnt = 0;

    const size_t s0 = (size_t)blockIdx.x * per_slice;
    const size_t s1 = s0 + per_slice < n ? s0 + per_slice : n;

    auto slow = [&](size_t row, uint32_t d) {
        // wave-uniform entry; lanes with d <= tau append, everyone may prune
        const bool hit = (int32_t)d <= tau;
        if (hit) {
            const uint64_t rid = ids[row];  // wave-uniform address
            L.d[cnt * kWave + lane] = d;
            L.id[cnt * kWave + lane] = rid;
            cnt++;
        }
        if (__any(cnt == (uint32_t)CAP)) {
            prune<CAP>(L, lane, cnt, k);
            if (cnt == k) {
                const int32_t kth = (int32_t)L.d[(k - 1) * kWave + lane];
                tau = kth < tau ? kth : tau;
            }
        }
    };

    // 16 codes per trip, software-pipelined: the scalar loads of trip t+1 (2 x s_load_dwordx16)
    // are issued before the 80 VALU ops of trip t, so their latency hides behind the popcounts.
    constexpr int G = 16;
    size_t i = s0;
    uint64_t cur[G], nxt[G];
    const bool have_full = i + G <= s1;
    if (have_full) {
#pragma unroll
        for (int j = 0; j < G; j++) cur[j] = codes[i + j];
    }
    for (; i + G <= s1; i += G) {
        const bool more = i + 2 * G <= s1;
        if (more) {
#pragma unroll
            for (int j = 0; j < G; j++) nxt[j] = codes[i + G + j];
        }
        uint32_t d[G];
        bool any_hit = false;
#pragma unroll
        for (int j = 0; j < G; j++) {
            d[j] = __builtin_popcount(qlo ^ (uint32_t)cur[j]) + __builtin_popcount(qhi ^ (uint32_t)(cur[j] >> 32));
            any_hit |= (int32_t)d[j] <= tau;
        }
        if (__any(any_hit)) {
#pragma unroll
            for (int j = 0; j < G; j++)
                if (__any((int32_t)d[j] <= tau)) slow(i + j, d[j]);
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < G; j++) cur[j] = nxt[j];
        }
    }
    for (; i < s1; i++) {
        const uint64_t x = codes[i];
        const uint32_t d = __builtin_popcount(qlo ^ (uint32_t)x) + __builtin_popcount(qhi ^ (uint32_t)(x >> 32));
        if (__any((int32_t)d <= tau)) slow(i, d);
    }

    prune<CAP>(L, lane, cnt, k);
    if (live) {
        const size_t base = ((size_t)blockIdx.x * nq + q) * k;
        for (uint32_t e = 0; e < k; e++) {
            const bool v = e < cnt;
            part_ids[base + e] = v ? L.id[e * kWave + lane] : ~0ull;
            part_d[base + e] = v ? L.d[e * kWave + lane] : 0xffffffffu;
        }
        part_cnt[(size_t)blockIdx.x * nq + q] = cnt;
    }
}

// ---- fast tier: a filter on the matrix cores ---------------------------------------------------
// Once an upper bound tau[q] on the final k-th distance is known, the corpus only has to be
// FILTERED: every row with d <= tau[q] goes to q's candidate list in global memory, everything else
// is dropped unseen.  With a batch of queries that scan is ALU-bound (HBM sees < 1 TB/s), and the
// distance is an exact small-integer contraction: with x_i = 1/0 for the code bits and y_i = +1/-1
// for the query bits,  sum_i x_i y_i = popc(q) - d(q, x).  0, +1 and -1 are exact in FP4 (e2m1: nibbles 0x0, 0x2,
// 0xA) and the products accumulate in f32, so ONE v_mfma_f32_32x32x64_f8f6f4 (cbsz = blgp = 4: both operands FP4,
// K = 64 = the whole code) with  A = 32 codes (rows),  B = 32 queries (columns)  gives 1024 of these sums in 32
// cycles -- twice the int8 rate (two v_mfma_i32_32x32x32_i8, 64 cycles: rounds 1-2 of this file) and ~9x the
// popcount path (~290) -- and a pair is a candidate iff sum >= popc(q) - tau[q].  Small integers in f32: nothing is
// approximated (tools/probe_mfma_fp4.hip checks the operand and result layout against popcounts).
//   K map   lane (nn, hh) holds the 32 bits [32 hh, 32 hh + 32) of row / column nn as 4 dwords of 8 nibbles; bit
//           4 i + j of that word sits in nibble i of dword j.  A and B use the SAME map, which is all a dot product
//           needs, and it makes the expansion four shifted ANDs with 0x22222222 per code tile.
//   layout  the 16 results a lane holds all belong to ONE query (column = lane & 31), so they are folded to one
//           maximum before the single compare against the lane's threshold.  Folding with v_max3_f32 (two results per
//           instruction, 8 per MFMA) put the vector issue port -- 8 cycles per MFMA + 32 -- over the matrix pipe's 32,
//           so two code tiles share an accumulator and the matrix core packs their sums into one f32 (UCFP_FOLD_PAIR
//           below): v_pk_maximum3_f16 then folds four sums per instruction and the matrix pipe is the bound again
//   LDS     the +-1 image of the whole batch (<= 4096 queries, 1 KiB per 32-query tile,
//           lane-contiguous for ds_read_b128) and the packed thresholds, built once per workgroup
//   wave    expands 4 code tiles = 2 pairs (128 codes) into registers, then walks all query tiles: 4 MFMAs,
//           17 v_pk_maximum3, a saturating subtract and a compare per tile, software-pipelined by one tile, in place
//   hits    rare by construction of tau.  The scan only RECORDS a suspect step (query tile, first code, the four
//           code tiles' ballots of the lanes over threshold) in the wave's own slice of a global log -- one
//           store, no atomics, nothing to wait for.  hamming_rescan then re-evaluates the flagged
//           (query, 16-code half) combinations with plain popcounts (exact; it checks the result layout it
//           assumes and raises the fallback flag on any disagreement) and appends the true candidates to the
//           per-query lists.
// The scan runs in stages over geometrically growing ranges: tau0 (k-th distance inside a 32k-code
// sample) filters [0, 4 x 32k); the k-th smallest distance of the candidates so far filters the next
// 4x larger range, and so on, so every stage admits only ~4k..20k candidates per query.  The final
// top-k is selected from the lists.  If a log or a list overflows (adversarial order) a flag routes
// the batch through the robust tier, device-side -- results never depend on the heuristic.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kQP = 4096;    // queries per pass (their +-1 image is resident in LDS): one pass per search call
constexpr int kTB = 4;       // 32-code tiles per wave step
constexpr int kMW = 16;      // waves per workgroup (4 per SIMD; the loop is software-pipelined inside a wave)
constexpr int kStep = kTB * 32;
constexpr size_t kFusedTauLds = 65 * 64 * 4 + 256 * 4;   // hamming_scan_mfma deriving its thresholds: counters [65][64] + 256 thresholds
constexpr int kStreamTiles = 8;     // hamming_scan_mfma: up to this many query tiles run as one pipeline across code steps
static_assert(kQP >= (int)kHammingMaxBatch, "one pass covers a whole search call");

// FP4 (E2M1) operand words of a 32-bit half: dword j nibble i <- bit 4 i + j.  A code bit is left IN PLACE inside its nibble
// where that is a valid magnitude -- bit 0 -> 0x1 (0.5), bit 1 -> 0x2 (1.0), bit 2 -> 0x4 (2.0); only bit 3 (the sign
// position) is moved, to 0x2 -- and the query image carries the reciprocal magnitude at the same position (+-2.0, +-1.0,
// +-0.5, +-1.0), so that every product is +-1 or 0 exactly.  5 vector instructions per half instead of 7: a batch of one
// query tile is bound by vector issue (round 4: 100 instructions per 128-code step against 4 matrix instructions).
__device__ __forceinline__ i32x4 expand_code_fp4(uint32_t w) {
    i32x4 v;
    v[0] = (int)(w & 0x11111111u);
    v[1] = (int)(w & 0x22222222u);
    v[2] = (int)(w & 0x44444444u);
    v[3] = (int)((w >> 2) & 0x22222222u);
    return v;
}
// Queries: 1 -> +m, 0 -> -m (sign = bit 3 of the nibble), m = 2.0 (0x4) / 1.0 (0x2) / 0.5 (0x1) / 1.0 (0x2) for dwords 0..3.
__device__ __forceinline__ i32x4 expand_query_fp4(uint32_t w) {
    const uint32_t n = ~w;
    i32x4 v;
    v[0] = (int)(0x44444444u | ((n << 3) & 0x88888888u));
    v[1] = (int)(0x22222222u | ((n << 2) & 0x88888888u));
    v[2] = (int)(0x11111111u | ((n << 1) & 0x88888888u));
    v[3] = (int)(0x22222222u | (n & 0x88888888u));
    return v;
}

size_t hamming_mfma_lds_bytes(uint32_t nq) {
    const uint32_t tiles = ((nq < (uint32_t)kQP ? nq : (uint32_t)kQP) + 31) / 32;
    return (size_t)(tiles + 2) * (1024 + 128);   // +2: the drain tile and its operand prefetch
}
uint32_t hamming_log_slices(uint32_t nq) { return 256u * ((nq + kQP - 1) / kQP) * kMW; }

// Two code tiles share one accumulator: the matrix core itself packs their sums into one f32,
//   D = A_a x B + C              C = 2^23 + 2^22 + 1088 in every element
//   D = (2^16 A_b) x B + D       v_mfma_scale_...: block scale 2^16 on A (E8M0 143), 1 on B (127)
// = 2^23 + 65536 (64 + s_b) + (1088 + s_a), an integer below 2^24, so its bit pattern is
// 0x4B00'0000 | (64 + s_b) << 16 | (1088 + s_a): two ordered 16-bit fields, both valid monotone f16 patterns, which
// v_pk_maximum3_f16 folds FOUR sums at a time -- 8 fold instructions per two MFMAs, which puts the loop back under the
// matrix pipe (tools/probe_mfma_fp4_pack.hip checks the packing bit for bit; tools/ubench_mfma_i8.hip mode 23 the rate:
// 63-64 T pairs/s on random operands against 54 for one v_max3_f32 fold per MFMA).  s_b = +64 would carry out of its
// field; it needs popc(q) = 64, and such a query is filtered as q with bit 0 cleared and tau + 1 (see filter_query).
// As text, in place: eight instructions folding the pair's PREVIOUS results (query tile t-1), then the two MFMAs of
// query tile t into the same registers.  %0 the results (read-write), %1 running max (packed), %2 / %21 the two code
// tiles (A), %3 the query tile (B), %4..%19 the 16 results as scalars (the same registers as %0), %20 the bias
// tuple C, %22 / %23 the block scales.
#define UCFP_FOLD_PAIR                                                                                   \
    "v_pk_maximum3_f16 %1, %4, %5, %6\n\t"                                                               \
    "v_pk_maximum3_f16 %1, %1, %7, %8\n\t"                                                               \
    "v_pk_maximum3_f16 %1, %1, %9, %10\n\t"                                                              \
    "v_pk_maximum3_f16 %1, %1, %11, %12\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %13, %14\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %15, %16\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %17, %18\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %19, %19\n\t"                                                             \
    "v_mfma_f32_32x32x64_f8f6f4 %0, %2, %3, %20 cbsz:4 blgp:4\n\t"                                       \
    "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %21, %3, %0, %22, %23 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
// the second pair of a step also takes the verdict: %2 scratch, %3 = lane mask of "some field >= its threshold" in an
// SGPR pair, inputs shifted by two, %26 the first pair's maximum, %27 the threshold word (each field = threshold - 1:
// a saturating subtraction leaves a non-zero field iff max >= threshold)
#define UCFP_FOLD_PAIR_LAST                                                                              \
    "v_pk_maximum3_f16 %1, %6, %7, %8\n\t"                                                               \
    "v_pk_maximum3_f16 %1, %1, %9, %10\n\t"                                                              \
    "v_pk_maximum3_f16 %1, %1, %11, %12\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %13, %14\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %15, %16\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %17, %18\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %19, %20\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %21, %21\n\t"                                                             \
    "v_mfma_f32_32x32x64_f8f6f4 %0, %4, %5, %22 cbsz:4 blgp:4\n\t"                                       \
    "v_pk_maximum3_f16 %2, %1, %26, %26\n\t"                                                             \
    "v_pk_sub_u16 %2, %2, %27 clamp\n\t"                                                                 \
    "v_cmp_ne_u32 %3, 0, %2\n\t"                                                                         \
    "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %23, %5, %0, %24, %25 op_sel_hi:[0,0,0] cbsz:4 blgp:4\n\t"     \
    "s_nop 0"
#define UCFP_FOLD_PAIR_IN(p)                                                                                      \
    "v"(A[2 * p]), "v"(bq), "v"(D[p][0]), "v"(D[p][1]), "v"(D[p][2]), "v"(D[p][3]), "v"(D[p][4]), "v"(D[p][5]),   \
        "v"(D[p][6]), "v"(D[p][7]), "v"(D[p][8]), "v"(D[p][9]), "v"(D[p][10]), "v"(D[p][11]), "v"(D[p][12]),      \
        "v"(D[p][13]), "v"(D[p][14]), "v"(D[p][15]), "v"(cc), "v"(A[2 * p + 1]), "v"(sa), "v"(sb)

constexpr uint32_t kFieldHi = 0x4B00u + 64u, kFieldLo = 1088u;   // a sum s sits in its field as kField + s

// The query the FILTER sees and the slack its threshold gets: popc(q) = 64 would let a sum reach +64 and carry out of
// the packed field, so that one query is filtered with bit 0 cleared and tau + 1 -- d(q', x) <= d(q, x) + 1, so nothing
// is lost, and hamming_rescan evaluates the suspects against the true query.
__device__ __forceinline__ uint64_t filter_query(uint64_t q, uint32_t& slack) {
    slack = q == ~0ull ? 1u : 0u;
    return q == ~0ull ? q & ~1ull : q;
}

// +-1 image of the pass's queries in LDS, [tile][lane] 16 B (the B operand of query tile t is one ds_read_b128 per lane),
// + 2 zero pad tiles.  Every workgroup builds it from the queries themselves: 8 bytes loaded and half a dozen vector
// instructions per entry, where a prebuilt image cost 16 bytes loaded -- and a launch of its own per search (4 us of the
// 90 a batch of 32 queries takes over 12.5 M codes).  8 loads in flight per thread (a plain loop would pay the global
// latency 17 times in a row for a full pass).
__device__ __forceinline__ void build_query_image(i32x4* QB, const uint64_t* __restrict__ queries, uint32_t nq, uint32_t q0,
                                                  uint32_t ntiles, uint32_t nthreads) {
    for (uint32_t s0 = threadIdx.x; s0 < (ntiles + 2) * 64; s0 += nthreads * 8) {
        uint64_t qw[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t s = s0 + u * nthreads, q = q0 + (s >> 6) * 32 + (s & 31);
            qw[u] = (s < ntiles * 64 && q < nq) ? queries[q] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t s = s0 + u * nthreads, q = q0 + (s >> 6) * 32 + (s & 31);
            if (s < (ntiles + 2) * 64) {
                uint32_t slack;
                i32x4 v = {0, 0, 0, 0};   // dead columns are all-zero: their sums are 0
                if (s < ntiles * 64 && q < nq) v = expand_query_fp4((uint32_t)(filter_query(qw[u], slack) >> (32 * ((s >> 5) & 1))));
                QB[s] = v;
            }
        }
    }
}

// log record (48 bytes, one per suspect step of a wave): query tile (global: q / 32), row - begin of the step's first
// code, then the lane ballots of its four 32-code tiles.
// Launched with 4, 8 or kMW waves per workgroup (launch_hamming_search picks per stage: fewer waves round a short
// stage up to a smaller quantum, more waves cover each other's stalls in a long one).
__global__ __launch_bounds__(kMW * 64) void hamming_scan_mfma(
    const uint64_t* __restrict__ codes, size_t begin, size_t end, const uint64_t* __restrict__ queries, uint32_t nq,
    const uint32_t* __restrict__ tau, uint4* __restrict__ log,
    uint32_t* __restrict__ log_cnt, uint32_t log_cap, uint32_t* __restrict__ overflow, size_t strict_from,
    const uint32_t* __restrict__ ids_ascending, uint32_t stream_tiles, const uint8_t* __restrict__ btab, uint32_t btab_groups,
    uint32_t btab_stride, uint32_t bound_k, uint32_t* __restrict__ tau_out) {
    const uint32_t nthreads = blockDim.x, mw = nthreads >> 6;   // waves in this workgroup
    extern __shared__ __attribute__((aligned(16))) uint8_t mf_lds[];
    const uint32_t q0 = blockIdx.y * kQP;
    const uint32_t nqp = nq - q0 < (uint32_t)kQP ? nq - q0 : (uint32_t)kQP;
    const uint32_t ntiles = (nqp + 31) / 32;
    // [tile][lane] 16 B: the B operand of query tile t is one ds_read_b128 per lane
    i32x4* QB = reinterpret_cast<i32x4*>(mf_lds);
    uint32_t* THR = reinterpret_cast<uint32_t*>(mf_lds + (size_t)(ntiles + 2) * 1024);   // [tile][32]: packed, threshold - 1 per field
    constexpr uint32_t kNever = 0xfffefffeu;   // no field exceeds it (nor after the strict rows' + 1 per field)
    // rows from strict_from on are filtered with tau - 1 where ids ascend with the row: the thresholds of the stage that
    // follows the bound pass are the k-th smallest of group minima over rows < strict_from, so k records at most that far
    // AND with smaller ids exist already -- a tie from a later row cannot displace them (as in hamming_list_tau)
    const bool strict_rows = ids_ascending && *ids_ascending;

    const int lane = threadIdx.x & 63;
    const int nn = lane & 31, hh = lane >> 5;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, in an SGPR
    // step st goes to wave (st / workgroups) % mw of workgroup st % workgroups: the steps of a last, partial round land on
    // different CUs and SIMDs (the waves of a SIMD share its matrix pipe, so a stage takes as long as the busiest SIMD has
    // steps: with the partial round's steps all in the first workgroups it cost a whole round, 231 us instead of ~160 for
    // the 1.6 M codes behind 8.4 M of a 10 M corpus)
    const size_t gwave = (size_t)wv * gridDim.x + blockIdx.x, nwaves = (size_t)gridDim.x * mw;
    const size_t nsuper = (end - begin + kStep - 1) / kStep;
    // lane (nn, hh) needs bits [32 hh, +32) of code nn: one 4-byte load, a wave reads 256 contiguous bytes per tile
    const uint32_t* __restrict__ halves = reinterpret_cast<const uint32_t*>(codes);
    auto load_codes = [&](uint32_t (&x)[kTB], size_t st) {
#pragma unroll
        for (int b = 0; b < kTB; b++) {
            const size_t row = begin + st * kStep + 32 * b + nn;
            const bool ok = st < nsuper && row < end;
            x[b] = ok ? __builtin_nontemporal_load(halves + row * 2 + hh) : 0u;
        }
    };
    // The codes of the wave's first two steps leave before the prologue: a batch of a few dozen queries is one query tile, a
    // wave's whole share of a 12.5 M-code shard is 20 steps of ~300 ns, and the memory latency of a step's codes (2-3 us
    // under load) then needs TWO steps of lead -- with one the last stage of 32 queries ran at 2.5 TB/s.
    // The few-tile path reads through a buffer descriptor rebuilt per step from wave-uniform values (first row of the step,
    // bytes of it that exist): the per-lane part of the address is one constant register, rows past the end read as zero
    // by the hardware's range check, and the step's 64-bit address arithmetic + four guards (28 of its 100 vector
    // instructions) become scalar work.
    const uint32_t voff = (uint32_t)(nn * 8 + hh * 4);
    auto load_codes_buf = [&](uint32_t (&x)[kTB], size_t st) {
        const size_t row0 = begin + st * kStep;
        const size_t left = st < nsuper ? end - row0 : 0;
        const uint32_t bytes = (uint32_t)(left < (size_t)kStep ? left : (size_t)kStep) * 8u;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(codes + row0), 0, (int)bytes, 0x00020000);
#pragma unroll
        for (int b = 0; b < kTB; b++) x[b] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, voff + 256u * b, 0, 2 /* nt */);
    };
    const bool few_tiles = ntiles <= stream_tiles;   // (kernel-uniform)
    // (two buffers, used alternately by the few-tile path: a buffer is refilled for the step after next as soon as its
    // codes are expanded -- copying "next" into "current" instead would wait for the next step's codes a whole step early)
    uint32_t x[kTB], xn[kTB];
    if (few_tiles) {
        load_codes_buf(x, gwave);
        load_codes_buf(xn, gwave + nwaves);
    } else {
        load_codes(x, gwave);
    }
    // thresholds popc(q) - tau[q]; dead columns (and the pad tiles) never hit.  The first round's inputs are requested
    // before the image is built, so that the prologue pays the global latency once, not twice.
    // The stage behind the bound pass derives its thresholds itself (batches of up to 256 queries: btab != nullptr): per
    // query the k-th smallest of the table's group minima, + the filter query's slack -- what hamming_bound_tau computes, a
    // launch of its own worth 7 us of such a batch's 80.  Every workgroup repeats it (16 KB of table per 64 queries, from L2);
    // workgroup 0 also publishes the thresholds for the rescan and the fallback tier.  64 queries per round: lane = query,
    // the waves split the groups, [bin][lane] counters in LDS.
    uint32_t* TAU = reinterpret_cast<uint32_t*>(mf_lds + (size_t)(ntiles + 2) * (1024 + 128)) + 65 * kWave;
    if (btab) {
        uint32_t* H = reinterpret_cast<uint32_t*>(mf_lds + (size_t)(ntiles + 2) * (1024 + 128));
        for (uint32_t r0 = 0; r0 < nqp; r0 += kWave) {
            for (uint32_t b = threadIdx.x; b < 65 * kWave; b += nthreads) H[b] = 0;
            __syncthreads();
            const uint32_t q = q0 + r0 + lane;
            const uint32_t qc = q < nq ? q : nq - 1;
            for (uint32_t g0 = wv; g0 < btab_groups; g0 += mw * 16) {
                uint32_t d[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const uint32_t g = g0 + mw * u;
                    d[u] = g < btab_groups ? btab[(size_t)g * btab_stride + qc] : 255u;
                }
#pragma unroll
                for (int u = 0; u < 16; u++)
                    if (d[u] <= 64u) atomicAdd(&H[d[u] * kWave + lane], 1u);
            }
            __syncthreads();
            if (wv == 0) {
                uint32_t cum = 0, t = 64;
                bool done = false;
                for (int b = 0; b < 65; b++) {
                    cum += H[b * kWave + lane];
                    if (!done && cum >= bound_k) {
                        t = b;
                        done = true;
                    }
                }
                uint32_t slack;
                (void)filter_query(queries[qc], slack);
                t += slack;   // d(q, x) <= d(fq, x) + slack
                t = t < 64u ? t : 64u;
                TAU[r0 + lane] = t;
                if (blockIdx.x == 0 && q < nq) tau_out[q] = t;
            }
            __syncthreads();
        }
    }
    uint64_t qv[4];
    uint32_t tv[4];
    auto thr_load = [&](uint32_t s0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t s = s0 + u * nthreads, q = q0 + s;
            const bool live = s < ntiles * 32 && q < nq;
            qv[u] = live ? queries[q] : 0ull;
            tv[u] = !live ? 0u : btab ? TAU[s] : (tau[q] < 64u ? tau[q] : 64u);
        }
    };
    auto thr_store = [&](uint32_t s0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t s = s0 + u * nthreads, q = q0 + s;
            if (s < (ntiles + 2) * 32) {
                uint32_t slack;
                const uint64_t fq = filter_query(qv[u], slack);
                const int thr = (int)__popcll(fq) - (int)(tv[u] + slack);   // a pair is a suspect iff its sum >= thr (>= -65)
                THR[s] = (s < ntiles * 32 && q < nq)
                             ? ((uint32_t)((int)kFieldHi + thr - 1) << 16) | (uint32_t)((int)kFieldLo + thr - 1)
                             : kNever;
            }
        }
    };
    thr_load(threadIdx.x);
    build_query_image(QB, queries, nq, q0, ntiles, nthreads);
    thr_store(threadIdx.x);
    for (uint32_t s0 = threadIdx.x + nthreads * 4; s0 < (ntiles + 2) * 32; s0 += nthreads * 4) {
        thr_load(s0);
        thr_store(s0);
    }
    __syncthreads();

    const size_t slice = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * mw + wv;
    uint4* __restrict__ mylog = log + slice * log_cap * 3;   // 48-byte records
    uint32_t ln = 0;   // records written, wave-uniform

    // results of the software pipeline, in place: one accumulator per PAIR of code tiles.  Their content at the start
    // of a code step is irrelevant: the first fold of every step runs against the threshold "never" ("tile -1").
    static_assert(kTB == 4, "two pairs of code tiles per step");
    f32x16 D[kTB / 2];
#pragma unroll
    for (int p = 0; p < kTB / 2; p++)
#pragma unroll
        for (int e = 0; e < 16; e++) D[p][e] = 0.f;
    f32x16 cc;   // the bias tuple C
#pragma unroll
    for (int e = 0; e < 16; e++) cc[e] = 8388608.f + 4194304.f + (float)kFieldLo;
    asm volatile("" : "+v"(cc));   // sixteen registers, not one rematerialised constant
    const int sa = 127 + 16, sb = 127;   // E8M0 block scales: 2^16 on the pair's second code tile, 1 on the queries
    // the four per-tile ballots of a suspect step (m01 = fields of tiles 1 | 0, m23 = tiles 3 | 2; a threshold field holds
    // threshold - 1) and ONE 48-byte record written by lanes 0..2
    auto log_step = [&](uint32_t thr, uint32_t m01, uint32_t m23, uint32_t tile, uint32_t off) {
        const uint32_t tl = thr & 0xffffu, th = thr >> 16;
        const uint64_t k0 = __ballot((m01 & 0xffffu) > tl), k1 = __ballot((m01 >> 16) > th),
                       k2 = __ballot((m23 & 0xffffu) > tl), k3 = __ballot((m23 >> 16) > th);
        const uint4 ra = make_uint4(tile, off, (uint32_t)k0, (uint32_t)(k0 >> 32));
        const uint4 rb = make_uint4((uint32_t)k1, (uint32_t)(k1 >> 32), (uint32_t)k2, (uint32_t)(k2 >> 32));
        const uint4 rc = make_uint4((uint32_t)k3, (uint32_t)(k3 >> 32), 0u, 0u);
        uint4 v;
        v.x = lane == 0 ? ra.x : lane == 1 ? rb.x : rc.x;
        v.y = lane == 0 ? ra.y : lane == 1 ? rb.y : rc.y;
        v.z = lane == 0 ? ra.z : lane == 1 ? rb.z : rc.z;
        v.w = lane == 0 ? ra.w : lane == 1 ? rb.w : rc.w;
        if (ln < log_cap) {
            if (lane < 3) mylog[(size_t)ln * 3 + lane] = v;
        } else if (lane == 0) {
            *overflow = 1;
        }
        ln++;
    };
    if (few_tiles) {
        // Few query tiles (a batch of up to 256 queries): ONE pipeline over all (code step, query tile) pairs of the wave --
        // the fold inside a step tests the previous pair's results, whichever code step that was, and a single pad tile
        // drains the pipeline at the very end.  The loop below drains after every code step, which for one query tile is
        // half of all matrix instructions plus 24 wait states per step (32 queries over 10.5 M codes: 29.7 us, 2.8 TB/s).
        i32x4 A[kTB];
        auto stepc = [&](const i32x4& bq, uint32_t thr, uint32_t tile, uint32_t off, uint32_t nxt, i32x4& nq_, uint32_t& nthr) {
            uint32_t m01, m23, vs;
            uint64_t hit;
            asm volatile(UCFP_FOLD_PAIR : "+v"(D[0]), "=&v"(m01) : UCFP_FOLD_PAIR_IN(0) : "memory");
            nq_ = QB[nxt * 64 + lane];
            nthr = THR[nxt * 32 + nn];
            asm volatile(UCFP_FOLD_PAIR_LAST
                         : "+v"(D[1]), "=&v"(m23), "=&v"(vs), "=s"(hit)
                         : UCFP_FOLD_PAIR_IN(1), "v"(m01), "v"(thr)
                         : "memory");
            if (__builtin_expect(hit != 0, 0)) log_step(thr, m01, m23, tile, off);
        };
        i32x4 bcur = QB[lane], bnext;
        uint32_t thr_cur = THR[nn], thr_next;
        uint32_t thr_prev = kNever, tile_prev = 0, off_prev = 0;   // the very first fold reads the zeroed accumulators
        auto code_step = [&](uint32_t (&xb)[kTB], size_t st) {
#pragma unroll
            for (int b = 0; b < kTB; b++) A[b] = expand_code_fp4(xb[b]);
            load_codes_buf(xb, st + 2 * nwaves);
            const uint32_t off = (uint32_t)(st * kStep);
            const uint32_t dlt = strict_rows && begin + st * kStep >= strict_from ? 0x00010001u : 0u;
            for (uint32_t t = 0; t < ntiles; t++) {
                stepc(bcur, thr_prev, tile_prev, off_prev, t + 1 == ntiles ? 0u : t + 1, bnext, thr_next);
                thr_prev = thr_cur + dlt;
                tile_prev = q0 / 32 + t;
                off_prev = off;
                bcur = bnext;
                thr_cur = thr_next;
            }
        };
        for (size_t st = gwave; st < nsuper; st += 2 * nwaves) {
            code_step(x, st);
            if (st + nwaves >= nsuper) break;
            code_step(xn, st + nwaves);
        }
        if (gwave < nsuper) {   // wave-uniform
            bcur = QB[ntiles * 64 + lane];   // a pad tile: all zero
            stepc(bcur, thr_prev, tile_prev, off_prev, ntiles, bnext, thr_next);
            // The pad tile's MFMAs are still writing D.  To the compiler D is dead after the last fold, and it hands D's
            // registers to that very step's fold results (seen: m23 in D[0][0], read again by log_step after the matrix
            // core had overwritten it -- one suspect record in a few thousand searches lost its ballots).  Keep both
            // accumulators alive until the wait states are over.
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(D[0]), "+v"(D[1])::"memory");
        }
    } else
    for (size_t st = gwave; st < nsuper; st += nwaves) {
        i32x4 A[kTB];
#pragma unroll
        for (int b = 0; b < kTB; b++) A[b] = expand_code_fp4(x[b]);
        load_codes(x, st + nwaves);   // next step's codes travel while this one computes
        const uint32_t off = (uint32_t)(st * kStep);   // row - begin of code tile 0 (the span is < 2^32)
        const uint32_t dlt = strict_rows && begin + st * kStep >= strict_from ? 0x00010001u : 0u;   // wave-uniform
        // Software pipeline over the query tiles: per pair of code tiles, the v_pk_maximum3 folding query tile t-1's
        // results, then the two MFMAs of query tile t into the same registers.  4 MFMAs (128 matrix-pipe cycles)
        // carry 19 vector instructions (76 issue cycles) + their own 32: the matrix pipe is the bound again, and the
        // waves of a SIMD cover each other's stalls.  Issue order is pinned by volatile asm (the scheduler otherwise
        // serialises MFMA bursts and fold bursts).  Hazards are covered by construction, not by compiler nops: a result
        // is first read one whole step (2 MFMAs and a fold, >= 96 cycles) after its last MFMA issued, the accumulating
        // MFMA follows its producer back to back (forwarded), and the MFMA that overwrites a result issues after the
        // fold that read it; B operands come from LDS (waitcnt on the asm operands), A was written by VALU hundreds of
        // cycles earlier.
        auto step = [&](uint32_t t, const i32x4& bq, uint32_t thr0, i32x4& nq_, uint32_t& nthr) {
            uint32_t m01, m23, vs;
            uint64_t hit;
            const uint32_t thr = thr0 + dlt;
            // The "memory" clobbers keep the operand prefetch of the next tile (plain LDS loads: the compiler
            // places their address arithmetic and waitcnt) where it is written, early in the step.
            asm volatile(UCFP_FOLD_PAIR : "+v"(D[0]), "=&v"(m01) : UCFP_FOLD_PAIR_IN(0) : "memory");
            nq_ = QB[(t + 1) * 64 + lane];   // the pad tiles end the array
            nthr = THR[(t + 1) * 32 + nn];
            asm volatile(UCFP_FOLD_PAIR_LAST
                         : "+v"(D[1]), "=&v"(m23), "=&v"(vs), "=s"(hit)
                         : UCFP_FOLD_PAIR_IN(1), "v"(m01), "v"(thr)
                         : "memory");
            // straight-line (in the first stages nearly every step comes through here, and a branch per code tile was most
            // of their time)
            if (__builtin_expect(hit != 0, 0)) log_step(thr, m01, m23, q0 / 32 + (t - 1), off);
        };
        i32x4 p = QB[lane], r;
        // tiles 0 .. ntiles: the last one is a pad tile that only drains the pipeline.  Two steps per
        // trip so that the operand registers ping-pong without moves.  The fold inside step t
        // tests tile t-1, so its threshold lags: tp = thr(t-1), tc = thr(t), tn = thr(t+1).
        uint32_t tp = kNever, tc = THR[nn], tn;
        uint32_t t = 0;
        for (; t + 2 <= ntiles + 1; t += 2) {
            step(t, p, tp, r, tn);
            tp = tc;
            tc = tn;
            step(t + 1, r, tp, p, tn);
            tp = tc;
            tc = tn;
        }
        if (t < ntiles + 1) step(t, p, tp, r, tn);
        // the drain tile's MFMAs may still be writing D: 18 wait states before anything else touches it
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    }
    if (lane == 0) log_cnt[slice] = ln < log_cap ? ln : log_cap;
}

// ---- bound pass: the first upper bounds from the matrix cores ----------------------------------
// The k-th smallest of the minima of G disjoint groups of codes is an upper bound on the k-th smallest distance of their
// union (the k groups below it each hold a code at least that close), and with G = 256 >> k it is the k-th distance itself
// or its neighbour.  So the filter's own loop -- same operands, same packed fold -- run WITHOUT thresholds over the first
// 256 k codes, keeping per (workgroup, query) the largest sum, yields in one launch at the matrix rate what the sample
// histogram (popcounts, 3 T pairs/s) plus two stages of filter / rescan / list threshold used to approach step by step.
// The fold's per-lane maximum goes to LDS with ds_max_u32 ([tile][query]; both 16-row halves of a query hit the same
// word, which also merges them); the workgroup's row of the table is d'(q) = popc(fq) - max sum, 255 for "no code seen".
#define UCFP_FOLD_PAIR_BOUND                                                                             \
    "v_pk_maximum3_f16 %1, %5, %6, %7\n\t"                                                               \
    "v_pk_maximum3_f16 %1, %1, %8, %9\n\t"                                                               \
    "v_pk_maximum3_f16 %1, %1, %10, %11\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %12, %13\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %14, %15\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %16, %17\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %18, %19\n\t"                                                             \
    "v_pk_maximum3_f16 %1, %1, %20, %20\n\t"                                                             \
    "v_mfma_f32_32x32x64_f8f6f4 %0, %3, %4, %21 cbsz:4 blgp:4\n\t"                                       \
    "v_pk_maximum3_f16 %2, %1, %25, %25\n\t"                                                             \
    "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %22, %4, %0, %23, %24 op_sel_hi:[0,0,0] cbsz:4 blgp:4\n\t"     \
    "s_nop 0"

__global__ __launch_bounds__(kMW * 64) void hamming_bound_mfma(
    const uint64_t* __restrict__ codes, size_t end, const uint64_t* __restrict__ queries, uint32_t nq,
    uint8_t* __restrict__ table, uint32_t table_stride, uint32_t* __restrict__ cand_cnt, uint32_t* __restrict__ overflow) {
    const uint32_t nthreads = blockDim.x, mw = nthreads >> 6;
    extern __shared__ __attribute__((aligned(16))) uint8_t mf_lds[];
    const uint32_t q0 = blockIdx.y * kQP;
    const uint32_t nqp = nq - q0 < (uint32_t)kQP ? nq - q0 : (uint32_t)kQP;
    const uint32_t ntiles = (nqp + 31) / 32;
    i32x4* QB = reinterpret_cast<i32x4*>(mf_lds);
    // [1 + tile][32]: largest (1088 + sum) seen for the query; row 0 takes the fold of "tile -1" (pipeline fill); 0 = nothing seen
    uint32_t* MX = reinterpret_cast<uint32_t*>(mf_lds + (size_t)(ntiles + 2) * 1024);
    // (cand_cnt != nullptr: no hamming_bound_tau follows -- the first scan derives the thresholds -- so the candidate lists
    // and the overflow flag are emptied here)
    if (cand_cnt && blockIdx.x == 0 && blockIdx.y == 0) {
        for (uint32_t s = threadIdx.x; s < nq * kSub; s += nthreads) cand_cnt[s] = 0;
        if (threadIdx.x == 0) *overflow = 0;
    }
    build_query_image(QB, queries, nq, q0, ntiles, nthreads);
    for (uint32_t s = threadIdx.x; s < (ntiles + 2) * 32; s += nthreads) MX[s] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int nn = lane & 31, hh = lane >> 5;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t gwave = (size_t)blockIdx.x * mw + wv, nwaves = (size_t)gridDim.x * mw;
    const size_t nsuper = end / kStep;   // whole steps only (the launcher rounds `end` down): a phantom zero code would fake a minimum
    const uint32_t* __restrict__ halves = reinterpret_cast<const uint32_t*>(codes);
    auto load_codes = [&](uint32_t (&x)[kTB], size_t st) {
#pragma unroll
        for (int b = 0; b < kTB; b++) {
            const size_t row = st * kStep + 32 * b + nn;
            x[b] = st < nsuper ? __builtin_nontemporal_load(halves + row * 2 + hh) : 0u;
        }
    };
    f32x16 D[kTB / 2];
#pragma unroll
    for (int p = 0; p < kTB / 2; p++)
#pragma unroll
        for (int e = 0; e < 16; e++) D[p][e] = 0.f;
    f32x16 cc;
#pragma unroll
    for (int e = 0; e < 16; e++) cc[e] = 8388608.f + 4194304.f + (float)kFieldLo;
    asm volatile("" : "+v"(cc));
    const int sa = 127 + 16, sb = 127;
    uint32_t x[kTB];
    load_codes(x, gwave);
    for (size_t st = gwave; st < nsuper; st += nwaves) {
        i32x4 A[kTB];
#pragma unroll
        for (int b = 0; b < kTB; b++) A[b] = expand_code_fp4(x[b]);
        load_codes(x, st + nwaves);
        // as in hamming_scan_mfma: the fold inside step t covers query tile t - 1
        auto step = [&](uint32_t t, const i32x4& bq, i32x4& nq_) {
            uint32_t m01, m23, vs;
            asm volatile(UCFP_FOLD_PAIR : "+v"(D[0]), "=&v"(m01) : UCFP_FOLD_PAIR_IN(0) : "memory");
            nq_ = QB[(t + 1) * 64 + lane];
            asm volatile(UCFP_FOLD_PAIR_BOUND
                         : "+v"(D[1]), "=&v"(m23), "=&v"(vs)
                         : UCFP_FOLD_PAIR_IN(1), "v"(m01)
                         : "memory");
            // fields: high = (0x4B00 + 64) + max sum of tiles 1 / 3, low = 1088 + max sum of tiles 0 / 2
            const uint32_t lo = vs & 0xffffu, hi = (vs >> 16) - (kFieldHi - kFieldLo);
            atomicMax(&MX[t * 32 + nn], lo > hi ? lo : hi);
        };
        i32x4 p = QB[lane], r;
        uint32_t t = 0;
        for (; t + 2 <= ntiles + 1; t += 2) {
            step(t, p, r);
            step(t + 1, r, p);
        }
        if (t < ntiles + 1) step(t, p, r);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < ntiles * 32; s += nthreads) {
        const uint32_t q = q0 + s;
        if (q >= nq) continue;
        uint32_t slack;
        const uint64_t fq = filter_query(queries[q], slack);
        const uint32_t m = MX[32 + s];
        table[(size_t)blockIdx.x * table_stride + q] = m ? (uint8_t)((uint32_t)__popcll(fq) + kFieldLo - m) : (uint8_t)255;
    }
}

// k-th smallest group minimum per query (+ the filter query's slack) -> tau0; empties the candidate lists and the
// overflow flag like hamming_tau0.  Block = 4 waves x 64 queries; the waves split the table's rows.
__global__ __launch_bounds__(256) void hamming_bound_tau(const uint8_t* __restrict__ table, uint32_t groups,
                                                         uint32_t table_stride, const uint64_t* __restrict__ queries,
                                                         uint32_t nq, uint32_t k, uint32_t* __restrict__ tau0,
                                                         uint32_t* __restrict__ cand_cnt, uint32_t* __restrict__ overflow) {
    __shared__ uint32_t h[65 * kWave];   // [bin][query lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x * kWave + lane;
    for (int b = threadIdx.x; b < 65 * kWave; b += 256) h[b] = 0;
    __syncthreads();
    const uint32_t qc = q < nq ? q : nq - 1;
    for (uint32_t g0 = wave; g0 < groups; g0 += 4 * 16) {
        uint32_t d[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const uint32_t g = g0 + 4 * u;
            d[u] = g < groups ? table[(size_t)g * table_stride + qc] : 255u;
        }
#pragma unroll
        for (int u = 0; u < 16; u++)
            if (d[u] <= 64u) atomicAdd(&h[d[u] * kWave + lane], 1u);
    }
    __syncthreads();
    if (wave != 0 || q >= nq) return;
    uint32_t cum = 0, t = 64;
    bool done = false;
    for (int b = 0; b < 65; b++) {
        cum += h[b * kWave + lane];
        if (!done && cum >= k) {
            t = b;
            done = true;
        }
    }
    uint32_t slack;
    (void)filter_query(queries[q], slack);
    t += slack;   // d(q, x) <= d(fq, x) + slack
    tau0[q] = t < 64u ? t : 64u;
    for (uint32_t u = 0; u < kSub; u++) cand_cnt[(size_t)q * kSub + u] = 0;
    if (q == 0) *overflow = 0;
}

// gridDim.y blocks per log slice.  Pass 1, one thread per record: its four tiles' ballots of flagged lanes are
// flattened into an LDS queue of (query, first row) entries; pass 2, one thread per ENTRY: exact distances of the 16 codes
// that lane folded, true candidates appended to the per-query lists.  (A thread that walks its own ballot makes the whole
// wave wait for the longest ballot, one dependent chain of loads and an atomic round trip per flagged lane.)
constexpr uint32_t kRescanQueue = 2048, kRescanThreads = 128;
__global__ __launch_bounds__(kRescanThreads) __attribute__((amdgpu_waves_per_eu(8))) void hamming_rescan(
    const uint64_t* __restrict__ codes, const uint64_t* __restrict__ ids, size_t begin, size_t end,
    const uint64_t* __restrict__ queries, const uint32_t* __restrict__ tau, const uint4* __restrict__ log,
    const uint32_t* __restrict__ log_cnt, uint32_t log_cap, uint32_t* __restrict__ cand_cnt,
    uint32_t* __restrict__ cand_d, uint64_t* __restrict__ cand_id, uint32_t cand_cap, uint32_t* __restrict__ overflow,
    size_t strict_from, const uint32_t* __restrict__ ids_ascending) {
    __shared__ uint2 queue[kRescanQueue];
    const bool strict_rows = ids_ascending && *ids_ascending;   // see hamming_scan_mfma
    __shared__ uint32_t qn;
    const uint32_t cnt = log_cnt[blockIdx.x];
    const uint32_t sub = blockIdx.x % kSub, sub_cap = cand_cap / kSub;
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    // one flagged lane: the 16 results lane (nn, hh) folded are rows (j & 3) + 8 (j >> 2) + 4 hh of the 32-code tile (32x32 C/D map)
    auto evaluate = [&](uint32_t q, uint32_t off) {
        const uint64_t qv = queries[q];
        const size_t row0 = begin + off;
        const int tq = (int)tau[q] - (strict_rows && row0 >= strict_from ? 1 : 0);   // -1: nothing passes
        uint64_t cv[16];
        if (row0 + 28 <= end) {   // four groups of four consecutive codes: 16-byte loads (row0 is a multiple of 4)
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const u64x2 c0 = *reinterpret_cast<const u64x2*>(codes + row0 + 8 * g);
                const u64x2 c1 = *reinterpret_cast<const u64x2*>(codes + row0 + 8 * g + 2);
                cv[4 * g] = c0[0];
                cv[4 * g + 1] = c0[1];
                cv[4 * g + 2] = c1[0];
                cv[4 * g + 3] = c1[1];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const size_t row = row0 + (j & 3) + 8 * (j >> 2);
                cv[j] = codes[row < end ? row : begin];
            }
        }
        // which of the 16 pass: a bit mask first, then ONE append per trip of a loop the whole wave walks together
        // (a branch per position put up to 16 atomic round trips of different lanes in a row: 62 -> 41 us in a first stage)
        uint32_t pm = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const size_t row = row0 + (j & 3) + 8 * (j >> 2);
            const int d = (int)__popcll(qv ^ cv[j]);
            pm |= (d <= tq && row < end) ? 1u << j : 0u;
        }
        // self-check: a flagged lane holds at least one true candidate by construction (rows past `end`
        // are all-zero and can only flag a lane whose threshold is <= 0); anything else means the scan and
        // this kernel disagree about the result layout -> distrust the filter, take the robust tier
        // (the all-ones query is filtered one bit off and one distance wider -- filter_query -- so its lanes may be
        // flagged in vain)
        if (!pm && (int)__popcll(qv) - tq > 0 && qv != ~0ull) *overflow = 1;
        while (pm) {
            const int j = __builtin_ctz(pm);
            pm &= pm - 1;
            uint64_t c = cv[0];
#pragma unroll
            for (int jj = 1; jj < 16; jj++) c = j == jj ? cv[jj] : c;
            const size_t row = row0 + (j & 3) + 8 * (j >> 2);
            const uint64_t id = ids[row];
            const uint32_t pos = atomicAdd(&cand_cnt[(size_t)q * kSub + sub], 1u);
            if (pos < sub_cap) {
                cand_d[(size_t)q * cand_cap + sub * sub_cap + pos] = (uint32_t)__popcll(qv ^ c);
                cand_id[(size_t)q * cand_cap + sub * sub_cap + pos] = id;
            } else {
                *overflow = 1;
            }
        }
    };
    const uint32_t stride = gridDim.y * kRescanThreads;
    for (uint32_t base = blockIdx.y * kRescanThreads; base < cnt; base += stride) {   // block-uniform trip count
        const uint32_t rix = base + threadIdx.x;
        uint4 ra = make_uint4(0, 0, 0, 0), rb = ra, rc = ra;
        if (rix < cnt) {
            const uint4* rec = log + ((size_t)blockIdx.x * log_cap + rix) * 3;
            ra = rec[0];
            rb = rec[1];
            rc = rec[2];
        }
        const uint64_t m0 = (uint64_t)ra.z | ((uint64_t)ra.w << 32), m1 = (uint64_t)rb.x | ((uint64_t)rb.y << 32),
                       m2 = (uint64_t)rb.z | ((uint64_t)rb.w << 32), m3 = (uint64_t)rc.x | ((uint64_t)rc.y << 32);
        const uint32_t nf = (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
        uint32_t pos = nf ? atomicAdd(&qn, nf) : 0u;
#pragma unroll
        for (uint32_t b = 0; b < (uint32_t)kTB; b++) {
            uint64_t mask = b == 0 ? m0 : b == 1 ? m1 : b == 2 ? m2 : m3;
            while (mask) {
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                // a flagged lane always has a live query; row offset of the lane's first code (a multiple of 4)
                const uint32_t q = ra.x * 32 + (l & 31), off = ra.y + 32 * b + 4 * (l >> 5);
                if (pos < kRescanQueue) queue[pos] = make_uint2(q, off);
                else evaluate(q, off);   // a trip that flags more than the queue holds (dense first stages of tiny corpora)
                pos++;
            }
        }
        __syncthreads();
        // snapshot the counter between two barriers: a wave that decides not to drain goes straight on to the next trip's
        // atomicAdd(&qn), and a slower wave reading qn after that would decide differently and wait alone at the drain's barrier
        const uint32_t cur = qn;
        __syncthreads();
        const uint32_t have = cur < kRescanQueue ? cur : kRescanQueue;
        const bool last = base + stride >= cnt;
        if (last || have + kRescanThreads * 8 > kRescanQueue) {
            for (uint32_t e = threadIdx.x; e < have; e += kRescanThreads) evaluate(queue[e].x, queue[e].y);
            __syncthreads();
            if (threadIdx.x == 0) qn = 0;
            __syncthreads();
        }   // (entries beyond the queue's end were evaluated in place, and a full queue always drains)
    }
}

// ---- few queries: lane = code ---------------------------------------------------------------
// With a handful of queries the matrix cores have nothing to amortise the operand expansion over and
// the scan is a pure HBM stream: each lane holds 8 codes (4 x 16-byte loads in flight), the queries
// and their thresholds are wave-uniform (SGPRs), a pair costs 2 v_xor + 2 v_bcnt + 1 compare, and the
// rare candidate is appended to its query's list directly.  Same staging and lists as the MFMA filter.
constexpr int kFewQueries = 64;   // most queries the lane-per-code scan takes (its histogram's LDS); see few_queries()
// Which filter a batch takes.  The lane-per-code scan costs ~ n x nq popcount work on top of ~0.1 ms of staging launches, the
// matrix filter ~0.15 ms (10 M codes) .. 0.37 ms (100 M) nearly flat up to 100 queries (its longer chain of launches:
// query image, rescans): measured crossovers 40 queries at 10 M codes, 20 at 100 M (tools/bench_hamming.py).
static bool few_queries(size_t n, uint32_t nq) {
    // (round 4, 12.5 M codes, us per search lanes | matrix filter: 9 queries 79 | 90, 16: 92 | 89, 32: 124 | 90 -- the bound pass
    // of round 3 made the matrix filter's chain shorter, so it takes over earlier than the 40 measured in round 2)
    // (round 4, after the matrix filter's short chain for batches of up to 256 queries: it is ahead from 9 queries on at every
    // corpus size it applies to -- 0.3 M / 1.25 M / 12.5 M codes, 9 queries 41 / 43 / 66 us against 57 / 54 / 82 on the lanes,
    // 12 queries 34 / 39 / 63 against 49 / 52 / 84; up to 8 queries with k <= 32 go to hamming_direct before they get here)
    uint32_t most = n >= ((size_t)1 << 18) ? 8u : (uint32_t)kFewQueries;
    static const char* ov = getenv("UCFP_HAMMING_FEW");          // (tuning)
    if (ov) most = (uint32_t)atoi(ov);
    return nq <= most;
}

__global__ __launch_bounds__(256) void hamming_sample_hist_lanes(const uint64_t* __restrict__ codes, size_t sample_n,
                                                                 const uint64_t* __restrict__ queries, uint32_t nq,
                                                                 uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[kFewQueries * 65];
    for (uint32_t i = threadIdx.x; i < nq * 65; i += 256) h[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < sample_n; i += (size_t)gridDim.x * 256) {
        const uint64_t c = codes[i];
        for (uint32_t j = 0; j < nq; j++) atomicAdd(&h[j * 65 + (uint32_t)__popcll(c ^ queries[j])], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nq * 65; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

__global__ __launch_bounds__(256) void hamming_scan_lanes(
    const uint64_t* __restrict__ codes, const uint64_t* __restrict__ ids, size_t begin, size_t end,
    const uint64_t* __restrict__ queries, uint32_t nq, const uint32_t* __restrict__ tau,
    uint32_t* __restrict__ cand_cnt, uint32_t* __restrict__ cand_d, uint64_t* __restrict__ cand_id,
    uint32_t cand_cap, uint32_t* __restrict__ overflow) {
    constexpr int kC = 8;   // codes per lane per trip
    const size_t trip = (size_t)gridDim.x * 256 * kC;
    for (size_t base = begin + (size_t)blockIdx.x * 256 * kC; base < end; base += trip) {
        // lane l owns codes base + 2 l + {0, 1} + 512 k'  (k' = 0..3): 16-byte loads, 1 KiB contiguous per wave
        uint64_t c[kC];
        size_t row[kC];
#pragma unroll
        for (int k = 0; k < kC; k += 2) {
            row[k] = base + (size_t)(k / 2) * 512 + 2 * threadIdx.x;
            row[k + 1] = row[k] + 1;
            if (row[k + 1] < end) {
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(codes + row[k]));
                c[k] = v[0];
                c[k + 1] = v[1];
            } else {
                c[k] = row[k] < end ? codes[row[k]] : 0ull;
                c[k + 1] = 0ull;
            }
        }
        for (uint32_t j = 0; j < nq; j++) {
            const uint64_t q = queries[j];   // wave-uniform: scalar loads
            const uint32_t t = tau[j];
            uint32_t d[kC];
            uint32_t best = 64;
#pragma unroll
            for (int k = 0; k < kC; k++) {
                d[k] = (uint32_t)__popcll(c[k] ^ q);
                best = d[k] < best ? d[k] : best;
            }
            if (__any(best <= t)) {
#pragma unroll
                for (int k = 0; k < kC; k++) {
                    if (d[k] <= t && row[k] < end) {
                        const uint32_t sub = blockIdx.x % kSub, sub_cap = cand_cap / kSub;
                        const uint32_t pos = atomicAdd(&cand_cnt[(size_t)j * kSub + sub], 1u);
                        if (pos < sub_cap) {
                            cand_d[(size_t)j * cand_cap + sub * sub_cap + pos] = d[k];
                            cand_id[(size_t)j * cand_cap + sub * sub_cap + pos] = ids[row[k]];
                        } else {
                            *overflow = 1;
                        }
                    }
                }
            }
        }
    }
}

// A query's sub-lists seen as one list of `total` entries: entry v lives in slot slot(v).  The counts are wave-uniform
// (one scalar load of kSub words).
struct SubLists {
    uint32_t end[kSub];   // running totals
    uint32_t total, sub_cap;
    __device__ __forceinline__ SubLists(const uint32_t* __restrict__ cnt, uint32_t cand_cap) {
        sub_cap = cand_cap / kSub;
        uint32_t run = 0;
#pragma unroll
        for (uint32_t u = 0; u < kSub; u++) {
            const uint32_t c = cnt[u];
            run += c < sub_cap ? c : sub_cap;
            end[u] = run;
        }
        total = run;
    }
    __device__ __forceinline__ uint32_t slot(uint32_t v) const {
        uint32_t u = 0, before = 0;
#pragma unroll
        for (uint32_t i = 0; i + 1 < kSub; i++) {
            const bool past = v >= end[i];
            u += past ? 1u : 0u;
            before = past ? end[i] : before;
        }
        return u * sub_cap + (v - before);
    }
};

// tau1[q] = k-th smallest distance among q's candidates so far (one wave per query); also the exact
// bound the second stage filters with.  Fewer than k candidates (only after an overflow) -> keep tau0.
__global__ __launch_bounds__(64) void hamming_list_tau(const uint32_t* __restrict__ cand_cnt,
                                                       const uint32_t* __restrict__ cand_d, uint32_t cand_cap,
                                                       uint32_t k, const uint32_t* __restrict__ tau0,
                                                       uint32_t* __restrict__ tau1, const uint32_t* __restrict__ ids_ascending) {
    __shared__ uint32_t h[65];
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    h[lane] = 0;
    if (lane == 0) h[64] = 0;
    wave_lds_sync();
    const SubLists sl(cand_cnt + (size_t)q * kSub, cand_cap);
    for (uint32_t c = lane; c < sl.total; c += kWave) atomicAdd(&h[cand_d[(size_t)q * cand_cap + sl.slot(c)]], 1u);
    wave_lds_sync();
    if (lane == 0) {
        uint32_t cum = 0, t = tau0[q];   // fewer than k candidates so far: the previous threshold stands
        for (uint32_t b = 0; b < 65; b++) {
            cum += h[b];
            if (cum >= k) {
                // b is the distance of the k-th best (d, id) so far.  Where record ids ascend with the row number, every
                // row still to come has a larger id than the current k-th, so a tie at distance b cannot displace it:
                // only d < b can still enter.  That cuts the fat boundary bin -- most of a stage's suspects -- away.
                t = (ids_ascending && *ids_ascending && b > 0) ? b - 1 : b;
                break;
            }
        }
        tau1[q] = t;
    }
}

// One wave per query: best k of the candidate list, order (d, id) ascending.  A distance histogram
// gives d* (the k-th smallest distance); only entries with d <= d* can win, and they are few, so
// they are compacted into LDS and the k selection rounds run there (straight from the global list
// when more than kSelCap of them tie).
constexpr int kSelCap = 512;   // 6 KiB of LDS per query: all 4096 one-wave workgroups of a batch are resident at once
__global__ __launch_bounds__(64) void hamming_final_select(
    const uint32_t* __restrict__ cand_cnt, const uint32_t* __restrict__ cand_d, const uint64_t* __restrict__ cand_id,
    uint32_t cand_cap, uint32_t k, uint64_t* __restrict__ out_ids, uint32_t* __restrict__ out_d,
    uint32_t* __restrict__ out_cnt, uint32_t regs_ok) {
    __shared__ uint32_t h[65];
    __shared__ uint32_t sd[kSelCap];
    __shared__ uint64_t si[kSelCap];
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    h[lane] = 0;
    if (lane == 0) h[64] = 0;
    wave_lds_sync();
    const SubLists sl(cand_cnt + (size_t)q * kSub, cand_cap);
    const uint32_t nc = sl.total;
    const uint32_t* __restrict__ gd = cand_d + (size_t)q * cand_cap;
    const uint64_t* __restrict__ gi = cand_id + (size_t)q * cand_cap;
    // A list of up to 8 entries per lane -- nearly every list -- is read ONCE, distances and ids together, into registers: the
    // histogram pass, the compaction's second read of the distances and its dependent read of the ids were three global
    // round trips in a row of a kernel that is one wave per query (10.7 us at 32 queries, most of it those).
    constexpr uint32_t kRegs = 8;
    const bool in_regs = regs_ok && nc <= kRegs * kWave;
    uint32_t rd[kRegs];
    uint64_t ri[kRegs];
    if (in_regs) {
#pragma unroll
        for (uint32_t u = 0; u < kRegs; u++) {
            const uint32_t c = u * kWave + lane;
            const uint32_t at = c < nc ? sl.slot(c) : 0u;
            rd[u] = c < nc ? gd[at] : 0xffffffffu;
            ri[u] = c < nc ? gi[at] : ~0ull;
        }
#pragma unroll
        for (uint32_t u = 0; u < kRegs; u++)
            if (rd[u] <= 64u) atomicAdd(&h[rd[u]], 1u);
    } else {
        for (uint32_t c = lane; c < nc; c += kWave) atomicAdd(&h[gd[sl.slot(c)]], 1u);
    }
    wave_lds_sync();
    // d* = the first bin at which the running count reaches k (64 if it never does: fewer than k candidates): lane b takes bin b,
    // a wave scan replaces 65 dependent LDS reads
    uint32_t dstar;
    {
        uint32_t inc = h[lane];
#pragma unroll
        for (int dl = 1; dl < kWave; dl <<= 1) {
            const uint32_t o = __shfl_up(inc, dl, kWave);
            if (lane >= dl) inc += o;
        }
        const uint64_t reach = __ballot(inc >= k);
        dstar = reach ? (uint32_t)__builtin_ctzll(reach) : 64u;
    }
    // compact the possible winners
    uint32_t m = 0;   // wave-uniform
    if (in_regs) {
        static_assert(kRegs * kWave <= (uint32_t)kSelCap, "a register-held list fits the LDS selection buffer");
#pragma unroll
        for (uint32_t u = 0; u < kRegs; u++) {
            const bool w = rd[u] <= dstar;   // (an empty place holds 0xffffffff)
            const uint64_t mask = __ballot(w);
            const uint32_t pos = m + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (w) {
                sd[pos] = rd[u];
                si[pos] = ri[u];
            }
            m += (uint32_t)__popcll(mask);
        }
    } else {
        for (uint32_t c0 = 0; c0 < nc; c0 += kWave) {
            const uint32_t c = c0 + lane;
            const uint32_t at = c < nc ? sl.slot(c) : 0u;
            const uint32_t dd = c < nc ? gd[at] : 0xffffffffu;
            const bool w = dd <= dstar;
            const uint64_t mask = __ballot(w);
            const uint32_t pos = m + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (w && pos < (uint32_t)kSelCap) {
                sd[pos] = dd;
                si[pos] = gi[at];
            }
            m += (uint32_t)__popcll(mask);
        }
    }
    wave_lds_sync();
    const bool in_lds = m <= (uint32_t)kSelCap;
    const uint32_t total = in_lds ? m : nc;
    uint32_t ld = 0, emitted = 0;
    uint64_t li = 0;
    bool first = true;
    for (uint32_t r = 0; r < k; r++) {
        uint32_t bd = 0xffffffffu;
        uint64_t bi = ~0ull;
        for (uint32_t c = lane; c < total; c += kWave) {
            const uint32_t at = in_lds ? c : sl.slot(c);
            const uint32_t dd = in_lds ? sd[c] : gd[at];
            const uint64_t ii = in_lds ? si[c] : gi[at];
            if ((first || key_less(ld, li, dd, ii)) && key_less(dd, ii, bd, bi)) {
                bd = dd;
                bi = ii;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t od = __shfl_xor(bd, off, kWave);
            const uint64_t oi = __shfl_xor(bi, off, kWave);
            if (key_less(od, oi, bd, bi)) {
                bd = od;
                bi = oi;
            }
        }
        if (bd == 0xffffffffu) break;
        if (lane == 0) {
            out_ids[(size_t)q * k + r] = bi;
            out_d[(size_t)q * k + r] = bd;
        }
        ld = bd;
        li = bi;
        first = false;
        emitted++;
    }
    if (lane == 0) {
        for (uint32_t r = emitted; r < k; r++) {
            out_ids[(size_t)q * k + r] = ~0ull;
            out_d[(size_t)q * k + r] = 0xffffffffu;
        }
        out_cnt[q] = emitted;
    }
}

__global__ void hamming_reset_lists(uint32_t nq, uint32_t* __restrict__ cand_cnt, uint32_t* __restrict__ overflow) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0) *overflow = 0;
    if (q < nq)
        for (uint32_t u = 0; u < kSub; u++) cand_cnt[(size_t)q * kSub + u] = 0;
}

// score = 1 - d/64 (higher is better, src/core/mod.rs:113-115); invalid -> 0 count handles it
__global__ void hamming_scores(const uint32_t* __restrict__ dist, size_t total, float* __restrict__ scores) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const uint32_t d = dist[i];
        scores[i] = d == 0xffffffffu ? -1.0f : 1.0f - (float)d * (1.0f / 64.0f);
    }
}

// ---- launch plan ---------------------------------------------------------------------------

namespace {
void slice_range(size_t n, uint32_t qgroups, uint32_t want_waves, size_t min_per_slice, uint32_t& slices,
                 size_t& per_slice) {
    uint32_t s = (want_waves + qgroups - 1) / (qgroups ? qgroups : 1);
    const size_t max_s = (n + min_per_slice - 1) / min_per_slice;
    if (s > max_s) s = (uint32_t)(max_s ? max_s : 1);
    if (s < 1) s = 1;
    if (s > 8192) s = 8192;
    per_slice = (n + s - 1) / s;
    per_slice = (per_slice + 31) & ~(size_t)31;  // whole 2 x 16-code trips, 256-B aligned loads
    slices = (uint32_t)((n + per_slice - 1) / (per_slice ? per_slice : 1));
    if (slices < 1) slices = 1;
}
}  // namespace

// state[0] = 1 while the record ids of a shard ascend with the row number; state[2..3] = the last id (u64).
// Called for every appended block (`first`: the shard was empty); rows are only ever appended to such a shard.
__global__ void ids_order_update_kernel(const uint64_t* __restrict__ ids, size_t n, bool first, uint32_t* __restrict__ state) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t* last = reinterpret_cast<uint64_t*>(state + 2);
    bool ok;
    if (i == 0) ok = first || ids[0] > *last;
    else ok = ids[i] > ids[i - 1];
    if (!ok) state[0] = 0;
}
__global__ void ids_order_last_kernel(const uint64_t* __restrict__ ids, size_t n, uint32_t* __restrict__ state) {
    *reinterpret_cast<uint64_t*>(state + 2) = ids[n - 1];
}
int launch_ids_order_update(const uint64_t* ids, size_t n, bool first, uint32_t* state, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(ids_order_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ids, n, first, state);
    hipLaunchKernelGGL(ids_order_last_kernel, dim3(1), dim3(1), 0, stream, ids, n, state);
    return 0;
}

static bool few_queries(size_t n, uint32_t nq);
constexpr uint32_t kBoundGroups = 256;        // workgroups of the bound pass = groups of codes whose minima bound the k-th distance
// codes the bound pass covers: 2^17 / 2^18 / 2^19 / 2^20 / 2^21 measured on one box at 10 M, 12.5 M and 100 M codes x 256 and 4096
// queries -- 2^18 is 0.5-1.5 % ahead of 2^19 everywhere, 2^17 and 2^20 1-3 % behind (stage growth 8 instead of 4: 2-5 % behind)
constexpr size_t kBoundCodes = (size_t)1 << 18;

HammingPlan hamming_plan(size_t n, uint32_t nq, uint32_t k) {
    HammingPlan p;
    p.qgroups = (nq + kWave - 1) / kWave;
    p.cap = k <= 16 ? 24 : k <= 40 ? 64 : 160;
    if (n == 0) return p;  // empty shard: nothing to launch
    // sample: the first 32k codes. A lane then accepts ~k*n/sample items over the robust range,
    // i.e. a wave leaves its fast path on ~64*k/sample = 2 % of the codes (k = 10).  (Sizing the sample so that a
    // whole number of 4x stages ends exactly at n -- one stage fewer at 10 M and 12.5 M -- measured the same.)
    size_t s = 32768;     // (16 k / 48 k / 64 k measured the same at 10 M, 12.5 M and 100 M codes)
    if (s > n) s = n;
    p.sample_n = s;
    p.sample_parts = (uint32_t)((s + 1023) / 1024);  // short parts: the pre-pass is latency-bound per wave
    p.per_part = (s + p.sample_parts - 1) / p.sample_parts;
    // matrix-core filter once the corpus dwarfs the sample: stages over ranges growing 4x, so a stage
    // admits ~ c k 4 candidates per query (c <= ~5: the boundary distance bin is fat)
    p.fast = n >= (size_t)1 << 18 && n - 1 <= 0xfffffff0u;
    p.robust_n = n;
    // batches on the matrix-core filter take their first bound from the filter itself (hamming_bound_mfma): the k-th
    // smallest of 256 group minima over the first 256 k codes -- as tight as two stages of lists used to make it
    p.bound = p.fast && !few_queries(n, nq) && k <= 64 && !getenv("UCFP_HAMMING_NO_BOUND");
    // A batch of up to 256 queries is a chain of short launches, each worth 4-10 us whatever it does: a bound pass over 2^20
    // codes and ONE stage over everything beat the 2^18 bound + two stages (12.5 M codes, round 4: 16 / 32 / 64 / 128 / 256
    // queries 79 / 75 / 85 / 108 / 152 us -> 70 / 64 / 74 / 95 / 142; 2^19 and 2^21 within 2 us of that).  Only for k <= 16:
    // the k-th of 256 group minima is a loose bound for large k, and the stages are what tightens it.
    const bool short_chain = p.bound && nq <= 256 && k <= 16;
    if (p.bound) {
        size_t bc = short_chain ? (size_t)1 << 20 : kBoundCodes;
        static const char* bl = getenv("UCFP_HAMMING_BOUND_LOG2_SMALL");   // (tuning)
        if (bl && nq <= 256) bc = (size_t)1 << atoi(bl);
        static const char* bg = getenv("UCFP_HAMMING_BOUND_LOG2_LARGE");   // (tuning)
        if (bg && nq > 256) bc = (size_t)1 << atoi(bg);
        p.bound_n = (n < bc ? n : bc) & ~(size_t)(128 - 1);
    }
    if (p.fast) {
        size_t e = p.bound ? p.bound_n : p.sample_n;
        // ranges grow 4x per stage (measured 2 / 3 / 4 / 6 / 8 / 16 / 32 at 10 M, 12.5 M and 100 M codes x 4096 queries:
        // 4 is fastest everywhere -- tighter thresholds mean fewer suspect blocks to rescan than a stage costs; the same
        // for batches of 9 .. 256 queries, where 16x measured 5-20 % slower)
        // ... and 8 for batches of up to 256 queries, whose searches are chains of short launches (round 4, 12.5 M codes:
        // 16 / 64 / 128 queries 99 / 110 / 124 us at 4, 89 / 99 / 116 at 8, 91 / 101 / 116 at 16)
        size_t growth = short_chain ? 64 : nq <= 256 ? 8 : 4;
        static const char* gs = getenv("UCFP_HAMMING_GROWTH_SMALL");   // (tuning)
        if (gs && nq <= 256) growth = (size_t)atoi(gs);
        static const char* gl = getenv("UCFP_HAMMING_GROWTH_LARGE");   // (tuning)
        if (gl && nq > 256) growth = (size_t)atoi(gl);
        // (with the bound pass the first stage starts over at row 0, so there is one even when bound_n == n)
        // (a last stage over a remainder below a quarter of the stage before it is not worth its launches -- scan, rescan and
        // threshold kernels, 35-60 us at 4096 queries -- nor its under-filled grid: 1.25 M codes x 4096 queries ran 2^20 + 0.2 M
        // in 103 + 33 us of scans; the stage before takes the remainder along)
        static const char* nomerge = getenv("UCFP_HAMMING_NO_STAGE_MERGE");   // (measurement)
        do {
            e = e * growth < n ? e * growth : n;
            if (!nomerge && n - e < e / 4) e = n;
            p.stage_end[p.nstages++] = e;
        } while (e < n && p.nstages < 12);
        p.stage_end[p.nstages - 1] = n;
        p.robust_n = 0;
        size_t cc = (size_t)k * growth * 5 * p.nstages * 2;   // 2x headroom
        if (cc < 2048) cc = 2048;
        if (cc > 65536) cc = 65536;
        cc = (cc + 63) & ~(size_t)63;   // kSub equal sub-lists
        p.cand_cap = (uint32_t)cc;
        p.log_cap = 1024;   // suspect steps per scan wave: 4096 slices x 1024 records x 48 B
        slice_range(n, p.qgroups, 256 * 16, 4096, p.fb_slices, p.fb_per_slice);
        return p;
    }
    slice_range(p.robust_n, p.qgroups, 256 * 16, 4096, p.slices, p.per_slice);
    return p;
}

namespace {
struct HammingWs {
    size_t btab, hist, tau0, part_ids, part_d, part_cnt, tau1, tau2, cand_cnt, overflow, cand_d, cand_id, log_cnt, log, total;
};
HammingWs hamming_ws_layout(const HammingPlan& p, uint32_t nq, uint32_t k) {
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    HammingWs w;
    size_t off = 0;
    const uint32_t ms = p.slices > p.fb_slices ? p.slices : p.fb_slices;
    w.btab = off;      off = align(off + (p.bound ? (size_t)kBoundGroups * p.qgroups * kWave : 0));
    w.hist = off;      off = align(off + (size_t)(p.sample_parts ? p.sample_parts : 1) * 65 * p.qgroups * kWave * 4);
    w.tau0 = off;      off = align(off + (size_t)nq * 4);
    w.part_ids = off;  off = align(off + (size_t)ms * nq * k * 8);
    w.part_d = off;    off = align(off + (size_t)ms * nq * k * 4);
    w.part_cnt = off;  off = align(off + (size_t)ms * nq * 4);
    w.tau1 = off;      off = align(off + (size_t)nq * 4);
    w.tau2 = off;      off = align(off + (size_t)nq * 4);
    w.cand_cnt = off;  off = align(off + (size_t)nq * 4 * kSub);
    w.overflow = off;  off = align(off + 4);
    w.cand_d = off;    off = align(off + (p.fast ? (size_t)nq * p.cand_cap * 4 : 0));
    w.cand_id = off;   off = align(off + (p.fast ? (size_t)nq * p.cand_cap * 8 : 0));
    w.log_cnt = off;   off = align(off + (p.fast ? (size_t)hamming_log_slices(nq) * 4 : 0));
    w.log = off;       off = align(off + (p.fast ? (size_t)hamming_log_slices(nq) * p.log_cap * 48 : 0));
    w.total = off;
    return w;
}

template <typename... Args>
void launch_robust(int cap, dim3 grid, hipStream_t stream, Args... args) {
    if (cap == 24) hipLaunchKernelGGL(hamming_scan<24>, grid, dim3(64), 0, stream, args...);
    else if (cap == 64) hipLaunchKernelGGL(hamming_scan<64>, grid, dim3(64), 0, stream, args...);
    else hipLaunchKernelGGL(hamming_scan<160>, grid, dim3(64), 0, stream, args...);
}
}  // namespace

size_t hamming_workspace_bytes(const HammingPlan& p, uint32_t nq, uint32_t k) {
    return hamming_ws_layout(p, nq, k).total + 1024;
}

int launch_hamming_search(const uint64_t* codes, const uint64_t* ids, size_t n,
                          const uint64_t* queries, uint32_t nq, uint32_t k, uint8_t* ws,
                          const HammingPlan& p, uint64_t* out_ids, uint32_t* out_dist,
                          float* out_scores, uint32_t* out_cnt, hipStream_t stream, const uint32_t* ids_ascending) {
    if (nq == 0) return 0;
    if (nq > kHammingMaxBatch) return -1;   // callers chunk (index.hip)
    const HammingWs w = hamming_ws_layout(p, nq, k);
    auto u32 = [&](size_t off) { return reinterpret_cast<uint32_t*>(ws + off); };
    auto u64 = [&](size_t off) { return reinterpret_cast<uint64_t*>(ws + off); };
    const unsigned score_blocks = (unsigned)(((size_t)nq * k + 255) / 256);

    if (n == 0) {
        // empty shard: every list is empty
        (void)hipMemsetAsync(out_ids, 0xff, (size_t)nq * k * 8, stream);
        (void)hipMemsetAsync(out_dist, 0xff, (size_t)nq * k * 4, stream);
        (void)hipMemsetAsync(out_cnt, 0, (size_t)nq * 4, stream);
        if (out_scores)
            hipLaunchKernelGGL(hamming_scores, dim3(score_blocks), dim3(256), 0, stream, out_dist,
                               (size_t)nq * k, out_scores);
        return 0;
    }
    // tau0: from the bound pass (matrix-core batches), else from the sample
    const bool few = few_queries(n, nq);
    // batches of up to 256 queries: the first scan derives its thresholds from the bound table itself (one launch fewer)
    static const bool no_fused_tau = getenv("UCFP_HAMMING_NO_FUSED_TAU") != nullptr;   // (bisecting)
    const bool fused_tau = p.bound && nq <= 256 && !no_fused_tau;
    if (p.bound) {
        const size_t lds = hamming_mfma_lds_bytes(nq);
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hamming_bound_mfma),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const size_t steps = p.bound_n / kStep;
        unsigned mw = steps >= (size_t)kBoundGroups * kMW ? kMW : steps >= (size_t)kBoundGroups * 8 ? 8 : 4;
        const uint32_t stride = p.qgroups * kWave;
        hipLaunchKernelGGL(hamming_bound_mfma, dim3(kBoundGroups, (nq + kQP - 1) / kQP), dim3(mw * 64), lds, stream, codes,
                           p.bound_n, queries, nq, ws + w.btab, stride, fused_tau ? u32(w.cand_cnt) : (uint32_t*)nullptr,
                           u32(w.overflow));
        if (!fused_tau)
            hipLaunchKernelGGL(hamming_bound_tau, dim3(p.qgroups), dim3(256), 0, stream, (const uint8_t*)(ws + w.btab),
                               kBoundGroups, stride, queries, nq, k, u32(w.tau0), u32(w.cand_cnt), u32(w.overflow));
    } else if (few) {
        (void)hipMemsetAsync(u32(w.hist), 0, (size_t)nq * 65 * 4, stream);
        hipLaunchKernelGGL(hamming_sample_hist_lanes, dim3((unsigned)((p.sample_n + 1023) / 1024)), dim3(256), 0, stream,
                           codes, p.sample_n, queries, nq, u32(w.hist));
        hipLaunchKernelGGL(hamming_tau0, dim3((nq + 255) / 256), dim3(256), 0, stream, (const uint32_t*)u32(w.hist), nq,
                           k, 1u, 65u, u32(w.tau0), p.fast ? u32(w.cand_cnt) : (uint32_t*)nullptr, u32(w.overflow));
    } else {
        hipLaunchKernelGGL(hamming_sample_hist, dim3(p.sample_parts, p.qgroups), dim3(256), 0, stream, codes,
                           p.sample_n, p.per_part, queries, nq, u32(w.hist));
        const uint32_t nqp = p.qgroups * kWave;
        hipLaunchKernelGGL(hamming_hist_reduce, dim3((65 * nqp + 255) / 256), dim3(256), 0, stream, u32(w.hist),
                           p.sample_parts, nqp);
        hipLaunchKernelGGL(hamming_tau0, dim3((nq + 63) / 64), dim3(64), 0, stream, (const uint32_t*)u32(w.hist), nq, k,
                           nqp, 1u, u32(w.tau0), p.fast ? u32(w.cand_cnt) : (uint32_t*)nullptr, u32(w.overflow));
    }
    if (!p.fast) {
        // robust tier over the whole (small) corpus: exact top-k straight into the outputs
        launch_robust(p.cap, dim3(p.slices, p.qgroups), stream, codes, ids, n, p.per_slice, queries, nq, k,
                      (const uint32_t*)u32(w.tau0), u64(w.part_ids), u32(w.part_d), u32(w.part_cnt),
                      (const uint32_t*)nullptr);
        launch_topk_merge_u32(u64(w.part_ids), u32(w.part_d), p.slices, nq, k, out_ids, out_dist, out_cnt, nullptr,
                              stream);
    } else {
        const uint32_t passes = (nq + kQP - 1) / kQP;
        static const uint32_t stream_tiles = getenv("UCFP_HAMMING_NO_STREAM") ? 0u : (uint32_t)kStreamTiles;   // (bisecting)
        static const uint32_t select_regs = getenv("UCFP_HAMMING_NO_SELECT_REGS") ? 0u : 1u;
        const size_t lds = hamming_mfma_lds_bytes(nq);
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hamming_scan_mfma),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        // stage thresholds alternate between tau1 and tau2: tau0 (the sample's, never strict) stays intact for the
        // fallback tier, which filters the WHOLE corpus with d <= tau0 -- a strict stage threshold would drop the k-th itself
        uint32_t* tau_cur = u32(w.tau0);
        uint32_t* tau_nxt = u32(w.tau1);
        uint32_t* tau_spare = u32(w.tau2);
        size_t begin = 0;
        for (uint32_t sidx = 0; sidx < p.nstages; sidx++) {
            const size_t end = p.stage_end[sidx];
            const size_t supers = (end - begin + kStep - 1) / kStep;
            // one workgroup per CU (the query image fills its LDS) of 4, 8 or 16 waves -- as many as the stage has steps for:
            // the waves of a SIMD share its matrix pipe and cover each other's stalls (a step takes 17.8 us with one wave per
            // SIMD, 11.9 us per wave with four), and steps are dealt wave-major so that a partial round costs its share
            unsigned mw = supers > 256 * 8 ? (unsigned)kMW : supers > 256 * 4 ? 8u : 4u;
            unsigned wgs = 256;
            if ((size_t)wgs * mw > supers) wgs = (unsigned)((supers + mw - 1) / mw);
            // suspects are dense in the short first stages (every step logs a record): more rescan blocks per slice there
            const unsigned rescan_parts = supers <= 4096 ? 4 : 1;
            const size_t strict_from = p.bound && sidx == 0 ? p.bound_n : ~(size_t)0;   // rows behind the bound pass's range
            if (few) {
                const size_t per_block = 256 * 8;
                size_t blocks = (end - begin + per_block - 1) / per_block;
                if (blocks > 256 * 8) blocks = 256 * 8;
                hipLaunchKernelGGL(hamming_scan_lanes, dim3((unsigned)blocks), dim3(256), 0, stream, codes, ids, begin,
                                   end, queries, nq, (const uint32_t*)tau_cur, u32(w.cand_cnt), u32(w.cand_d),
                                   u64(w.cand_id), p.cand_cap, u32(w.overflow));
            } else {
            // every wave of the scan writes its slice's record count, and the slices of a launch are 0 .. wgs * passes *
            // mw - 1: the rescan covers exactly those (no memset of the counters)
            const bool derive = fused_tau && sidx == 0;
            hipLaunchKernelGGL(hamming_scan_mfma, dim3(wgs, passes), dim3(mw * 64), lds + (derive ? kFusedTauLds : 0), stream,
                               codes, begin, end, queries, nq, (const uint32_t*)tau_cur,
                               reinterpret_cast<uint4*>(ws + w.log), u32(w.log_cnt), p.log_cap, u32(w.overflow), strict_from,
                               ids_ascending, stream_tiles, derive ? (const uint8_t*)(ws + w.btab) : (const uint8_t*)nullptr,
                               kBoundGroups, p.qgroups * kWave, k, u32(w.tau0));
            hipLaunchKernelGGL(hamming_rescan, dim3(wgs * passes * mw, rescan_parts), dim3(kRescanThreads), 0, stream, codes, ids, begin, end, queries,
                               (const uint32_t*)tau_cur, reinterpret_cast<const uint4*>(ws + w.log),
                               (const uint32_t*)u32(w.log_cnt), p.log_cap, u32(w.cand_cnt), u32(w.cand_d),
                               u64(w.cand_id), p.cand_cap, u32(w.overflow), strict_from, ids_ascending);
            }
            if (sidx + 1 < p.nstages) {
                hipLaunchKernelGGL(hamming_list_tau, dim3(nq), dim3(64), 0, stream, (const uint32_t*)u32(w.cand_cnt),
                                   (const uint32_t*)u32(w.cand_d), p.cand_cap, k, (const uint32_t*)tau_cur, tau_nxt,
                                   ids_ascending);
                uint32_t* t = tau_cur == u32(w.tau0) ? tau_spare : tau_cur;
                tau_cur = tau_nxt;
                tau_nxt = t;
            }
            begin = end;
        }
        hipLaunchKernelGGL(hamming_final_select, dim3(nq), dim3(64), 0, stream, (const uint32_t*)u32(w.cand_cnt),
                           (const uint32_t*)u32(w.cand_d), (const uint64_t*)u64(w.cand_id), p.cand_cap, k, out_ids,
                           out_dist, out_cnt, select_regs);
        // fallback: only runs (device-side check) when some candidate list overflowed
        launch_robust(p.cap, dim3(p.fb_slices, p.qgroups), stream, codes, ids, n, p.fb_per_slice, queries, nq, k,
                      (const uint32_t*)u32(w.tau0), u64(w.part_ids), u32(w.part_d), u32(w.part_cnt),
                      (const uint32_t*)u32(w.overflow));
        // (the gated merge also turns the final distances into scores, whichever kernel selected them)
        launch_topk_merge_u32(u64(w.part_ids), u32(w.part_d), p.fb_slices, nq, k, out_ids, out_dist, out_cnt,
                              u32(w.overflow), stream, out_scores);
        return 0;
    }
    if (out_scores)
        hipLaunchKernelGGL(hamming_scores, dim3(score_blocks), dim3(256), 0, stream, out_dist, (size_t)nq * k,
                           out_scores);
    return 0;
}

int launch_hamming_scores(const uint32_t* dist, size_t total, float* scores, hipStream_t stream) {
    if (total == 0) return 0;
    hipLaunchKernelGGL(hamming_scores, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dist, total,
                       scores);
    return 0;
}

}  // namespace ucfp
