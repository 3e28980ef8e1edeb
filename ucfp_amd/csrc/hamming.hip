// hamming.hip -- exact brute-force Hamming top-k over 64-bit fingerprints for gfx950.
//
// The reference has no Hamming search (/v1/query is cosine-only,
// src/index/embedded/mod.rs:268-360); this is the new capability BASELINE config 5 puts behind
// that route.  Semantics (DESIGN.md "Hamming spec"): distance d = popcount(q ^ x); results
// ordered by (d ascending, record_id ascending); at most k per query.
//
// Mapping (wave64-first): ONE LANE = ONE QUERY, the corpus code is WAVE-UNIFORM.  A wave walks
// its corpus slice with scalar loads (s_load_dwordx16 = 8 codes), so x sits in SGPRs and the
// per-pair work is 4 VALU ops (2 x v_xor with an SGPR operand, 2 x v_bcnt_u32_b32 chained
// through the accumulator) plus one compare against the lane's running threshold tau.  The
// corpus is read once per 64 queries and is Infinity-Cache resident (<= 100 MB per GPU at
// BASELINE sizes), so with a batch of queries the kernel is VALU-bound, not HBM-bound.
//
// Selection never touches the fast path: a lane appends (d, id) to a private LDS list only when
// d <= tau, and tau starts from an upper bound tau0 obtained from a sample pre-pass (the k-th
// smallest distance inside any subset bounds the k-th over the whole corpus from above), so
// appends are rare.  Lists are pruned wave-synchronously (all lanes run the same selection code
// on their own list), which keeps divergence out of the slow path too.
//
//   hamming_sample_hist   partial (d-histogram) of a corpus sample per query  -> global hist
//   hamming_tau0          k-th smallest sampled distance per query            -> tau0[q]
//   hamming_scan          per (slice, 64-query group): local top-k            -> partial lists
//   topk_merge_u32        per query: merge partial lists by (d, id)           -> final top-k
// The same merge kernel is the last step after the multi-GPU all-gather.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace ucfp {

constexpr int kWave = 64;

__device__ __forceinline__ uint32_t hamming64(uint64_t q, uint64_t x) {
    const uint32_t lo = (uint32_t)q ^ (uint32_t)x;
    const uint32_t hi = (uint32_t)(q >> 32) ^ (uint32_t)(x >> 32);
    return __builtin_popcount(hi) + __builtin_popcount(lo);
}

// ---- sample pre-pass --------------------------------------------------------------------
// grid (parts, qgroups), block 64. hist layout: [q][65] u32 (global, zeroed by the launcher).
__global__ __launch_bounds__(64) void hamming_sample_hist(const uint64_t* __restrict__ codes,
                                                          size_t sample_n, size_t per_part,
                                                          const uint64_t* __restrict__ queries,
                                                          uint32_t nq, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[65 * kWave];  // [bin][lane]: lane-private columns, conflict-free
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.y * kWave + lane;
    for (int b = 0; b < 65; b++) h[b * kWave + lane] = 0;
    const uint64_t qv = queries[q < nq ? q : nq - 1];
    const size_t s0 = (size_t)blockIdx.x * per_part;
    const size_t s1 = s0 + per_part < sample_n ? s0 + per_part : sample_n;
    for (size_t i = s0; i < s1; i++) {
        const uint64_t x = codes[i];  // wave-uniform address -> scalar load
        const uint32_t d = hamming64(qv, x);
        h[d * kWave + lane] += 1;
    }
    if (q < nq) {
        for (int b = 0; b < 65; b++) {
            const uint32_t c = h[b * kWave + lane];
            if (c) atomicAdd(&hist[(size_t)q * 65 + b], c);
        }
    }
}

__global__ void hamming_tau0(const uint32_t* __restrict__ hist, uint32_t nq, uint32_t k,
                             uint32_t* __restrict__ tau0) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint32_t cum = 0, t = 64;
    for (uint32_t b = 0; b < 65; b++) {
        cum += hist[(size_t)q * 65 + b];
        if (cum >= k) {
            t = b;
            break;
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
    uint64_t* __restrict__ part_ids, uint32_t* __restrict__ part_d, uint32_t* __restrict__ part_cnt) {
    __shared__ CandLists<CAP> L;
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

    size_t i = s0;
    // 8 codes per trip: one s_load_dwordx16, 40 VALU ops, one branch
    for (; i + 8 <= s1; i += 8) {
        uint32_t d[8];
        bool any_hit = false;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint64_t x = codes[i + j];
            d[j] = __builtin_popcount(qlo ^ (uint32_t)x) + __builtin_popcount(qhi ^ (uint32_t)(x >> 32));
            any_hit |= (int32_t)d[j] <= tau;
        }
        if (__any(any_hit)) {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (__any((int32_t)d[j] <= tau)) slow(i + j, d[j]);
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

// score = 1 - d/64 (higher is better, src/core/mod.rs:113-115); invalid -> 0 count handles it
__global__ void hamming_scores(const uint32_t* __restrict__ dist, size_t total, float* __restrict__ scores) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const uint32_t d = dist[i];
        scores[i] = d == 0xffffffffu ? -1.0f : 1.0f - (float)d * (1.0f / 64.0f);
    }
}

// ---- launch plan ---------------------------------------------------------------------------

HammingPlan hamming_plan(size_t n, uint32_t nq, uint32_t k) {
    HammingPlan p;
    p.qgroups = (nq + kWave - 1) / kWave;
    p.cap = k <= 16 ? 24 : k <= 40 ? 64 : 160;
    if (n == 0) {  // empty shard: nothing to scan, no partial lists
        p.slices = 0;
        p.sample_parts = 0;
        return p;
    }
    // enough independent waves to fill 256 CUs several times over, slices of >= 4096 codes
    const uint32_t want_waves = 256 * 16;
    uint32_t slices = (want_waves + p.qgroups - 1) / p.qgroups;
    const size_t max_slices = (n + 4095) / 4096;
    if (slices > max_slices) slices = (uint32_t)(max_slices ? max_slices : 1);
    if (slices < 1) slices = 1;
    if (slices > 4096) slices = 4096;
    p.slices = slices;
    p.per_slice = (n + slices - 1) / slices;
    p.per_slice = (p.per_slice + 7) & ~(size_t)7;  // keep 64-byte aligned scalar loads
    p.slices = (uint32_t)((n + p.per_slice - 1) / (p.per_slice ? p.per_slice : 1));
    if (p.slices < 1) p.slices = 1;
    // sample: the first 128k codes. A lane then accepts ~k*n/sample items over the whole corpus,
    // i.e. a wave leaves its fast path on ~64*k/sample = 0.5 % of the codes.
    size_t s = 131072;
    if (s > n) s = n;
    p.sample_n = s;
    p.sample_parts = (uint32_t)((s + 4095) / 4096);
    if (p.sample_parts < 1) p.sample_parts = 1;
    p.per_part = (s + p.sample_parts - 1) / p.sample_parts;
    p.cap = k <= 16 ? 24 : k <= 40 ? 64 : 160;
    return p;
}

size_t hamming_workspace_bytes(const HammingPlan& p, uint32_t nq, uint32_t k) {
    size_t b = 0;
    b += (size_t)nq * 65 * 4;                      // hist
    b += (size_t)nq * 4;                           // tau0
    b += (size_t)p.slices * nq * k * 8;            // part ids
    b += (size_t)p.slices * nq * k * 4;            // part d
    b += (size_t)p.slices * nq * 4;                // part cnt
    return b + 1024;
}

int launch_hamming_search(const uint64_t* codes, const uint64_t* ids, size_t n,
                          const uint64_t* queries, uint32_t nq, uint32_t k, uint8_t* ws,
                          const HammingPlan& p, uint64_t* out_ids, uint32_t* out_dist,
                          float* out_scores, uint32_t* out_cnt, hipStream_t stream) {
    if (nq == 0) return 0;
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws + off);
    off = align(off + (size_t)nq * 65 * 4);
    uint32_t* tau0 = reinterpret_cast<uint32_t*>(ws + off);
    off = align(off + (size_t)nq * 4);
    uint64_t* part_ids = reinterpret_cast<uint64_t*>(ws + off);
    off = align(off + (size_t)p.slices * nq * k * 8);
    uint32_t* part_d = reinterpret_cast<uint32_t*>(ws + off);
    off = align(off + (size_t)p.slices * nq * k * 4);
    uint32_t* part_cnt = reinterpret_cast<uint32_t*>(ws + off);

    if (n == 0) {
        // empty shard: every list is empty
        (void)hipMemsetAsync(out_ids, 0xff, (size_t)nq * k * 8, stream);
        (void)hipMemsetAsync(out_dist, 0xff, (size_t)nq * k * 4, stream);
        (void)hipMemsetAsync(out_cnt, 0, (size_t)nq * 4, stream);
        if (out_scores)
            hipLaunchKernelGGL(hamming_scores, dim3((unsigned)(((size_t)nq * k + 255) / 256)), dim3(256), 0,
                               stream, out_dist, (size_t)nq * k, out_scores);
        return 0;
    }
    (void)hipMemsetAsync(hist, 0, (size_t)nq * 65 * 4, stream);
    hipLaunchKernelGGL(hamming_sample_hist, dim3(p.sample_parts, p.qgroups), dim3(64), 0, stream, codes,
                       p.sample_n, p.per_part, queries, nq, hist);
    hipLaunchKernelGGL(hamming_tau0, dim3((nq + 255) / 256), dim3(256), 0, stream, hist, nq, k, tau0);
    dim3 grid(p.slices, p.qgroups);
    if (p.cap == 24)
        hipLaunchKernelGGL(hamming_scan<24>, grid, dim3(64), 0, stream, codes, ids, n, p.per_slice, queries,
                           nq, k, tau0, part_ids, part_d, part_cnt);
    else if (p.cap == 64)
        hipLaunchKernelGGL(hamming_scan<64>, grid, dim3(64), 0, stream, codes, ids, n, p.per_slice, queries,
                           nq, k, tau0, part_ids, part_d, part_cnt);
    else
        hipLaunchKernelGGL(hamming_scan<160>, grid, dim3(64), 0, stream, codes, ids, n, p.per_slice, queries,
                           nq, k, tau0, part_ids, part_d, part_cnt);
    launch_topk_merge_u32(part_ids, part_d, p.slices, nq, k, out_ids, out_dist, out_cnt, stream);
    if (out_scores)
        hipLaunchKernelGGL(hamming_scores, dim3((unsigned)(((size_t)nq * k + 255) / 256)), dim3(256), 0,
                           stream, out_dist, (size_t)nq * k, out_scores);
    return 0;
}

int launch_hamming_scores(const uint32_t* dist, size_t total, float* scores, hipStream_t stream) {
    if (total == 0) return 0;
    hipLaunchKernelGGL(hamming_scores, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dist, total,
                       scores);
    return 0;
}

}  // namespace ucfp
