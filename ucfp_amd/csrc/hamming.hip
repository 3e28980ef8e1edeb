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

// ---- fast tier ---------------------------------------------------------------------------
// Once an exact top-k of a corpus PREFIX is known, its k-th distance tau1 bounds the final
// k-th from above, so the rest of the corpus only has to be FILTERED: a lane appends the rare
// row with d <= tau1 straight to its query's candidate list in global memory.  No LDS, a dozen
// VGPRs -> 8 waves per SIMD, and the loop is the bare xor/bcnt/min3 stream.  Two 16-code SGPR
// buffers ping-pong so the scalar loads of one half hide behind the VALU work of the other.
// If a list overflows (adversarial order: the prefix says nothing about the tail) a flag is
// raised and the robust tier re-does the whole corpus; results never depend on the heuristic.
__global__ void hamming_tau1(const uint32_t* __restrict__ pre_d, const uint32_t* __restrict__ pre_cnt,
                             uint32_t nq, uint32_t k, uint32_t* __restrict__ tau1,
                             uint32_t* __restrict__ cand_cnt, uint32_t* __restrict__ overflow) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0) *overflow = 0;
    if (q >= nq) return;
    tau1[q] = pre_cnt[q] >= k ? pre_d[(size_t)q * k + k - 1] : 64u;
    cand_cnt[q] = 0;
}

__global__ __launch_bounds__(64) void hamming_scan_fast(
    const uint64_t* __restrict__ codes, const uint64_t* __restrict__ ids, size_t begin, size_t end,
    size_t per_slice, const uint64_t* __restrict__ queries, uint32_t nq, const uint32_t* __restrict__ tau1,
    uint32_t* __restrict__ cand_cnt, uint32_t* __restrict__ cand_d, uint64_t* __restrict__ cand_id,
    uint32_t cand_cap, uint32_t* __restrict__ overflow) {
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.y * kWave + lane;
    const bool live = q < nq;
    const uint64_t qv = queries[live ? q : nq - 1];
    const uint32_t qlo = (uint32_t)qv, qhi = (uint32_t)(qv >> 32);
    const int32_t tau = live ? (int32_t)tau1[q] : -1;
    const size_t s0 = begin + (size_t)blockIdx.x * per_slice;
    const size_t s1 = s0 + per_slice < end ? s0 + per_slice : end;
    if (s0 >= s1) return;

    auto append = [&](size_t row, uint32_t d) {
        if ((int32_t)d <= tau) {
            const uint32_t pos = atomicAdd(&cand_cnt[q], 1u);
            if (pos < cand_cap) {
                cand_d[(size_t)q * cand_cap + pos] = d;
                cand_id[(size_t)q * cand_cap + pos] = ids[row];
            } else {
                *overflow = 1;
            }
        }
    };
    constexpr int G = 16;
    auto process = [&](const uint64_t (&buf)[G], size_t base) {
        uint32_t d[G];
        bool any_hit = false;
#pragma unroll
        for (int j = 0; j < G; j++) {
            d[j] = __builtin_popcount(qlo ^ (uint32_t)buf[j]) + __builtin_popcount(qhi ^ (uint32_t)(buf[j] >> 32));
            any_hit |= (int32_t)d[j] <= tau;
        }
        if (__any(any_hit)) {
#pragma unroll
            for (int j = 0; j < G; j++) append(base + j, d[j]);
        }
    };
    uint64_t a[G], b[G];
    size_t i = s0;
    if (i + G <= s1) {
#pragma unroll
        for (int j = 0; j < G; j++) a[j] = codes[i + j];
    }
    // invariant at loop head: a[] holds codes [i, i+G)
    while (i + 2 * G <= s1) {
#pragma unroll
        for (int j = 0; j < G; j++) b[j] = codes[i + G + j];
        process(a, i);
        if (i + 3 * G <= s1) {
#pragma unroll
            for (int j = 0; j < G; j++) a[j] = codes[i + 2 * G + j];
        }
        process(b, i + G);
        i += 2 * G;
    }
    if (i + G <= s1) {
        process(a, i);
        i += G;
    }
    for (; i < s1; i++) {
        const uint64_t x = codes[i];
        const uint32_t d = __builtin_popcount(qlo ^ (uint32_t)x) + __builtin_popcount(qhi ^ (uint32_t)(x >> 32));
        if (__any((int32_t)d <= tau)) append(i, d);
    }
}

// One wave per query: best k of (prefix top-k list) U (candidate list), order (d, id) ascending.
__global__ __launch_bounds__(64) void hamming_final_merge(
    const uint64_t* __restrict__ pre_ids, const uint32_t* __restrict__ pre_d,
    const uint32_t* __restrict__ cand_cnt, const uint32_t* __restrict__ cand_d,
    const uint64_t* __restrict__ cand_id, uint32_t cand_cap, uint32_t nq, uint32_t k,
    uint64_t* __restrict__ out_ids, uint32_t* __restrict__ out_d, uint32_t* __restrict__ out_cnt) {
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    uint32_t nc = cand_cnt[q];
    nc = nc < cand_cap ? nc : cand_cap;
    const uint32_t total = k + nc;
    uint32_t ld = 0;
    uint64_t li = 0;
    bool first = true;
    uint32_t emitted = 0;
    for (uint32_t r = 0; r < k; r++) {
        uint32_t bd = 0xffffffffu;
        uint64_t bi = ~0ull;
        for (uint32_t c = lane; c < total; c += kWave) {
            uint32_t dd;
            uint64_t ii;
            if (c < k) {
                dd = pre_d[(size_t)q * k + c];
                ii = pre_ids[(size_t)q * k + c];
            } else {
                dd = cand_d[(size_t)q * cand_cap + (c - k)];
                ii = cand_id[(size_t)q * cand_cap + (c - k)];
            }
            if (dd == 0xffffffffu) continue;
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

HammingPlan hamming_plan(size_t n, uint32_t nq, uint32_t k) {
    HammingPlan p;
    p.qgroups = (nq + kWave - 1) / kWave;
    p.cap = k <= 16 ? 24 : k <= 40 ? 64 : 160;
    if (n == 0) return p;  // empty shard: nothing to launch
    // sample: the first 32k codes. A lane then accepts ~k*n/sample items over the robust range,
    // i.e. a wave leaves its fast path on ~64*k/sample = 2 % of the codes (k = 10).
    size_t s = 32768;
    if (s > n) s = n;
    p.sample_n = s;
    p.sample_parts = (uint32_t)((s + 1023) / 1024);  // short parts: the pre-pass is latency-bound per wave
    p.per_part = (s + p.sample_parts - 1) / p.sample_parts;
    // two tiers once the corpus is big enough for the prefix to be a small fraction of it
    p.fast = n >= (size_t)1 << 20;
    p.robust_n = n;
    if (p.fast) {
        size_t pre = n / 32;
        if (pre < 131072) pre = 131072;
        if (pre > ((size_t)4 << 20)) pre = (size_t)4 << 20;
        p.robust_n = (pre + 31) & ~(size_t)31;
        slice_range(n - p.robust_n, p.qgroups, 256 * 32 * 2, 8192, p.fslices, p.fper_slice);
        // expected candidates per query ~ (k .. 5k) * n / prefix (fat boundary bin); 8x headroom
        size_t cc = (size_t)k * 8 * (n / p.robust_n + 1) * 5;
        if (cc < 1024) cc = 1024;
        if (cc > 16384) cc = 16384;
        p.cand_cap = (uint32_t)cc;
        slice_range(n, p.qgroups, 256 * 16, 4096, p.fb_slices, p.fb_per_slice);
    }
    slice_range(p.robust_n, p.qgroups, 256 * 16, 4096, p.slices, p.per_slice);
    return p;
}

namespace {
struct HammingWs {
    size_t hist, tau0, part_ids, part_d, part_cnt, pre_ids, pre_d, pre_cnt, tau1, cand_cnt, overflow, cand_d,
        cand_id, total;
};
HammingWs hamming_ws_layout(const HammingPlan& p, uint32_t nq, uint32_t k) {
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    HammingWs w;
    size_t off = 0;
    const uint32_t ms = p.slices > p.fb_slices ? p.slices : p.fb_slices;
    w.hist = off;      off = align(off + (size_t)nq * 65 * 4);
    w.tau0 = off;      off = align(off + (size_t)nq * 4);
    w.part_ids = off;  off = align(off + (size_t)ms * nq * k * 8);
    w.part_d = off;    off = align(off + (size_t)ms * nq * k * 4);
    w.part_cnt = off;  off = align(off + (size_t)ms * nq * 4);
    w.pre_ids = off;   off = align(off + (size_t)nq * k * 8);
    w.pre_d = off;     off = align(off + (size_t)nq * k * 4);
    w.pre_cnt = off;   off = align(off + (size_t)nq * 4);
    w.tau1 = off;      off = align(off + (size_t)nq * 4);
    w.cand_cnt = off;  off = align(off + (size_t)nq * 4);
    w.overflow = off;  off = align(off + 4);
    w.cand_d = off;    off = align(off + (p.fast ? (size_t)nq * p.cand_cap * 4 : 0));
    w.cand_id = off;   off = align(off + (p.fast ? (size_t)nq * p.cand_cap * 8 : 0));
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
                          float* out_scores, uint32_t* out_cnt, hipStream_t stream) {
    if (nq == 0) return 0;
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
    // tau0 from the sample
    (void)hipMemsetAsync(u32(w.hist), 0, (size_t)nq * 65 * 4, stream);
    hipLaunchKernelGGL(hamming_sample_hist, dim3(p.sample_parts, p.qgroups), dim3(64), 0, stream, codes,
                       p.sample_n, p.per_part, queries, nq, u32(w.hist));
    hipLaunchKernelGGL(hamming_tau0, dim3((nq + 255) / 256), dim3(256), 0, stream, u32(w.hist), nq, k,
                       u32(w.tau0));
    // robust tier over [0, robust_n): exact top-k of the prefix (or of everything)
    uint64_t* r_ids = p.fast ? u64(w.pre_ids) : out_ids;
    uint32_t* r_d = p.fast ? u32(w.pre_d) : out_dist;
    uint32_t* r_cnt = p.fast ? u32(w.pre_cnt) : out_cnt;
    launch_robust(p.cap, dim3(p.slices, p.qgroups), stream, codes, ids, p.robust_n, p.per_slice, queries, nq, k,
                  (const uint32_t*)u32(w.tau0), u64(w.part_ids), u32(w.part_d), u32(w.part_cnt),
                  (const uint32_t*)nullptr);
    launch_topk_merge_u32(u64(w.part_ids), u32(w.part_d), p.slices, nq, k, r_ids, r_d, r_cnt, nullptr, stream);
    if (p.fast) {
        hipLaunchKernelGGL(hamming_tau1, dim3((nq + 255) / 256), dim3(256), 0, stream, r_d, r_cnt, nq, k,
                           u32(w.tau1), u32(w.cand_cnt), u32(w.overflow));
        hipLaunchKernelGGL(hamming_scan_fast, dim3(p.fslices, p.qgroups), dim3(64), 0, stream, codes, ids,
                           p.robust_n, n, p.fper_slice, queries, nq, (const uint32_t*)u32(w.tau1), u32(w.cand_cnt),
                           u32(w.cand_d), u64(w.cand_id), p.cand_cap, u32(w.overflow));
        hipLaunchKernelGGL(hamming_final_merge, dim3(nq), dim3(64), 0, stream, r_ids, r_d, u32(w.cand_cnt),
                           u32(w.cand_d), u64(w.cand_id), p.cand_cap, nq, k, out_ids, out_dist, out_cnt);
        // fallback: only runs (device-side check) when some candidate list overflowed
        launch_robust(p.cap, dim3(p.fb_slices, p.qgroups), stream, codes, ids, n, p.fb_per_slice, queries, nq, k,
                      (const uint32_t*)u32(w.tau0), u64(w.part_ids), u32(w.part_d), u32(w.part_cnt),
                      (const uint32_t*)u32(w.overflow));
        launch_topk_merge_u32(u64(w.part_ids), u32(w.part_d), p.fb_slices, nq, k, out_ids, out_dist, out_cnt,
                              u32(w.overflow), stream);
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
