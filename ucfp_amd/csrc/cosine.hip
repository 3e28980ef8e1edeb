// cosine.hip -- exact brute-force cosine top-k over f32 embeddings for gfx950.
//
// Replaces EmbeddedBackend::knn phase 2 (src/index/embedded/mod.rs:324-340): for every row v of
// the tenant, score = dot(q, v) / (|q| |v|), rows with |v| = 0 skipped, a query with |q| = 0
// returns nothing (:283-286, :328-330), best k by score.  Tolerance vs the reference's 8-lane
// accumulation order (dot_product :454-472): 1e-5 absolute (BASELINE north_star); ties and the
// NaN case, which the reference leaves to rayon's split order, are fixed here as "ascending
// record_id" and "NaN scores are dropped".
//
// Three steps, all streaming:
//   cosine_norms     |v| per row, once at upsert time (rows are immutable until overwritten)
//   cosine_keys      one pass over the rows per group of QT queries: 8 lanes share a row
//                    (8 x 16 B = one 128-B line per row per step, 8 rows per wave instruction),
//                    query chunks come from LDS by broadcast; the score is mapped to an
//                    order-preserving u32 key (ascending key = descending score) in a scratch
//                    matrix keys[q][row].  HBM-bound for QT <= 16: 4*dim bytes per row.
//   select_topk_u32  (topk.hip) per (slice, query): wave-shared candidate list + threshold
//   topk_merge_u32   merge of the slices (and of the GPUs after the all-gather)

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace ucfp {

constexpr int kQT = 16;  // queries per corpus pass

__device__ __forceinline__ uint32_t score_to_key(float s) {
    // ascending-float order image, then inverted so that a larger score is a smaller key
    uint32_t u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~u;
}
__device__ __forceinline__ float key_to_score(uint32_t k) {
    uint32_t u = ~k;
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

// one wave per row; dim arbitrary (scalar tail).
__global__ __launch_bounds__(256) void cosine_norms(const float* __restrict__ rows, size_t n, uint32_t dim,
                                                    float* __restrict__ norms) {
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* v = rows + row * dim;
    float acc = 0.f;
    for (uint32_t i = lane; i < dim; i += 64) acc = fmaf(v[i], v[i], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) norms[row] = sqrtf(acc);
}

// grid: row tiles of 32 rows per block (4 waves x 8 rows); block 256.
// queries: [QT][dim] (zero-padded by the launcher to QT rows), qnorm[QT].
// keys out: keys[qt * n + row].  Requires dim % 4 == 0 and 16-byte aligned rows for the
// vector path; the tail (dim % 32) is handled by predication on the chunk index.
__global__ __launch_bounds__(256) void cosine_keys(const float* __restrict__ rows, const float* __restrict__ norms,
                                                   size_t n, uint32_t dim, const float* __restrict__ queries,
                                                   const float* __restrict__ qnorm, uint32_t nq_pass,
                                                   uint32_t* __restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [QT][dim4] padded to 16 B
    const uint32_t dim4 = (dim + 3) & ~3u;
    for (uint32_t i = threadIdx.x; i < nq_pass * dim4; i += 256) {
        const uint32_t qt = i / dim4, c = i - qt * dim4;
        qs[i] = (c < dim) ? queries[(size_t)qt * dim + c] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 7, grp = lane >> 3;
    const size_t row = ((size_t)blockIdx.x * 4 + wave) * 8 + grp;
    const bool live = row < n;
    const float* v = rows + (live ? row : 0) * (size_t)dim;
    float acc[kQT];
#pragma unroll
    for (int t = 0; t < kQT; t++) acc[t] = 0.f;
    const bool vec_ok = (dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(rows) & 15u) == 0);
    for (uint32_t c0 = 0; c0 < dim4; c0 += 32) {
        const uint32_t c = c0 + 4 * sub;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && c < dim) {
            if (vec_ok) {
                x = *reinterpret_cast<const float4*>(v + c);
            } else {
                x.x = v[c];
                if (c + 1 < dim) x.y = v[c + 1];
                if (c + 2 < dim) x.z = v[c + 2];
                if (c + 3 < dim) x.w = v[c + 3];
            }
        }
        if (c < dim4) {
#pragma unroll
            for (int t = 0; t < kQT; t++) {
                if ((uint32_t)t >= nq_pass) break;  // wave-uniform
                const float4 q = *reinterpret_cast<const float4*>(&qs[t * dim4 + c]);
                acc[t] = fmaf(x.x, q.x, acc[t]);
                acc[t] = fmaf(x.y, q.y, acc[t]);
                acc[t] = fmaf(x.z, q.z, acc[t]);
                acc[t] = fmaf(x.w, q.w, acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < kQT; t++) {
        acc[t] += __shfl_xor(acc[t], 1, 64);
        acc[t] += __shfl_xor(acc[t], 2, 64);
        acc[t] += __shfl_xor(acc[t], 4, 64);
    }
    if (live && sub == 0) {
        const float vn = norms[row];
#pragma unroll
        for (int t = 0; t < kQT; t++) {
            if ((uint32_t)t < nq_pass) {
                const float qn = qnorm[t];
                uint32_t key = 0xffffffffu;
                if (vn != 0.f && qn != 0.f) {
                    const float s = acc[t] / (qn * vn);
                    if (s == s) key = score_to_key(s);
                }
                keys[(size_t)t * n + row] = key;
            }
        }
    }
}

__global__ void cosine_scores_from_keys(const uint32_t* __restrict__ keys, size_t total, float* __restrict__ scores) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) scores[i] = keys[i] == 0xffffffffu ? -2.0f : key_to_score(keys[i]);
}

int launch_cosine_norms(const float* rows, size_t n, uint32_t dim, float* norms, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(cosine_norms, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, rows, n, dim, norms);
    return 0;
}

// queries per corpus pass: kQT, or fewer when kQT query rows do not fit 144 KiB of LDS
int cosine_queries_per_pass(uint32_t dim) {
    const uint32_t dim4 = (dim + 3) & ~3u;
    const size_t fit = (144u * 1024u) / ((size_t)dim4 * sizeof(float));
    return (int)(fit < (size_t)kQT ? fit : (size_t)kQT);
}

int launch_cosine_keys(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries,
                       const float* qnorm, uint32_t nq_pass, uint32_t* keys, hipStream_t stream) {
    if (n == 0 || nq_pass == 0) return 0;
    const uint32_t dim4 = (dim + 3) & ~3u;
    const size_t lds = (size_t)nq_pass * dim4 * sizeof(float);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cosine_keys),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(cosine_keys, dim3((unsigned)((n + 31) / 32)), dim3(256), lds, stream, rows, norms, n, dim,
                       queries, qnorm, nq_pass, keys);
    return 0;
}

int launch_cosine_scores_from_keys(const uint32_t* keys, size_t total, float* scores, hipStream_t stream) {
    if (total == 0) return 0;
    hipLaunchKernelGGL(cosine_scores_from_keys, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, keys,
                       total, scores);
    return 0;
}

}  // namespace ucfp
