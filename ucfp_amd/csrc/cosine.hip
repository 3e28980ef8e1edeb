// cosine.hip -- exact brute-force cosine top-k over f32 embeddings for gfx950 (every reported score is an f32 dot product).
//
// Replaces EmbeddedBackend::knn phase 2 (src/index/embedded/mod.rs:324-340): for every row v of
// the tenant, score = dot(q, v) / (|q| |v|), rows with |v| = 0 skipped, a query with |q| = 0
// returns nothing (:283-286, :328-330), best k by score.  Tolerance vs the reference's 8-lane
// accumulation order (dot_product :454-472): 1e-5 absolute (BASELINE north_star); ties and the
// NaN case, which the reference leaves to rayon's split order, are fixed here as "ascending
// record_id" and "NaN scores are dropped".
//
// Steps, all streaming:
//   cosine_norms        |v| per row, once at upsert time (rows are immutable until overwritten)
//   keys, four forms by batch shape (launch_cosine_keys / launch_cosine_keys_filtered):
//     cosine_keys_stream<NQ,NB>   1-4 queries, dim <= 1024: a wave reads a row as contiguous 1 KiB loads, the
//                                 queries sit in registers -- HBM-bound, ideal coalescing
//     cosine_keys_mfma<G,FULL>    up to 48 queries (dim % 4 == 0): f32 MFMA tile, query rows resident in LDS
//     cosine_keys_gemm<NG,FILT,RT> batches: 256 queries per corpus read, K slices staged by LDS-DMA; FILT keeps
//                                 only rows that beat a per-query threshold (candidate lists, no key matrix)
//     cosine_keys                 VALU kernel for every other shape: 8 lanes share a row, queries from LDS
//   round 4: batches of 2 .. 64 queries per pass over >= 2^17 rows of dim % 64 == 0 do not compute exact keys for every row:
//     cosine_norms_image + cosine_mins_f16<G>   the smallest APPROXIMATE key of every 16-row chunk through the f16 matrix
//                                 pipe (rows loaded coalesced, transposed through LDS), within cosine_mins_eps of the exact one
//     topk.hip prune_* (eps)      thresholds widened by that bound -> ~k listed chunks per query
//     cosine_keys_mfma<G,.,list>  the exact f32 keys of the listed chunks -> prune_final picks the answer
//   the score is mapped to an order-preserving u32 key (ascending key = descending score), keys[q][row]
//   select_topk_u32       (topk.hip) per (slice, query): wave-shared candidate list + threshold
//   topk_select_lists_u32 (topk.hip) best k of the sample's answer + a candidate list (thresholded pass)
//   topk_merge_u32        merge of the slices (and of the GPUs after the all-gather)

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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

// ---- 5 .. 48 queries without a key matrix ---------------------------------------------------------------------------
// Writing the nq x n key matrix costs far more than its bytes: 64 MB of keys next to 3 GB of row reads (16 queries over
// 1 M x 768) took 140 of the keys kernel's 730 us -- the same stores aimed at a cache-resident slot cost 18 -- and the
// selection then reads them back.  cosine_keys_mfma therefore runs in one of three modes:
//   kKeysDense  keys[q][row] (the fallback, and every other caller)
//   kKeysMins   no keys: the smallest key of every chunk of rows (1 << cs_shift of them) per query, mins[q][chunk]
//   kKeysList   the keys of listed (query, chunk) pairs only, ckeys[entry][row in chunk] -- the SAME arithmetic on the
//               same tiles, so a recomputed key equals the one the minimum was taken from bit for bit
// topk.hip's prune kernels turn the minima into a threshold per query (the k-th smallest chunk minimum: at least k keys
// are <= it, so the k best all are, and each lies in a chunk whose minimum is <= it) and a list of about k chunks, and
// pick the answer from the listed chunks' keys.
enum { kKeysDense = 0, kKeysMins = 1, kKeysList = 2 };
struct CosinePrune {
    uint32_t* mins;          // kKeysMins: [nchunks][qpad] -- a chunk's minima leave as one contiguous run
    uint32_t qpad;           // queries padded to the kernel's tile: 16 G
    uint32_t* wmin;          // kKeysMins: [qpad][waves of the launch]: the smallest key each wave saw per query (the k-th smallest
                             // of these bounds the k-th smallest chunk minimum: k waves hold a key at most that large)
    uint32_t cs_shift;       // rows per chunk = 1 << cs_shift = 16: a tile
    const uint2* list;       // kKeysList: (query, chunk) entries
    const uint32_t* nlist;   // their number (device word)
    uint32_t* ckeys;         // kKeysList: [entry][1 << cs_shift]
    uint32_t slices;         // kKeysList: a batch above the LDS image's 16 G queries is listed once and rescored in `slices`
    uint32_t slice_q;        //            slices of slice_q queries by ONE launch (0 / 1: the whole batch is one image)
};

// Query image [nrows][qstride] in LDS from queries[nq_pass][dim] (16-byte aligned, dim % 4 == 0), zero-filled past dim and
// nq_pass.  Wave w fills rows w, w + waves, ...; a row's pieces are all requested before the first is stored (the fill is a
// few load latencies, not one per piece: it used to be ~25 us of every launch, which matters to the short list pass).
typedef float f32x4q __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fill_query_image(float* __restrict__ qs, const float* __restrict__ queries, uint32_t nrows,
                                                 uint32_t nq_pass, uint32_t dim, uint32_t qstride, uint32_t nthreads) {
    const uint32_t lane = threadIdx.x & 63, waves = nthreads >> 6;
    for (uint32_t qt = threadIdx.x >> 6; qt < nrows; qt += waves) {
        for (uint32_t c0 = 0; c0 < qstride; c0 += 5 * 256) {
            f32x4q v[5];
#pragma unroll
            for (int u = 0; u < 5; u++) {
                const uint32_t c = c0 + 256 * u + 4 * lane;
                v[u] = f32x4q{0.f, 0.f, 0.f, 0.f};
                if (qt < nq_pass && c < dim) v[u] = *reinterpret_cast<const f32x4q*>(queries + (size_t)qt * dim + c);
            }
#pragma unroll
            for (int u = 0; u < 5; u++) {
                const uint32_t c = c0 + 256 * u + 4 * lane;
                if (c < qstride) *reinterpret_cast<f32x4q*>(qs + (size_t)qt * qstride + c) = v[u];
            }
        }
    }
}

// ---- MFMA variant: 16 rows x (16*G queries) per wave step ---------------------------------------
// v_mfma_f32_16x16x4_f32: A = queries (M = query, lane l: Q[l&15][k]), B = rows (N = row, lane l:
// R[l&15][k]), k = 4 consecutive dims per lane taken from ONE float4 (global for rows, LDS for
// queries): lane (n, q) loads R[n][16s+4q .. 16s+4q+3] and feeds element j to MFMA j, while A
// feeds Q[m][16s+4q+j] -- the same permutation of k on both operands, so the sum is complete.
// All 16*G query rows stay in LDS (row stride dim16 + 4 floats: ds_read_b128 conflict-free);
// rows stream from HBM exactly once per pass, 64 B per row per instruction.  HBM-bound up to
// G = 3 (48 queries per pass) at dim 768.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// FULL: dim is a multiple of 32 * U (256): every chunk load is unconditional (dead rows read row 0), so the
// loop body is loads + LDS reads + MFMAs with no exec masking in between.
constexpr int kCW = 8;    // waves per workgroup (one workgroup per CU: the queries fill its LDS); 12 measured the same
template <int G, bool FULL, int MODE>
__global__ __launch_bounds__(kCW * 64) void cosine_keys_mfma(const float* __restrict__ rows,
                                                        const float* __restrict__ norms, size_t n, uint32_t dim,
                                                        const float* __restrict__ queries_all,
                                                        const float* __restrict__ qnorm_all, uint32_t nq_all,
                                                        uint32_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ run_flag, CosinePrune pr) {
    if (run_flag && (MODE == kKeysList ? *run_flag != 0 : *run_flag == 0)) return;   // as in cosine_keys_blocks
    // kKeysList over a batch above the image's 16 G queries: `slices` query slices in ONE launch, workgroup b takes slice
    // b % slices (every slice walks the whole list and stores its own queries' entries); workgroups beyond the list leave
    // before they fill an image (the launch is sized for the longest list the pass can produce).
    uint32_t bid = blockIdx.x, nblocks = gridDim.x, q_base = 0, nq_pass = nq_all;
    const float* __restrict__ queries = queries_all;
    const float* __restrict__ qnorm = qnorm_all;
    if (MODE == kKeysList) {
        if (pr.slices > 1) {
            const uint32_t sl = bid % pr.slices;
            bid /= pr.slices;
            nblocks /= pr.slices;
            q_base = sl * pr.slice_q;
            nq_pass = nq_all - q_base < pr.slice_q ? nq_all - q_base : pr.slice_q;
            queries += (size_t)q_base * dim;
            qnorm += q_base;
        }
        if ((size_t)bid * kCW >= (size_t)*pr.nlist) return;   // workgroup-uniform, before any barrier
    }
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [16*G][dim16 + 4]
    const uint32_t dim16 = (dim + 15) & ~15u;
    const uint32_t qstride = dim16 + 4;
    if ((reinterpret_cast<uintptr_t>(queries) & 15u) == 0) {   // (dim % 4 == 0: mfma_ok)
        fill_query_image(qs, queries, 16u * G, nq_pass, dim, qstride, kCW * 64);
    } else {
        for (uint32_t i = threadIdx.x; i < 16u * G * qstride; i += kCW * 64) {
            const uint32_t qt = i / qstride, c = i - qt * qstride;
            qs[i] = (qt < nq_pass && c < dim) ? queries[(size_t)qt * dim + c] : 0.f;
        }
    }
    // (the barrier behind the fill comes after the first row loads have been requested, below)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nn = lane & 15, q4 = lane >> 4;
    const size_t tiles = (n + 15) / 16;
    // per-pass constants of this lane's 4 G result slots: query norm (0 marks "no such query")
    float qnr[G][4];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t qt = g * 16 + 4 * q4 + r;
            qnr[g][r] = qt < nq_pass ? qnorm[qt] : 0.f;
        }
    // U row chunks (16 floats per row each) are in flight while the previous U are consumed: one block per CU
    // (LDS holds the queries), so latency must be hidden inside the wave -- including across tiles: the first
    // chunks of the NEXT tile are requested before this tile's epilogue.
    constexpr int U = 8;
    float4 xa[U], xb[U];
    const size_t tstep = (size_t)nblocks * kCW;
    auto row_ptr = [&](size_t t) {
        const size_t r = t * 16 + nn;
        return rows + (r < n ? r : 0) * (size_t)dim;   // dead rows read row 0; their results are not stored
    };
    auto load_chunks = [&](float4 (&x)[U], const float* __restrict__ v, bool live, uint32_t c0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t c = c0 + 16 * u + 4 * q4;
            if (FULL) {
                const f32x4v t = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(v + c));
                x[u] = make_float4(t[0], t[1], t[2], t[3]);
            } else {
                x[u] = (live && c < dim) ? *reinterpret_cast<const float4*>(v + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    // (the read one chunk past dim16 at the very end lands in the next query row / the slack behind the
    // last row and is never used)
    auto load_q = [&](float4 (&qv)[G], uint32_t c16) {
#pragma unroll
        for (int g = 0; g < G; g++) qv[g] = *reinterpret_cast<const float4*>(&qs[(g * 16 + nn) * qstride + c16 + 4 * q4]);
    };
    // work items: every 16-row tile, or the tiles of the listed chunks (1 << tsh of them per entry)
    const uint32_t tsh = 0;   // a chunk is one tile (cs_shift == 4)
    const size_t items = MODE == kKeysList ? (size_t)*pr.nlist << tsh : tiles;
    auto tile_of = [&](size_t it) -> size_t {
        if (MODE != kKeysList) return it;
        const size_t t = ((size_t)pr.list[it >> tsh].y << tsh) + (it & (((size_t)1 << tsh) - 1));
        return t < tiles ? t : tiles - 1;   // a chunk's tiles past the last row: recompute the last tile, nobody reads the keys
    };
    size_t it = (size_t)bid * kCW + wave;
    uint32_t wave_min[G][4];   // kKeysMins: smallest key per result slot over this wave's tiles
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) wave_min[g][r] = 0xffffffffu;
    if (it < items) {
        const size_t t0 = tile_of(it);
        load_chunks(xa, row_ptr(t0), t0 * 16 + nn < n, 0);
    }
    __syncthreads();   // the query image is complete
    for (; it < items; it += tstep) {
        const size_t tile = tile_of(it);
        const size_t tile_next = it + tstep < items ? tile_of(it + tstep) : tiles;
        const size_t row = tile * 16 + nn;
        const bool live = row < n;
        const float* __restrict__ v = row_ptr(tile);
        const float vn = norms[live ? row : 0];   // requested now, used in the epilogue
        f32x4v acc[G];
#pragma unroll
        for (int g = 0; g < G; g++) acc[g] = f32x4v{0.f, 0.f, 0.f, 0.f};
        auto mfma4 = [&](const float4 (&qv)[G], const float4& xv) {
#pragma unroll
            for (int g = 0; g < G; g++) {
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[g].x, xv.x, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[g].y, xv.y, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[g].z, xv.z, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[g].w, xv.w, acc[g], 0, 0, 0);
            }
        };
        float4 qA[G], qB[G];   // operands of even / odd chunks: ping-pong, no moves
        load_q(qA, 0);
        auto consume = [&](const float4 (&x)[U], uint32_t c0) {
            static_assert(U % 2 == 0, "chunks are consumed in pairs");
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                if (FULL || c0 + 16 * u < dim16) {  // wave-uniform
                    load_q(qB, c0 + 16 * (u + 1));
                    mfma4(qA, x[u]);
                }
                if (FULL || c0 + 16 * (u + 1) < dim16) {
                    load_q(qA, c0 + 16 * (u + 2));
                    mfma4(qB, x[u + 1]);
                }
            }
        };
        for (uint32_t c0 = 0; c0 < dim16; c0 += 32 * U) {
            load_chunks(xb, v, live, c0 + 16 * U);
            consume(xa, c0);
            if (c0 + 32 * U < dim16) {
                load_chunks(xa, v, live, c0 + 32 * U);
            } else if (tile_next < tiles) {   // xa is free: the next tile's first chunks travel during the rest
                load_chunks(xa, row_ptr(tile_next), tile_next * 16 + nn < n, 0);
            }
            consume(xb, c0 + 16 * U);
        }
        // D: col = lane&15 = row in tile, row = 4*(lane>>4) + reg = query in group
        const uint32_t qe = MODE == kKeysList ? pr.list[it >> tsh].x : 0u;   // wave-uniform
        if (live || MODE == kKeysMins) {
#pragma unroll
            for (int g = 0; g < G; g++) {
                uint32_t mq[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t qt = g * 16 + 4 * q4 + r;
                    const float qn = qnr[g][r];
                    uint32_t key = 0xffffffffu;
                    if (live && vn != 0.f && qn != 0.f) {
                        const float sc = acc[g][r] / (qn * vn);
                        if (sc == sc) key = score_to_key(sc);
                    }
                    if (MODE == kKeysDense) {
                        if (qt < nq_pass) keys[(size_t)qt * n + row] = key;
                    } else if (MODE == kKeysMins) {
                        // minimum over the tile's 16 rows = the 16 lanes of this DPP row
                        uint32_t m = key;
                        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xf, 0xf, false));   // row_ror:8
                        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xf, 0xf, false));   // row_ror:4
                        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x122, 0xf, 0xf, false));   // row_ror:2
                        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x121, 0xf, 0xf, false));   // row_ror:1
                        mq[r] = m;
                        wave_min[g][r] = min(wave_min[g][r], m);
                    } else if (qt < nq_pass && qt + q_base == qe) {   // (qt < nq_pass: the image's padding rows belong to the next slice)
                        pr.ckeys[((it >> tsh) << pr.cs_shift) + ((it & (((size_t)1 << tsh) - 1)) << 4) + nn] = key;
                    }
                }
                // queries 16 g + 4 q4 .. + 3 of this tile: 16 bytes per lane row, 64 G contiguous bytes per tile
                if (MODE == kKeysMins && nn == 0)
                    *reinterpret_cast<uint4*>(pr.mins + tile * (16 * G) + g * 16 + 4 * q4) = make_uint4(mq[0], mq[1], mq[2], mq[3]);
            }
        }
    }
    if (MODE == kKeysMins && nn == 0) {
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                pr.wmin[(size_t)(g * 16 + 4 * q4 + r) * ((size_t)gridDim.x * kCW) + (size_t)blockIdx.x * kCW + wave] = wave_min[g][r];
    }
}

// ---- the minima of 5 .. 64 queries through the f16 matrix pipe (round 4) -----------------------------------------------
// The chunk minima only steer the pruning (which ~k chunks get their exact keys computed by the kKeysList pass above), so they
// need not be exact -- they need an error BOUND.  cosine_mins_f16 computes score~ = sum_i fl16(q_i / |q|) fl16(v_i / |v|) with
// v_mfma_f32_16x16x32_f16 (16 cycles for 8192 products; the f32 tile takes 32 cycles for 1024), f32 accumulation:
//   |score~ - score| <= 2^-10 (1 + 2^-11) sum |q^_i v^_i|   (two roundings to 11 bits, round to nearest even; <= 2^-10 by Cauchy-Schwarz)
//                     + 2^-25 (sum |q^_i| + sum |v^_i|)     (f16 subnormals: elements below 2^-14, absolute error 2^-25)
//                     + dim 2^-24                            (f32 accumulation, here and in the exact kernel)
// cosine_mins_eps(dim) rounds that up; the prune kernels widen their thresholds by it (topk.hip `eps`): every row that can be
// among the exact k best lies in a listed chunk, and the listed chunks' exact keys decide.  A row or query whose norm is
// outside [1e-30, 1e30] (the reciprocal would leave the normal f32 range) or whose approximate score is not finite raises
// the fallback flag: the dense pass answers.  The pass reads the rows exactly as the f32 tile does (16 rows x 64 B per wave
// load) and is bound by that stream at every batch size it takes: 64 queries cost 4 matrix instructions of 16 cycles and
// 8 conversions per 32-float step against 2 KiB of rows.
// Query image: f16, normalised, in the order the B-side loads deliver the rows' floats -- lane (row, q4) of step s holds floats
// 32 s + 4 q4 .. + 3 and 32 s + 16 + 4 q4 .. + 3 -- written once per search by cosine_norms_image, copied to LDS by every
// workgroup with a row stride of dim halves + 16 B (ds_read_b128 conflict-free).
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool norm_in_range(float v) { return v >= 1e-30f && v <= 1e30f; }

// one wave per query: |q| exactly as cosine_norms computes it, and the query's row of the f16 image (dim % 32 == 0)
__global__ __launch_bounds__(256) void cosine_norms_image(const float* __restrict__ queries, size_t nq, uint32_t dim,
                                                          float* __restrict__ norms, _Float16* __restrict__ image,
                                                          uint32_t* __restrict__ zero2) {
    if (zero2 && blockIdx.x == 0 && threadIdx.x < 2) zero2[threadIdx.x] = 0;   // the pass's flag and list counter (saves a memset launch)
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= nq) return;
    const float* v = queries + row * dim;
    float acc = 0.f;
    for (uint32_t i = lane; i < dim; i += 64) acc = fmaf(v[i], v[i], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    const float qn = sqrtf(acc);
    if (lane == 0) norms[row] = qn;
    const float s = norm_in_range(qn) ? 1.0f / qn : 0.f;   // (out of range, NaN: cosine_mins_f16 raises the fallback flag)
    for (uint32_t c = 4 * lane; c < dim; c += 256) {
        const f32x4v x = *reinterpret_cast<const f32x4v*>(v + c);
        f16x4v h;
#pragma unroll
        for (int e = 0; e < 4; e++) h[e] = (_Float16)(x[e] * s);
        const uint32_t st = c >> 5, r = c & 31u;
        *reinterpret_cast<f16x4v*>(image + row * dim + st * 32 + ((r & 15u) >> 2) * 8 + (r >> 4) * 4) = h;
    }
}

template <int G>
__global__ __launch_bounds__(kCW * 64) void cosine_mins_f16(const float* __restrict__ rows, const float* __restrict__ norms,
                                                            size_t n, uint32_t dim, const _Float16* __restrict__ image,
                                                            const float* __restrict__ qnorm, uint32_t nq_pass,
                                                            uint32_t* __restrict__ flag, CosinePrune pr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t qimg[];   // [16 G][dim halves + 16 B], 64 B of slack, [wave][16][272 B] stage
    const uint32_t stride = dim * 2 + 16;
    const uint32_t stage_off = 16u * G * stride + 64;
    {   // the image: 16-byte pieces, all of a thread's loads requested before its first store
        const uint32_t ppr = dim / 8, total = 16u * G * ppr;
        for (uint32_t i0 = threadIdx.x; i0 < total; i0 += 4 * kCW * 64) {
            uint4 t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * kCW * 64, qt = i / ppr;
                t[u] = make_uint4(0, 0, 0, 0);
                if (i < total && qt < nq_pass) t[u] = *reinterpret_cast<const uint4*>(image + (size_t)i * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * kCW * 64, qt = i / ppr;
                if (i < total) *reinterpret_cast<uint4*>(qimg + qt * stride + (i - qt * ppr) * 16) = t[u];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nn = lane & 15, q4 = lane >> 4;
    const size_t tiles = (n + 15) / 16;
    float qnr[G][4];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t qt = g * 16 + 4 * q4 + r;
            qnr[g][r] = qt < nq_pass ? qnorm[qt] : 0.f;
        }
    bool bad = false;
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) bad |= qnr[g][r] != 0.f && !norm_in_range(qnr[g][r]);   // (NaN too)
    // A wave's work is ONE stream of blocks (a block = 64 floats of the 16 rows of a tile = 4 KiB = 2 matrix steps), tile after
    // tile.  The rows are LOADED coalesced -- a wave load is 4 rows x 256 contiguous bytes, lane = (row in 4, 16-byte piece) --
    // and reach the matrix operand's layout (lane = (row in 16, k group)) through a 4.25 KiB stage per wave in LDS: written as
    // loaded (a quarter wave = one row's 256 B), read back as two 16-byte pieces per lane and step with a row stride of 272 B
    // (16 rows at one piece: 16 different bank groups).  Loaded straight into the operand layout (16 rows x 64 B per wave
    // load, every quarter wave touching 16 lines) the pass measured 0.55 ms per million 768-d rows whatever it computed; the
    // same bytes coalesced, 0.43.  LDS executes a wave's instructions in order, so the stage needs no barrier and no second
    // buffer; four register buffers keep three blocks (12 KiB per wave) in flight.  Every refill is unconditional (past the
    // end of the wave's work it reads the last tile again, into a buffer nobody uses): with loads under branches the compiler
    // has to wait for the counter's worst case at every join, which drains the queue.
    constexpr int R = 4, U = 4;
    f32x4v x[R][U];
    float nx[R];                               // the norm of this lane's operand row (row nn of the block's tile), loaded with the block
    const size_t tstep = (size_t)gridDim.x * kCW;
    const uint32_t nblk = dim / 64;
    const uint8_t* __restrict__ qlane = qimg + (size_t)nn * stride + q4 * 16;
    uint8_t* __restrict__ stage = qimg + stage_off + (size_t)wave * (16 * 272);
    uint8_t* __restrict__ st_wr = stage + (lane >> 4) * 272 + (lane & 15) * 16;      // + u * 4 * 272
    const uint8_t* __restrict__ st_rd = stage + nn * 272 + q4 * 16;                   // + (8 s + 4 h) * 16
    auto load_q = [&](f16x8v (&qv)[G], uint32_t step) {
#pragma unroll
        for (int g = 0; g < G; g++) qv[g] = *reinterpret_cast<const f16x8v*>(qlane + (size_t)g * 16 * stride + step * 64);
    };
    size_t ld_tile = (size_t)blockIdx.x * kCW + wave;      // the position the next refill reads
    uint32_t ld_blk = 0;
    auto refill = [&](f32x4v (&xr)[U], float& nr) {
        const size_t t = ld_tile < tiles ? ld_tile : tiles - 1;
        const size_t rn = t * 16 + nn < n ? t * 16 + nn : n - 1;
        nr = norms[rn];                            // FIRST: a tile's first multiply waits for it, and must not wait for the data behind it
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t r0 = t * 16 + 4 * u + (lane >> 4);
            const size_t r = r0 < n ? r0 : n - 1;     // dead rows read the last row; their results are not used
            xr[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(rows + r * (size_t)dim + 64 * ld_blk + 4 * (lane & 15)));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (++ld_blk == nblk) {
            ld_blk = 0;
            ld_tile += tstep;
        }
    };
    uint32_t wave_min[G][4];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) wave_min[g][r] = 0xffffffffu;
    size_t it = ld_tile;                                    // the position being consumed
    uint32_t blk = 0;
#pragma unroll
    for (int b = 0; b < R; b++) refill(x[b], nx[b]);
    __syncthreads();   // the query image is complete
    f32x4v acc[G];
    f16x8v qA[G], qB[G];
    load_q(qA, 0);
    float sv = 0.f;
    bool scored = false;
    auto block = [&](f32x4v (&xr)[U], float& nr) {
        if (blk == 0) {            // wave-uniform: a tile begins
            const bool live = it * 16 + nn < n;
            scored = live && nr != 0.f;
            if (scored && !norm_in_range(nr)) bad = true;
            sv = scored && norm_in_range(nr) ? 1.0f / nr : 0.f;
#pragma unroll
            for (int g = 0; g < G; g++) acc[g] = f32x4v{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; u++) *reinterpret_cast<f32x4v*>(st_wr + u * (4 * 272)) = xr[u];
        refill(xr, nr);
        // (aligned register pairs spelled out: left to itself the vectoriser pairs elements 1-2 and 3-0 of neighbouring pieces and
        // repacks every register with a move)
        const f32x2v sv2 = f32x2v{sv, sv};
        auto operand = [&](const f32x4v& lo, const f32x4v& hi) {
            const f16x2v h0 = __builtin_convertvector(__builtin_shufflevector(lo, lo, 0, 1) * sv2, f16x2v);
            const f16x2v h1 = __builtin_convertvector(__builtin_shufflevector(lo, lo, 2, 3) * sv2, f16x2v);
            const f16x2v h2 = __builtin_convertvector(__builtin_shufflevector(hi, hi, 0, 1) * sv2, f16x2v);
            const f16x2v h3 = __builtin_convertvector(__builtin_shufflevector(hi, hi, 2, 3) * sv2, f16x2v);
            const f16x4v l = __builtin_shufflevector(h0, h1, 0, 1, 2, 3), h = __builtin_shufflevector(h2, h3, 0, 1, 2, 3);
            return __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        const f32x4v l0 = *reinterpret_cast<const f32x4v*>(st_rd), h0 = *reinterpret_cast<const f32x4v*>(st_rd + 64);
        const f32x4v l1 = *reinterpret_cast<const f32x4v*>(st_rd + 128), h1 = *reinterpret_cast<const f32x4v*>(st_rd + 192);
        const uint32_t step0 = blk * 2;
        const uint32_t next0 = blk + 1 == nblk ? 0u : step0 + 2;      // the first step of the stream's next block
        load_q(qB, step0 + 1);
        const f16x8v b0 = operand(l0, h0);
#pragma unroll
        for (int g = 0; g < G; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qA[g], b0, acc[g], 0, 0, 0);
        load_q(qA, next0);
        const f16x8v b1 = operand(l1, h1);
#pragma unroll
        for (int g = 0; g < G; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qB[g], b1, acc[g], 0, 0, 0);
        if (++blk == nblk) {       // wave-uniform: the tile is complete
            blk = 0;
            // D: col = lane & 15 = row in tile, row = 4 (lane >> 4) + reg = query in group
            if (it < tiles) {
#pragma unroll
                for (int g = 0; g < G; g++) {
                    uint32_t mq[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        uint32_t key = 0xffffffffu;
                        if (scored && qnr[g][r] != 0.f) {
                            const float sc = acc[g][r];
                            if (!(fabsf(sc) <= 2.0f)) bad = true;
                            key = score_to_key(sc);
                        }
                        uint32_t m = key;
                        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xf, 0xf, false));   // row_ror:8
                        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xf, 0xf, false));   // row_ror:4
                        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x122, 0xf, 0xf, false));   // row_ror:2
                        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x121, 0xf, 0xf, false));   // row_ror:1
                        mq[r] = m;
                        wave_min[g][r] = min(wave_min[g][r], m);
                    }
                    if (nn == 0)
                        *reinterpret_cast<uint4*>(pr.mins + it * (16 * G) + g * 16 + 4 * q4) = make_uint4(mq[0], mq[1], mq[2], mq[3]);
                }
            }
            it += tstep;
        }
    };
    while (it < tiles) {       // (past the wave's last tile: blocks of the last tile again, multiplied and dropped)
#pragma unroll
        for (int b = 0; b < R; b++) block(x[b], nx[b]);
    }
    if (nn == 0) {
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                pr.wmin[(size_t)(g * 16 + 4 * q4 + r) * ((size_t)gridDim.x * kCW) + (size_t)blockIdx.x * kCW + wave] = wave_min[g][r];
    }
    if (__any(bad) && lane == 0) *flag = 1;
}

// ---- 5 .. 48 queries: the row stream through v_mfma_f32_4x4x1_16B_f32 ------------------------------------------
// The 16x16x4 tile above needs 16 rows per instruction, so a wave load touches 16 rows x 64 B (four lanes per row):
// 5.4 TB/s at best.  The 4x4x1 form is 16 independent 4x4 outer products ("blocks"; layout measured with
// tools/probe_mfma4x4.hip: lane = 4 block + i, A_b[i] / B_b[j] one value per lane, D_b[i][j] in register i of lane
// 4 block + j).  Block b takes the k-slice 4 b .. 4 b + 3 of every 64-float chunk: lane (b, j) loads float4
// row j, dims 64 c + 4 b .., so ONE wave load is 4 rows x 256 contiguous bytes, and its four components feed four
// successive MFMAs against the queries' same dims (A: lane (b, i) = query 4 g + i, one ds_read_b128 per chunk and
// query group from an LDS image whose row stride is 16 mod 64 floats: the read is conflict-free).  Two row tiles
// (8 rows) share every A read.  The 16 blocks' partial dot products are summed across lanes at the end (two DPP
// rotations inside a row of 16 lanes, two exchanges across rows).  HBM-bound up to 48 queries: 8 rows x 12 chunks
// cost 96 G MFMAs (G = query groups of 4) of 8 cycles against 24 KiB of rows.
struct CosineFilter {
    const uint32_t* tau;
    const float* uq;
    uint32_t* ccnt;
    uint32_t* ckey;
    uint32_t* crow;
    uint32_t cap;
    uint32_t row_base;
};
constexpr int kBW = 8;     // waves per workgroup (one workgroup per CU: the query image fills its LDS)
// A wave owns SUPERTILES of 32 consecutive rows (four 8-row tiles): the keys of a supertile are collected in LDS and
// leave as 128-byte runs per query (stored straight from the accumulator lanes they were 16-byte pieces: a fifth of
// the kernel's time at 16 queries).
template <int NCH>         // 64-float chunks per row: dim <= 64 NCH
__global__ __launch_bounds__(kBW * 64) void cosine_keys_blocks(const float* __restrict__ rows,
                                                          const float* __restrict__ norms, size_t n, uint32_t dim,
                                                          const float* __restrict__ queries,
                                                          const float* __restrict__ qnorm, uint32_t nq_pass,
                                                          uint32_t qstride, uint32_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ run_flag) {
    if (run_flag && *run_flag == 0) return;
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [16][qstride] query image, zero-filled past dim / nq_pass;
    uint32_t* stage = reinterpret_cast<uint32_t*>(qs + 16 * qstride) + (threadIdx.x >> 6) * (16 * 32);   // then [wave][16][32] keys
    fill_query_image(qs, queries, 16u, nq_pass, dim, qstride, kBW * 64);   // (dim % 4 == 0, queries 16-byte aligned: blocks_path)
    // (the barrier behind the fill comes after the first row loads have been requested, below)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = lane >> 2, j = lane & 3;
    const size_t tiles = (n + 7) / 8, supers = (n + 31) / 32;
    const size_t sstep = (size_t)gridDim.x * kBW;
    const float* __restrict__ qrow = qs + (size_t)j * qstride + 4 * blk;     // this lane's A source: query 4 g + j
    // A tile's rows are loaded as two halves (H chunks each).  While one half is multiplied the other -- or the same
    // half of the NEXT tile -- is in flight: a wave always has ~12 KiB of row data outstanding.
    constexpr int H = NCH / 2;
    static_assert(NCH % 2 == 0, "two halves");
    f32x4v xa0[H], xa1[H], xb0[H], xb1[H];
    auto load_half = [&](f32x4v (&h0)[H], f32x4v (&h1)[H], size_t tile, int c0) {
        const size_t r0 = tile * 8 + j, r1 = r0 + 4;
        const float* __restrict__ v0 = rows + (r0 < n ? r0 : n - 1) * (size_t)dim + 4 * blk;   // clamped duplicates are not stored
        const float* __restrict__ v1 = rows + (r1 < n ? r1 : n - 1) * (size_t)dim + 4 * blk;
#pragma unroll
        for (int c = 0; c < H; c++) {
            const bool in = 64u * (c0 + c) + 4u * blk < dim;
            h0[c] = in ? __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(v0 + 64 * (c0 + c))) : f32x4v{0.f, 0.f, 0.f, 0.f};
            h1[c] = in ? __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(v1 + 64 * (c0 + c))) : f32x4v{0.f, 0.f, 0.f, 0.f};
        }
    };
    size_t sup = (size_t)blockIdx.x * kBW + wave;
    if (sup < supers) {
        load_half(xa0, xa1, sup * 4, 0);
        load_half(xb0, xb1, sup * 4, H);
    }
    __syncthreads();   // the query image is complete
    for (; sup < supers; sup += sstep) {
#pragma unroll 1
        for (int t = 0; t < 4; t++) {
            const size_t tile = sup * 4 + t;
            // the tile after this one: the supertile's next, or the first of the wave's next supertile (past the end
            // the loads fall on clamped rows and are dropped)
            const size_t next = t < 3 ? tile + 1 : (sup + sstep) * 4;
            const bool more = next < tiles;
            const size_t r0 = tile * 8 + j, r1 = r0 + 4;
            const float vn0 = norms[r0 < n ? r0 : n - 1], vn1 = norms[r1 < n ? r1 : n - 1];
            f32x4v acc0[4], acc1[4];
#pragma unroll
            for (int g = 0; g < 4; g++) acc0[g] = acc1[g] = f32x4v{0.f, 0.f, 0.f, 0.f};
            auto mul_half = [&](const f32x4v (&h0)[H], const f32x4v (&h1)[H], int c0) {
#pragma unroll
                for (int c = 0; c < H; c++) {
                    f32x4v a[4];
#pragma unroll
                    for (int g = 0; g < 4; g++)
                        a[g] = *reinterpret_cast<const f32x4v*>(qrow + (size_t)g * 4 * qstride + 64 * (c0 + c));
                    // the eight accumulators are independent: each MFMA's successor on the same accumulator is eight
                    // instructions away (issued back to back a dependent pair waits for the first one's passes)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            acc0[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g][e], h0[c][e], acc0[g], 0, 0, 0);
                            acc1[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g][e], h1[c][e], acc1[g], 0, 0, 0);
                        }
                    }
                }
            };
            mul_half(xa0, xa1, 0);
            if (more) load_half(xa0, xa1, next, 0);
            mul_half(xb0, xb1, H);
            if (more) load_half(xb0, xb1, next, H);
            // sum over the 16 blocks: lanes with the same j.  Inside a row of 16 lanes two rotations; across the four
            // rows two exchanges.  Afterwards every lane holds the full dot products of its j.
#pragma unroll
            for (int g = 0; g < 4; g++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float a0 = acc0[g][i], a1 = acc1[g][i];
                    // row_ror:8 FIRST, then row_ror:4: the pairs {b, b + 2} and then {pair, other pair} are the same two
                    // operands in every lane, so all lanes end with the same bits (the other order associates
                    // differently per lane, and copies of a row in the two halves of a tile then differ by an ulp)
                    a0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x128, 0xf, 0xf, false));   // row_ror:8
                    a1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0x128, 0xf, 0xf, false));
                    a0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x124, 0xf, 0xf, false));   // row_ror:4
                    a1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0x124, 0xf, 0xf, false));
                    a0 += __shfl_xor(a0, 16, 64);
                    a1 += __shfl_xor(a1, 16, 64);
                    a0 += __shfl_xor(a0, 32, 64);
                    a1 += __shfl_xor(a1, 32, 64);
                    // lanes of block b = 2 g + {0, 1} key rows j / 4 + j of the tile for query 4 g + i
                    const uint32_t qt = g * 4 + i;
                    if ((blk >> 1) == g) {
                        const bool second = blk & 1;
                        const float dot = second ? a1 : a0, vn = second ? vn1 : vn0, qn = qt < nq_pass ? qnorm[qt] : 0.f;
                        uint32_t key = 0xffffffffu;
                        if (vn != 0.f && qn != 0.f) {
                            const float sc = dot / (qn * vn);
                            if (sc == sc) key = score_to_key(sc);
                        }
                        stage[qt * 32 + 8 * t + (second ? 4 : 0) + j] = key;
                    }
                }
            }
        }
        // the supertile's keys: 16 queries x 32 rows, two queries (2 x 128 B) per store
        wave_lds_fence();
        const size_t row = sup * 32 + (lane & 31);
#pragma unroll
        for (int qq = 0; qq < 16; qq += 2) {
            const uint32_t qt = qq + (lane >> 5);
            const uint32_t key = stage[qt * 32 + (lane & 31)];
            if (qt < nq_pass && row < n) keys[(size_t)qt * n + row] = key;
        }
        wave_lds_fence();
    }
}

// ---- streaming variant for a handful of queries (the reference's own shape: one query per request) ----
// The MFMA tile above feeds a wave-load with 16 rows x 64 B; for NQ <= 4 queries the matrix pipes do nothing
// useful and the row stream is all that matters.  Here a wave reads a ROW as contiguous 1 KiB wave-loads (lane l
// holds dims 256 b + 4 l .. + 4 of block b; the same dims of each query sit in its registers), multiplies, and
// butterflies the 64 partial sums; four rows per step keep 4 NB KiB in flight per wave.  HBM-bound with
// ideal coalescing.  dim <= 1024, dim % 4 == 0 (blocks past dim are zero-filled).
template <int NQ, int NB>
__global__ __launch_bounds__(256) void cosine_keys_stream(const float* __restrict__ rows, const float* __restrict__ norms,
                                                          size_t n, uint32_t dim, const float* __restrict__ queries,
                                                          const float* __restrict__ qnorm, uint32_t nq_pass,
                                                          uint32_t* __restrict__ keys) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (size_t)gridDim.x * 4;
    f32x4v q[NQ][NB];
    float qn[NQ];
#pragma unroll
    for (int t = 0; t < NQ; t++) {
        const bool qlive = (uint32_t)t < nq_pass;
        qn[t] = qlive ? qnorm[t] : 0.f;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const uint32_t c = 256u * b + 4u * lane;
            q[t][b] = (qlive && c < dim) ? *reinterpret_cast<const f32x4v*>(queries + (size_t)t * dim + c)
                                         : f32x4v{0.f, 0.f, 0.f, 0.f};
        }
    }
    constexpr int R = 4;   // rows per step
    for (size_t r0 = wave * R; r0 < n; r0 += waves * R) {
        f32x4v x[R][NB];
#pragma unroll
        for (int i = 0; i < R; i++) {
            const size_t row = r0 + i < n ? r0 + i : n - 1;      // a clamped duplicate, not stored
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const uint32_t c = 256u * b + 4u * lane;
                x[i][b] = c < dim ? __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(rows + row * (size_t)dim + c))
                                  : f32x4v{0.f, 0.f, 0.f, 0.f};
            }
        }
        float vn = 0.f;
        if (lane < R) vn = norms[r0 + lane < n ? r0 + lane : n - 1];
#pragma unroll
        for (int t = 0; t < NQ; t++) {
            float mine = 0.f;   // lane i ends up with row i's dot product
#pragma unroll
            for (int i = 0; i < R; i++) {
                float a = 0.f;
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    a = fmaf(x[i][b][0], q[t][b][0], a);
                    a = fmaf(x[i][b][1], q[t][b][1], a);
                    a = fmaf(x[i][b][2], q[t][b][2], a);
                    a = fmaf(x[i][b][3], q[t][b][3], a);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
                if (lane == i) mine = a;
            }
            if (lane < R && r0 + lane < n && (uint32_t)t < nq_pass) {
                uint32_t key = 0xffffffffu;
                if (vn != 0.f && qn[t] != 0.f) {
                    const float sc = mine / (qn[t] * vn);
                    if (sc == sc) key = score_to_key(sc);
                }
                keys[(size_t)t * n + r0 + lane] = key;
            }
        }
    }
}

// ---- GEMM variant for large batches: every query of the pass against the corpus in ONE read of the rows ----
// The LDS-resident form above holds whole query rows, so 48 queries are all that fit and a batch of 256 reads
// the corpus six times, each pass sitting on the MFMA/HBM ridge.  Here a wave owns 16 rows x ALL 16*NG queries
// (4 NG accumulator registers) and the workgroup walks K in slices of 32 dims: the slice of every query is
// staged through LDS (double-buffered, laid out so that the A operand of (group, 16-dim chunk) is one
// conflict-free ds_read_b128 per lane), the rows stream from HBM once.  f32 MFMA-bound: NG = 16 is 2.5 ms of
// matrix work per million 768-d rows.
constexpr int kGW = 8;      // waves per workgroup
constexpr int kGK = 32;     // dims per K slice

// hand-issued LDS operand read and its counted wait (see cosine_keys_gemm)
__device__ __forceinline__ void lds_read128(f32x4v& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(f32x4v& reg) {
    if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(reg));
    else if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(reg));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(reg));
}

// FILT: instead of the nq x n key matrix, a row whose key is <= tau[q] joins q's candidate list (row number + key).
// tau is the k-th best key of a sample of the corpus, so about k n / sample rows per query pass; the test costs four
// VALU operations per result (acc against uq[q] * |row|, uq = score(tau) * |q|, with a 2^-21 relative slack for the
// roundings of the exact path) and only the passers pay for the division and the exact key.
template <int NG, bool FILT, int RT>
__global__ __launch_bounds__(kGW * 64) __attribute__((amdgpu_waves_per_eu(4 / RT, 4 / RT))) void cosine_keys_gemm(const float* __restrict__ rows,
                                                             const float* __restrict__ norms, size_t n, uint32_t dim,
                                                             const float* __restrict__ queries,
                                                             const float* __restrict__ qnorm, uint32_t nq_pass,
                                                             uint32_t* __restrict__ keys, CosineFilter flt,
                                                             const uint32_t* __restrict__ run_flag) {
    if (run_flag && *run_flag == 0) return;
    // [buffer][group][chunk][lane] float4: 2 x NG x 2 KiB
    __shared__ __attribute__((aligned(16))) float4 qsl[2][NG][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nn = lane & 15, q4 = lane >> 4;
    const uint32_t nks = dim / kGK;
    // staging: the slice of (group g, chunk ch) is one global_load_lds_dwordx4 -- 64 lanes x 16 B land lane-linear
    // in qsl[buf][g][ch][], straight from L2, with no register in between -- issued at the TOP of the slice before
    // the one that reads it, so its latency sits under 128 MFMAs.  (Staged through registers at the END of a
    // slice, every wave of both resident workgroups met the L2 round trip and the barrier at the same moment:
    // the matrix pipes idled a third of the time, SQ_VALU_MFMA_BUSY_CYCLES.)  Queries past the batch re-read the
    // last one; their columns are never stored.
    constexpr int kDma = NG * 2 / kGW;
    static_assert(kDma >= 1, "fewer (group, chunk) pairs than waves");
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // this lane's operand (query nn, dims 4 q4 .. + 4) sits in slot 4 nn + q4 (a 2-way bank conflict on the read)
    const uint32_t lds_lane = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&qsl[0][0][0][4 * nn + q4];
    auto stage_dma = [&](uint32_t ks, int buf) {
#pragma unroll
        for (int i = 0; i < kDma; i++) {
            const int idx = wv * kDma + i, g = idx >> 1, ch = idx & 1;
            // slot l of the 1 KiB chunk holds (query l / 4, dims 4 (l % 4) .. + 4): four consecutive lanes fetch 64
            // contiguous bytes (with one lane per query the instruction touched 64 lines for 16 B each)
            uint32_t q = (uint32_t)(g * 16 + (lane >> 2));
            q = q < nq_pass ? q : nq_pass - 1;
            const float* src = queries + (size_t)q * dim + ks * kGK + 16 * ch + 4 * (lane & 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)&qsl[buf][g][ch][0], 16, 0, 0);
        }
    };
    const size_t tiles = (n + 15) / 16;
    // a wave owns RT row tiles (16 rows each) of the workgroup's kGW * RT: with RT = 2 every operand read feeds eight
    // MFMAs and the query slices are fetched once per 256 rows (the L2 -> LDS stream is what bounds RT = 1)
    const size_t tstep = (size_t)gridDim.x * kGW * RT;
    size_t tb = (size_t)blockIdx.x * kGW * RT;
    if (tb >= tiles) return;
    auto row_of = [&](size_t t, int r) { return (t + (size_t)wave * RT + r) * 16 + nn; };
    auto row_ptr = [&](size_t t, int r) {
        const size_t row = row_of(t, r);
        return rows + (row < n ? row : 0) * (size_t)dim + 4 * q4;   // dead rows read row 0; their results are not stored
    };
    // row chunks of slices ks and ks + 1 in flight
    float4 xc[RT][2], xn[RT][2];
#pragma unroll
    for (int r = 0; r < RT; r++) xn[r][0] = xn[r][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_x = [&](float4 (&x)[2], const float* __restrict__ v, uint32_t ks) {
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            const f32x4v t = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(v + ks * kGK + 16 * ch));
            x[ch] = make_float4(t[0], t[1], t[2], t[3]);
        }
    };
    // The slices of all tiles form ONE pipeline: the buffer parity runs on across tiles and the last slice of a tile
    // already requests the first slice of the next one (its query slice and this wave's next rows), so the epilogue
    // and the tile change cost no HBM round trip and no extra barrier.
    uint32_t gs = 0;
    const float* __restrict__ v[RT];
#pragma unroll
    for (int r = 0; r < RT; r++) v[r] = row_ptr(tb, r);
    stage_dma(0, 0);
#pragma unroll
    for (int r = 0; r < RT; r++) load_x(xc[r], v[r], 0);
    __syncthreads();
    for (; tb < tiles; tb += tstep) {
        const bool more = tb + tstep < tiles;
        const float* __restrict__ vnext[RT];
        float vn[RT];
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const size_t row = row_of(tb, r);
            vn[r] = norms[row < n ? row : 0];
            vnext[r] = row_ptr(more ? tb + tstep : tb, r);
        }
        f32x4v acc[RT][NG];
#pragma unroll
        for (int r = 0; r < RT; r++)
#pragma unroll
            for (int g = 0; g < NG; g++) acc[r][g] = f32x4v{0.f, 0.f, 0.f, 0.f};
        for (uint32_t ks = 0; ks < nks; ks++, gs++) {
            const int buf = gs & 1;
            if (ks + 1 < nks) {
                stage_dma(ks + 1, buf ^ 1);   // last read during the previous slice; every wave passed the barrier since
#pragma unroll
                for (int r = 0; r < RT; r++) load_x(xn[r], v[r], ks + 1);
            } else if (more) {
                stage_dma(0, buf ^ 1);
#pragma unroll
                for (int r = 0; r < RT; r++) load_x(xn[r], vnext[r], 0);
            }
            // The 2 NG operand reads of the slice are hand-issued two ahead of their MFMAs (a ring of three
            // registers) with counted lgkmcnt waits.  Left to the compiler (128-VGPR budget) every read was followed
            // by lgkmcnt(0) and its four MFMAs -- an LDS round trip per 128 matrix cycles -- and, being visible LDS
            // reads behind an LDS-DMA, each slice opened with vmcnt(0) on the prefetch just issued: the matrix pipes
            // ran 2/3 of the time (SQ_VALU_MFMA_BUSY_CYCLES).  No scalar loads happen inside (lgkmcnt is LDS-only).
            {
                const uint32_t a0 = lds_lane + (uint32_t)buf * (uint32_t)(NG * 2 * 64 * 16);
                f32x4v q3[3];
                constexpr int kSteps = 2 * NG;               // step s = (chunk s / NG, group s % NG)
                auto off = [](int st) { return (uint32_t)((st % NG) * 2048 + (st / NG) * 1024); };
                lds_read128(q3[0], a0 + off(0));
                lds_read128(q3[1], a0 + off(1));
#pragma unroll
                for (int st = 0; st < kSteps; st++) {
                    if (st + 2 < kSteps) lds_read128(q3[(st + 2) % 3], a0 + off(st + 2));
                    if (st + 2 < kSteps) lds_wait<2>(q3[st % 3]);
                    else if (st + 1 < kSteps) lds_wait<1>(q3[st % 3]);
                    else lds_wait<0>(q3[st % 3]);
                    const int g = st % NG;
                    const f32x4v qa = q3[st % 3];
#pragma unroll
                    for (int r = 0; r < RT; r++) {
                        const float4 xv = xc[r][st / NG];
                        acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[0], xv.x, acc[r][g], 0, 0, 0);
                        acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[1], xv.y, acc[r][g], 0, 0, 0);
                        acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[2], xv.z, acc[r][g], 0, 0, 0);
                        acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[3], xv.w, acc[r][g], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RT; r++) {
                xc[r][0] = xn[r][0];
                xc[r][1] = xn[r][1];
            }
            __syncthreads();
        }
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            v[rt] = vnext[rt];
            const size_t row = row_of(tb, rt);
            if (row >= n) continue;
#pragma unroll
            for (int g = 0; g < NG; g++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    uint32_t qt = g * 16 + 4 * q4 + r;
                    asm volatile("" : "+v"(qt));   // keep the 64 addresses of a tile's results out of the slice loop's registers
                    if (qt < nq_pass) {
                        const float a = acc[rt][g][r];
                        if constexpr (FILT) {
                            const float t = flt.uq[qt] * vn[rt];
                            if (a >= t - fabsf(t) * 0x1p-21f - 0x1p-120f) {
                                const float qn = qnorm[qt];
                                if (vn[rt] != 0.f && qn != 0.f) {
                                    const float sc = a / (qn * vn[rt]);
                                    const uint32_t key = sc == sc ? score_to_key(sc) : 0xffffffffu;
                                    if (key != 0xffffffffu && key <= flt.tau[qt]) {
                                        const uint32_t pos = atomicAdd(&flt.ccnt[qt], 1u);
                                        if (pos < flt.cap) {
                                            flt.ckey[(size_t)qt * flt.cap + pos] = key;
                                            flt.crow[(size_t)qt * flt.cap + pos] = flt.row_base + (uint32_t)row;
                                        }
                                    }
                                }
                            }
                        } else {
                            const float qn = qnorm[qt];   // once per 16 rows x 768 dims of matrix work: not worth 4 NG registers
                            uint32_t key = 0xffffffffu;
                            if (vn[rt] != 0.f && qn != 0.f) {
                                const float sc = a / (qn * vn[rt]);
                                if (sc == sc) key = score_to_key(sc);
                            }
                            keys[(size_t)qt * n + row] = key;
                        }
                    }
                }
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

namespace {
// groups of 16 queries whose rows (stride dim16 + 4 floats) fit 156 KiB of LDS, at most 3
int mfma_groups(uint32_t dim) {
    const uint32_t dim16 = (dim + 15) & ~15u;
    const size_t per_group = 16 * (size_t)(dim16 + 4) * sizeof(float);
    const size_t g = (156u * 1024u) / per_group;
    return (int)(g > 3 ? 3 : g);
}
bool mfma_ok(const float* rows, uint32_t dim) {
    return dim % 4 == 0 && (reinterpret_cast<uintptr_t>(rows) & 15u) == 0 && mfma_groups(dim) >= 1;
}
}  // namespace

// queries per corpus pass. MFMA path (dim % 4 == 0): 16 per LDS-resident group, up to 48;
// VALU path: kQT, or fewer when kQT query rows do not fit 144 KiB of LDS
int cosine_queries_per_pass(uint32_t dim, size_t nq) {
    // batches beyond what the LDS-resident kernel holds go through the GEMM kernel, 256 queries per corpus read
    if (dim % kGK == 0 && dim % 4 == 0 && mfma_groups(dim) >= 1 && nq > (size_t)16 * mfma_groups(dim)) return 256;
    if (dim % 4 == 0 && mfma_groups(dim) >= 1) return 16 * mfma_groups(dim);
    const uint32_t dim4 = (dim + 3) & ~3u;
    const size_t fit = (144u * 1024u) / ((size_t)dim4 * sizeof(float));
    return (int)(fit < (size_t)kQT ? fit : (size_t)kQT);
}

namespace {
bool gemm_path(const float* rows, uint32_t dim, const float* queries, uint32_t nq_pass) {
    return mfma_ok(rows, dim) && dim % kGK == 0 && nq_pass > (uint32_t)16 * mfma_groups(dim) &&
           (reinterpret_cast<uintptr_t>(queries) & 15u) == 0;
}
template <bool FILT>
void launch_gemm(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries, const float* qnorm,
                 uint32_t nq_pass, uint32_t* keys, const CosineFilter& flt, const uint32_t* run_flag, hipStream_t stream) {
    const size_t tiles = (n + 15) / 16;
    auto grid_for = [&](int rt) {
        unsigned grid = (unsigned)((tiles + kGW * rt - 1) / (kGW * rt));
        const unsigned cap = 256u * 2u / rt;   // resident workgroups: 128 VGPRs per row tile held
        return grid > cap ? cap : grid;
    };
    if (nq_pass <= 64)
        hipLaunchKernelGGL((cosine_keys_gemm<4, FILT, 1>), dim3(grid_for(1)), dim3(kGW * 64), 0, stream, rows, norms, n, dim,
                           queries, qnorm, nq_pass, keys, flt, run_flag);
    else if (nq_pass <= 128)
        hipLaunchKernelGGL((cosine_keys_gemm<8, FILT, 1>), dim3(grid_for(1)), dim3(kGW * 64), 0, stream, rows, norms, n, dim,
                           queries, qnorm, nq_pass, keys, flt, run_flag);
    else
        // RT = 2 (32 rows per wave, half the query-slice traffic) measured slower: at 256 VGPRs the compiler spills
        // (3.7 vs 3.45 ms per 256 queries over 1 M x 768, also with the epilogue addresses kept out of the loop)
        hipLaunchKernelGGL((cosine_keys_gemm<16, FILT, 1>), dim3(grid_for(1)), dim3(kGW * 64), 0, stream, rows, norms, n, dim,
                           queries, qnorm, nq_pass, keys, flt, run_flag);
}

// tau[q] = the k-th key of the sample's answer (0xffffffff while the sample holds fewer than k scored rows: then
// everything passes, the lists overflow and the dense path answers); uq = score(tau) * |q|; candidate counters zeroed
__global__ void cosine_tau_kernel(const uint32_t* __restrict__ base_key, uint32_t k, const float* __restrict__ qnorm,
                                  uint32_t nq, uint32_t* __restrict__ tau, float* __restrict__ uq,
                                  uint32_t* __restrict__ ccnt) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t t = base_key[(size_t)q * k + k - 1];
    tau[q] = t;
    uq[q] = (t == 0xffffffffu ? -2.0f : key_to_score(t)) * qnorm[q];
    ccnt[q] = 0;
}
}  // namespace

namespace {
// 5 .. 16 queries over rows of at most 1024 floats: the 4x4x1 row-stream kernel (beyond 16 queries its four times
// smaller matrix instruction costs more issue slots than the 16x16x4 tile's narrower loads cost bandwidth)
bool blocks_path(const float* rows, uint32_t dim, const float* queries, uint32_t nq_pass, size_t n) {
    return mfma_ok(rows, dim) && dim >= 512 && dim <= 1024 && nq_pass > 4 && nq_pass <= 16 && n >= 4096 &&
           (reinterpret_cast<uintptr_t>(queries) & 15u) == 0;
}
void launch_blocks(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries, const float* qnorm,
                   uint32_t nq_pass, uint32_t* keys, const uint32_t* run_flag, hipStream_t stream) {
    const uint32_t nch = ((dim + 63) / 64 + 1) & ~1u;      // whole chunks, an even number of them
    // the kernel instance walks tn >= nch chunks (rows beyond dim load as zeros): the query image must be that wide too, or the
    // last queries' operands of the surplus chunks come from whatever lies behind the image -- a NaN pattern there (the key
    // stage's 0xffffffff) times a zero row is a NaN score, and the row is dropped (dims 577-640 and 833-896; found by the soak)
    const uint32_t tn = nch <= 2 ? 2 : nch <= 4 ? 4 : nch <= 6 ? 6 : nch <= 8 ? 8 : nch <= 12 ? 12 : 16;
    const uint32_t qstride = tn * 64 + 16;                 // >= dim, and 16 mod 64 floats: conflict-free A reads
    const size_t lds = (size_t)16 * qstride * sizeof(float) + (size_t)kBW * 16 * 32 * sizeof(uint32_t);
    const size_t supers = (n + 31) / 32;
    unsigned grid = (unsigned)((supers + kBW - 1) / kBW);
    if (grid > 256) grid = 256;
    auto go = [&](auto kern) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBW * 64), lds, stream, rows, norms, n, dim, queries, qnorm, nq_pass, qstride,
                           keys, run_flag);
    };
    if (nch <= 2) go(cosine_keys_blocks<2>);
    else if (nch <= 4) go(cosine_keys_blocks<4>);
    else if (nch <= 6) go(cosine_keys_blocks<6>);
    else if (nch <= 8) go(cosine_keys_blocks<8>);
    else if (nch <= 12) go(cosine_keys_blocks<12>);
    else go(cosine_keys_blocks<16>);
}
template <int MODE>
void launch_mfma(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries, const float* qnorm,
                 uint32_t nq_pass, uint32_t* keys, const uint32_t* run_flag, hipStream_t stream, const CosinePrune& pr) {
    const int G = (int)((nq_pass + 15) / 16);  // <= mfma_groups(dim) by construction of the pass size
    const uint32_t dim16 = (dim + 15) & ~15u;
    const size_t lds = (size_t)16 * G * (dim16 + 4) * sizeof(float) + 64;   // + slack for the operand prefetch past the last row
    const size_t tiles = (n + 15) / 16;
    unsigned grid = (unsigned)((tiles + kCW - 1) / kCW);
    if (grid > 256 * 4) grid = 256 * 4;
    auto go = [&](auto kern) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kCW * 64), lds, stream, rows, norms, n, dim, queries, qnorm, nq_pass,
                           keys, run_flag, pr);
    };
    const bool full = dim % 256 == 0;   // 32 * U
    if (full) {
        if (G == 1) go(cosine_keys_mfma<1, true, MODE>);
        else if (G == 2) go(cosine_keys_mfma<2, true, MODE>);
        else go(cosine_keys_mfma<3, true, MODE>);
    } else {
        if (G == 1) go(cosine_keys_mfma<1, false, MODE>);
        else if (G == 2) go(cosine_keys_mfma<2, false, MODE>);
        else go(cosine_keys_mfma<3, false, MODE>);
    }
}
}  // namespace

bool cosine_filter_ok(const float* rows, uint32_t dim, const float* queries, uint32_t nq_pass, size_t n) {
    return gemm_path(rows, dim, queries, nq_pass) && n >= ((size_t)1 << 18) && n < ((size_t)1 << 32);
}

int launch_cosine_tau(const uint32_t* base_key, uint32_t k, const float* qnorm, uint32_t nq, uint32_t* tau, float* uq,
                      uint32_t* ccnt, hipStream_t stream) {
    if (nq == 0) return 0;
    hipLaunchKernelGGL(cosine_tau_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, base_key, k, qnorm, nq, tau, uq, ccnt);
    return 0;
}

int launch_cosine_keys_filtered(const float* rows, const float* norms, size_t n, size_t row_base, uint32_t dim,
                                const float* queries, const float* qnorm, uint32_t nq_pass, const uint32_t* tau,
                                const float* uq, uint32_t* ccnt, uint32_t* ckey, uint32_t* crow, uint32_t cap,
                                hipStream_t stream) {
    if (n == 0 || nq_pass == 0) return 0;
    const CosineFilter flt{tau, uq, ccnt, ckey, crow, cap, (uint32_t)row_base};
    launch_gemm<true>(rows, norms, n, dim, queries, qnorm, nq_pass, nullptr, flt, nullptr, stream);
    return 0;
}

int launch_cosine_keys(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries,
                       const float* qnorm, uint32_t nq_pass, uint32_t* keys, hipStream_t stream,
                       const uint32_t* run_flag) {
    if (n == 0 || nq_pass == 0) return 0;
    if (gemm_path(rows, dim, queries, nq_pass)) {
        launch_gemm<false>(rows, norms, n, dim, queries, qnorm, nq_pass, keys, CosineFilter{}, run_flag, stream);
        return 0;
    }
    if (nq_pass <= 4 && dim % 4 == 0 && dim <= 1024 && (reinterpret_cast<uintptr_t>(rows) & 15u) == 0 &&
        (reinterpret_cast<uintptr_t>(queries) & 15u) == 0 && n >= 4096) {
        unsigned grid = (unsigned)((n + 15) / 16);
        if (grid > 256 * 8) grid = 256 * 8;
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, stream, rows, norms, n, dim, queries, qnorm, nq_pass, keys);
        };
        const int nb = (int)((dim + 255) / 256);
        if (nq_pass == 1) {
            if (nb == 1) go(cosine_keys_stream<1, 1>);
            else if (nb == 2) go(cosine_keys_stream<1, 2>);
            else if (nb == 3) go(cosine_keys_stream<1, 3>);
            else go(cosine_keys_stream<1, 4>);
        } else if (nq_pass == 2) {
            if (nb == 1) go(cosine_keys_stream<2, 1>);
            else if (nb == 2) go(cosine_keys_stream<2, 2>);
            else if (nb == 3) go(cosine_keys_stream<2, 3>);
            else go(cosine_keys_stream<2, 4>);
        } else {
            if (nb == 1) go(cosine_keys_stream<4, 1>);
            else if (nb == 2) go(cosine_keys_stream<4, 2>);
            else if (nb == 3) go(cosine_keys_stream<4, 3>);
            else go(cosine_keys_stream<4, 4>);
        }
        return 0;
    }
    if (blocks_path(rows, dim, queries, nq_pass, n)) {
        launch_blocks(rows, norms, n, dim, queries, qnorm, nq_pass, keys, run_flag, stream);
        return 0;
    }
    if (mfma_ok(rows, dim)) {
        launch_mfma<kKeysDense>(rows, norms, n, dim, queries, qnorm, nq_pass, keys, run_flag, stream, CosinePrune{});
        return 0;
    }
    const uint32_t dim4 = (dim + 3) & ~3u;
    const size_t lds = (size_t)nq_pass * dim4 * sizeof(float);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cosine_keys),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(cosine_keys, dim3((unsigned)((n + 31) / 32)), dim3(256), lds, stream, rows, norms, n, dim,
                       queries, qnorm, nq_pass, keys);
    return 0;
}

// ---- the pass without a key matrix (see CosinePrune) ----
// Always the 16x16x4 kernel, also at <= 16 queries: without its key stores it is 3 % ahead of the 4x4x1 kernel there (0.62 vs
// 0.64 ms, 16 queries over 1 M x 768) and its list pass is one tile per entry (28 vs 52 us).
bool cosine_prune_ok(const float* rows, uint32_t dim, const float* queries, uint32_t nq_pass, size_t n, uint32_t k) {
    // 5 .. 48 queries; corpora large enough that ~k chunks are a sliver of them; k below the 256 thread minima of prune_bound_kernel
    return nq_pass > 4 && k >= 1 && k <= 64 && n >= ((size_t)1 << 17) && n < ((size_t)1 << 32) && mfma_ok(rows, dim) &&
           nq_pass <= (uint32_t)16 * mfma_groups(dim) && !gemm_path(rows, dim, queries, nq_pass);
}
// ---- the f16 minima (cosine_mins_f16) ----
float cosine_mins_eps(uint32_t dim) { return 9.9e-4f + 1.25e-7f * (float)dim; }   // the bound derived at the kernel, rounded up
uint32_t cosine_list_queries(uint32_t dim) { return (uint32_t)16 * (uint32_t)mfma_groups(dim); }
bool cosine_mins_f16_ok(const float* rows, uint32_t dim, const float* queries, uint32_t nq_pass, size_t n, uint32_t k) {
    if (!(nq_pass > 1 && nq_pass <= 64 && k >= 1 && k <= 64 && n >= ((size_t)1 << 17) && n < ((size_t)1 << 32))) return false;
    if (dim % 64 != 0 || !mfma_ok(rows, dim) || (reinterpret_cast<uintptr_t>(queries) & 15u) != 0) return false;
    const uint32_t G = (nq_pass + 15) / 16;
    if ((size_t)16 * G * ((size_t)dim * 2 + 16) + 64 + (size_t)kCW * 16 * 272 > 158u * 1024u) return false;
    // a batch above the exact list pass's image is rescored in slices and falls back to the GEMM's dense keys
    return nq_pass <= cosine_list_queries(dim) || gemm_path(rows, dim, queries, nq_pass);
}
int launch_cosine_norms_image(const float* queries, size_t nq, uint32_t dim, float* norms, void* image, uint32_t* zero2,
                              hipStream_t stream) {
    if (nq == 0) return 0;
    hipLaunchKernelGGL(cosine_norms_image, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, queries, nq, dim, norms,
                       reinterpret_cast<_Float16*>(image), zero2);
    return 0;
}
int launch_cosine_mins_f16(const float* rows, const float* norms, size_t n, uint32_t dim, const void* image, const float* qnorm,
                           uint32_t nq_pass, const CosinePrunePlan& p, uint32_t* mins, uint32_t* wmin, uint32_t* flag,
                           hipStream_t stream) {
    CosinePrune pr{};
    pr.mins = mins;
    pr.wmin = wmin;
    pr.qpad = p.qpad;
    pr.cs_shift = p.cs_shift;
    const int G = (int)((nq_pass + 15) / 16);
    const size_t lds = (size_t)16 * G * ((size_t)dim * 2 + 16) + 64 + (size_t)kCW * 16 * 272;   // image, slack for the operand prefetch past the last row, stages
    const unsigned grid = p.waves / kCW;
    auto go = [&](auto kern) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kCW * 64), lds, stream, rows, norms, n, dim,
                           reinterpret_cast<const _Float16*>(image), qnorm, nq_pass, flag, pr);
    };
    if (G == 1) go(cosine_mins_f16<1>);
    else if (G == 2) go(cosine_mins_f16<2>);
    else if (G == 3) go(cosine_mins_f16<3>);
    else go(cosine_mins_f16<4>);
    return 0;
}

CosinePrunePlan cosine_prune_plan(size_t n, uint32_t nq_pass, uint32_t k, bool approx) {
    CosinePrunePlan p;
    p.cs_shift = 4;                                              // a chunk = a 16-row tile
    p.qpad = 16u * ((nq_pass + 15) / 16);
    p.nchunks = (uint32_t)(((n - 1) >> p.cs_shift) + 1);
    unsigned grid = (unsigned)(((size_t)p.nchunks + kCW - 1) / kCW);   // as launch_mfma sizes it
    unsigned cap = 256 * 4;
    if (const char* e = approx ? getenv("UCFP_COSINE_MINS_GRID") : nullptr) {   // measurement: workgroups of the f16 minima pass
        const int v = atoi(e);
        if (v >= 1 && v <= 1024) cap = (unsigned)v;
    }
    if (grid > cap) grid = cap;
    p.waves = grid * kCW;
    p.capq = (2 * k + 31) & ~31u;                                // chunks listed per query: k of them beat the threshold, ties add a few
    if (p.capq < 32) p.capq = 32;
    if (approx) p.capq = (3 * k + 63) & ~31u;                    // ... and the chunks inside the widened threshold's margin
    return p;
}
int launch_cosine_keys_mins(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries,
                            const float* qnorm, uint32_t nq_pass, const CosinePrunePlan& p, uint32_t* mins, uint32_t* wmin,
                            hipStream_t stream) {
    CosinePrune pr{};
    pr.mins = mins;
    pr.wmin = wmin;
    pr.qpad = p.qpad;
    pr.cs_shift = p.cs_shift;
    launch_mfma<kKeysMins>(rows, norms, n, dim, queries, qnorm, nq_pass, nullptr, nullptr, stream, pr);
    return 0;
}
int launch_cosine_keys_list(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries,
                            const float* qnorm, uint32_t nq_pass, const CosinePrunePlan& p, const void* list,
                            const uint32_t* nlist, uint32_t* ckeys, const uint32_t* fallback_flag, hipStream_t stream) {
    CosinePrune pr{};
    pr.cs_shift = p.cs_shift;
    pr.qpad = p.qpad;
    pr.list = reinterpret_cast<const uint2*>(list);
    pr.nlist = nlist;
    pr.ckeys = ckeys;
    const uint32_t per = cosine_list_queries(dim);
    pr.slices = (nq_pass + per - 1) / per;
    pr.slice_q = (nq_pass + pr.slices - 1) / pr.slices;
    const int G = (int)((pr.slice_q + 15) / 16);
    const uint32_t dim16 = (dim + 15) & ~15u;
    const size_t lds = (size_t)16 * G * (dim16 + 4) * sizeof(float) + 64;
    // sized for the longest list the pass can produce (capq chunks per query, one tile per wave)
    unsigned grid = (unsigned)(((size_t)nq_pass * p.capq + kCW - 1) / kCW);
    if (grid > 1024) grid = 1024;
    grid *= pr.slices;
    auto go = [&](auto kern) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kCW * 64), lds, stream, rows, norms, n, dim, queries, qnorm, nq_pass,
                           (uint32_t*)nullptr, fallback_flag, pr);
    };
    const bool full = dim % 256 == 0;
    if (full) {
        if (G == 1) go(cosine_keys_mfma<1, true, kKeysList>);
        else if (G == 2) go(cosine_keys_mfma<2, true, kKeysList>);
        else go(cosine_keys_mfma<3, true, kKeysList>);
    } else {
        if (G == 1) go(cosine_keys_mfma<1, false, kKeysList>);
        else if (G == 2) go(cosine_keys_mfma<2, false, kKeysList>);
        else go(cosine_keys_mfma<3, false, kKeysList>);
    }
    return 0;
}
// the dense keys of the same kernel (the gated fallback of the pass above: same arithmetic, so the answer does not depend on
// which of the two produced it)
int launch_cosine_keys_dense_mfma(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries,
                                  const float* qnorm, uint32_t nq_pass, uint32_t* keys, const uint32_t* run_flag,
                                  hipStream_t stream) {
    launch_mfma<kKeysDense>(rows, norms, n, dim, queries, qnorm, nq_pass, keys, run_flag, stream, CosinePrune{});
    return 0;
}

int launch_cosine_scores_from_keys(const uint32_t* keys, size_t total, float* scores, hipStream_t stream) {
    if (total == 0) return 0;
    hipLaunchKernelGGL(cosine_scores_from_keys, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, keys,
                       total, scores);
    return 0;
}

}  // namespace ucfp
