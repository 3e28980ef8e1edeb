// topk.hip -- selection primitives shared by the Hamming and cosine searches.
//
// Keys are u32, SMALLER IS BETTER (Hamming distance, or the inverted order-image of a cosine
// score); 0xffffffff marks "no entry".  Order everywhere: (key ascending, record_id ascending).
//
//   select_topk_u32   row-parallel selection from a precomputed key row keys[q][0..n):
//                     lane = row (coalesced), ONE wave-shared candidate list in LDS and a
//                     wave-uniform threshold tau = current k-th key, so after warm-up a wave
//                     only leaves its load/compare loop for the rare row that beats tau.
//   topk_merge_u32    one wave per query merges `parts` partial lists (k rounds of a DPP wave argmin over
//                     LDS-staged candidates); used for the slices of one GPU and, after the RCCL all-gather,
//                     for the shards of a node.
//   topk_select_lists_u32  one wave per query: best k of a base list + a list of candidate row numbers.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ucfp_hip.h"
#include "common.h"

namespace ucfp {

namespace {
constexpr int kWave = 64;
constexpr int kSelCap = 256;  // wave-shared candidate slots (>= k + 64)

__device__ __forceinline__ bool key_less(uint32_t d1, uint64_t i1, uint32_t d2, uint64_t i2) {
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

// wave-wide argmin over (key, id) pairs held one per lane; every lane returns the winner.  An inclusive DPP scan
// (row_shr 1, 2, 4, 8, then the row broadcasts 15 and 31) leaves the minimum in lane 63 -- VALU moves only, where a
// ds_bpermute butterfly pays six dependent LDS-crossbar round trips of three dwords each (the k selection rounds of
// a merge are nothing but this reduction).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void argmin_step(uint32_t& bd, uint64_t& bi) {
    const int od = __builtin_amdgcn_update_dpp(-1, (int)bd, CTRL, ROW_MASK, 0xf, false);
    const int ol = __builtin_amdgcn_update_dpp(-1, (int)(uint32_t)bi, CTRL, ROW_MASK, 0xf, false);
    const int oh = __builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(bi >> 32), CTRL, ROW_MASK, 0xf, false);
    const uint32_t d2 = (uint32_t)od;
    const uint64_t i2 = ((uint64_t)(uint32_t)oh << 32) | (uint32_t)ol;   // lanes without a source see (max key, max id)
    if (key_less(d2, i2, bd, bi)) {
        bd = d2;
        bi = i2;
    }
}
__device__ __forceinline__ void wave_argmin(uint32_t& bd, uint64_t& bi) {
    argmin_step<0x111, 0xf>(bd, bi);   // row_shr:1
    argmin_step<0x112, 0xf>(bd, bi);   // row_shr:2
    argmin_step<0x114, 0xf>(bd, bi);   // row_shr:4
    argmin_step<0x118, 0xf>(bd, bi);   // row_shr:8   -> lane 15 of every row holds its row's minimum
    argmin_step<0x142, 0xa>(bd, bi);   // row_bcast:15 into rows 1 and 3
    argmin_step<0x143, 0xc>(bd, bi);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's minimum
    bd = (uint32_t)__builtin_amdgcn_readlane((int)bd, 63);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bi, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(bi >> 32), 63);
    bi = ((uint64_t)hi << 32) | lo;
}
}  // namespace

// grid (slices, nq), block 64. keys: [nq][n]; ids: [n].
// partial out: [slice][q][k] ids/keys, [slice][q] counts.
__global__ __launch_bounds__(64) void select_topk_u32(const uint32_t* __restrict__ keys,
                                                      const uint64_t* __restrict__ ids, size_t n,
                                                      size_t per_slice, uint32_t nq, uint32_t k,
                                                      uint64_t* __restrict__ part_ids,
                                                      uint32_t* __restrict__ part_key,
                                                      uint32_t* __restrict__ part_cnt,
                                                      const uint32_t* __restrict__ run_flag) {
    if (run_flag && *run_flag == 0) return;
    __shared__ uint64_t l_id[2][kSelCap];
    __shared__ uint32_t l_key[2][kSelCap];
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.y;
    const uint32_t* __restrict__ kq = keys + (size_t)q * n;
    const size_t s0 = (size_t)blockIdx.x * per_slice;
    const size_t s1 = s0 + per_slice < n ? s0 + per_slice : n;
    uint32_t tau = 0xffffffffu;  // wave-uniform: accept keys < tau, or == tau while it is a real key
    uint32_t cnt = 0;            // wave-uniform
    int cur = 0;

    // keep the best min(cnt, k) entries of list `cur`, sorted, into list `cur ^ 1`
    auto prune = [&]() {
        const int nxt = cur ^ 1;
        uint32_t ld = 0;
        uint64_t li = 0;
        bool first = true;
        uint32_t kept = 0;
        const uint32_t want = cnt < k ? cnt : k;
        for (uint32_t r = 0; r < want; r++) {
            uint32_t bd = 0xffffffffu;
            uint64_t bi = ~0ull;
            for (uint32_t e = lane; e < cnt; e += kWave) {
                const uint32_t dd = l_key[cur][e];
                const uint64_t ii = l_id[cur][e];
                if ((first || key_less(ld, li, dd, ii)) && key_less(dd, ii, bd, bi)) {
                    bd = dd;
                    bi = ii;
                }
            }
            wave_argmin(bd, bi);
            if (bd == 0xffffffffu && bi == ~0ull) break;
            if (lane == 0) {
                l_key[nxt][r] = bd;
                l_id[nxt][r] = bi;
            }
            ld = bd;
            li = bi;
            first = false;
            kept++;
        }
        wave_lds_sync();
        cnt = kept;
        cur = nxt;
        if (cnt >= k) tau = l_key[cur][k - 1];
    };

    // one candidate per lane (wave-wide step): append the hits, prune when the list nears its capacity
    auto offer = [&](uint32_t key, size_t row) {
        const bool hit = key != 0xffffffffu && key <= tau;
        const uint64_t mask = __ballot(hit);
        if (mask) {
            if (hit) {
                const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                l_key[cur][pos] = key;
                l_id[cur][pos] = ids[row];
            }
            cnt += (uint32_t)__popcll(mask);
            wave_lds_sync();
            if (cnt > (uint32_t)(kSelCap - kWave)) prune();
        }
    };
    // The scan is a chain of dependent HBM round trips if every step loads 4 bytes per lane and then decides:
    // the body reads 16 bytes per lane, two steps ahead, and looks at a quad only if one of its keys can still
    // enter the list (after the first prune almost none does).  Head and tail run key by key up to the 16-byte
    // alignment of this query's row of keys.
    size_t base = s0;
    {
        const size_t mis = ((reinterpret_cast<uintptr_t>(kq + s0) + 15) & ~(uintptr_t)15) - reinterpret_cast<uintptr_t>(kq + s0);
        size_t head = mis / 4;                                   // keys before the first aligned quad
        if (head > s1 - s0) head = s1 - s0;
        if (head) {
            const size_t row = base + lane;
            offer((size_t)lane < head ? kq[row] : 0xffffffffu, row);
            base += head;
        }
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const size_t quads = (s1 - base) / (4 * kWave);              // whole wave-steps of 4 keys per lane
    if (quads) {
        const u32x4* __restrict__ vq = reinterpret_cast<const u32x4*>(kq + base) + lane;
        u32x4 v0 = __builtin_nontemporal_load(vq);
        u32x4 v1 = quads > 1 ? __builtin_nontemporal_load(vq + kWave) : v0;
        for (size_t i = 0; i < quads; i++) {
            const u32x4 v = v0;
            v0 = v1;
            if (i + 2 < quads) v1 = __builtin_nontemporal_load(vq + (i + 2) * kWave);
            uint32_t best = v[0] < v[1] ? v[0] : v[1];
            const uint32_t b2 = v[2] < v[3] ? v[2] : v[3];
            best = best < b2 ? best : b2;
            if (__ballot(best <= tau && best != 0xffffffffu)) {
                const size_t r0 = base + i * (4 * kWave) + 4 * (size_t)lane;
#pragma unroll
                for (int c = 0; c < 4; c++) offer(v[c], r0 + c);
            }
        }
        base += quads * (4 * kWave);
    }
    for (; base < s1; base += kWave) {
        const size_t row = base + lane;
        offer(row < s1 ? kq[row] : 0xffffffffu, row);
    }
    prune();
    const size_t obase = ((size_t)blockIdx.x * nq + q) * k;
    for (uint32_t e = lane; e < k; e += kWave) {
        const bool v = e < cnt;
        part_ids[obase + e] = v ? l_id[cur][e] : ~0ull;
        part_key[obase + e] = v ? l_key[cur][e] : 0xffffffffu;
    }
    if (lane == 0) part_cnt[(size_t)blockIdx.x * nq + q] = cnt;
}

// One wave per query. parts x k candidates; invalid entries carry key 0xffffffff.
// Output: best k by (key, id) ascending, each (key, id) pair emitted once.
// PACKED: the lists are 16-byte entries {id u64, key u32, pad u32} (the wire format of the sharded search: what ONE
// all-gather moves, see shard.hip); `part_ids` then points at the entries and `part_key` is unused.
template <bool PACKED>
__global__ __launch_bounds__(64) void topk_merge_u32(const uint64_t* __restrict__ part_ids,
                                                     const uint32_t* __restrict__ part_key,
                                                     uint32_t parts, uint32_t nq, uint32_t k,
                                                     uint64_t* __restrict__ out_ids,
                                                     uint32_t* __restrict__ out_key,
                                                     uint32_t* __restrict__ out_cnt,
                                                     const uint32_t* __restrict__ run_flag, uint32_t group_parts,
                                                     float* __restrict__ hscores, uint64_t* __restrict__ missing = nullptr) {
    // hscores (optional; only with one group): Hamming similarity 1 - d / 64 of the FINAL keys of this query, written
    // whether or not the merge runs -- the gated fallback of the Hamming search ends with it, so that the distances another
    // kernel selected get their scores without a launch of their own
    if (run_flag && *run_flag == 0) {
        if (hscores)
            for (uint32_t e = threadIdx.x; e < k; e += kWave) {
                const uint32_t d = out_key[(size_t)blockIdx.x * k + e];
                hscores[(size_t)blockIdx.x * k + e] = d == 0xffffffffu ? -1.0f : 1.0f - (float)d * (1.0f / 64.0f);
            }
        return;
    }
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    if (PACKED && missing && blockIdx.x == 0 && blockIdx.y == 0) {
        // a shard whose rank could not scan joins the all-gather with 0xff bytes (shard.hip): the pad word of its first entry
        // is 0xffffffff where a packed entry carries 0.  Bit p of *missing = part p is absent from this answer (parts <= 64).
        const bool gone = (uint32_t)lane < parts && reinterpret_cast<const uint4*>(part_ids)[(size_t)lane * nq * k].w == 0xffffffffu;
        const uint64_t m = __ballot(gone);
        if (lane == 0) *missing = m;
    }
    // blockIdx.y = group of `group_parts` consecutive parts (tree merge: one output list per group and query)
    const uint32_t p_lo = blockIdx.y * group_parts;
    const uint32_t p_n = parts - p_lo < group_parts ? parts - p_lo : group_parts;
    const uint4* __restrict__ part_ent = reinterpret_cast<const uint4*>(part_ids) + (size_t)p_lo * nq * k;
    part_ids += (size_t)p_lo * nq * k;
    if (!PACKED) part_key += (size_t)p_lo * nq * k;
    out_ids += (size_t)blockIdx.y * nq * k;
    out_key += (size_t)blockIdx.y * nq * k;
    const uint32_t total = p_n * k;
    // the k selection rounds run over LDS when the group's candidates fit (they do for the tree's fan-in at the
    // usual k and for the per-GPU lists after the all-gather): read from global memory in every round, each round
    // is a chain of dependent L2 round trips -- 28 us per level for 64 lists of 10
    constexpr uint32_t kStage = 2048;
    __shared__ uint32_t s_key[kStage];
    __shared__ uint64_t s_id[kStage];
    const bool staged = total <= kStage;
    if (staged) {
        for (uint32_t c = lane; c < total; c += kWave) {
            const uint32_t p = c / k, e = c - p * k;
            const size_t off = ((size_t)p * nq + q) * k + e;
            if (PACKED) {
                const uint4 en = part_ent[off];
                s_key[c] = en.z;
                s_id[c] = ((uint64_t)en.y << 32) | en.x;
            } else {
                s_key[c] = part_key[off];
                s_id[c] = part_ids[off];
            }
        }
        wave_lds_sync();
    }
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
            if (staged) {
                dd = s_key[c];
                ii = s_id[c];
            } else {
                const uint32_t p = c / k, e = c - p * k;
                const size_t off = ((size_t)p * nq + q) * k + e;
                if (PACKED) {
                    const uint4 en = part_ent[off];
                    dd = en.z;
                    ii = ((uint64_t)en.y << 32) | en.x;
                } else {
                    dd = part_key[off];
                    ii = part_ids[off];
                }
            }
            if (dd == 0xffffffffu) continue;
            if ((first || key_less(ld, li, dd, ii)) && key_less(dd, ii, bd, bi)) {
                bd = dd;
                bi = ii;
            }
        }
        wave_argmin(bd, bi);
        if (bd == 0xffffffffu) break;
        if (lane == 0) {
            out_ids[(size_t)q * k + r] = bi;
            out_key[(size_t)q * k + r] = bd;
            if (hscores) hscores[(size_t)q * k + r] = 1.0f - (float)bd * (1.0f / 64.0f);
        }
        ld = bd;
        li = bi;
        first = false;
        emitted++;
    }
    if (lane == 0) {
        for (uint32_t r = emitted; r < k; r++) {
            out_ids[(size_t)q * k + r] = ~0ull;
            out_key[(size_t)q * k + r] = 0xffffffffu;
            if (hscores) hscores[(size_t)q * k + r] = -1.0f;
        }
        if (out_cnt) out_cnt[(size_t)blockIdx.y * nq + q] = emitted;
    }
}

// ---- few queries over a dense key matrix: chunk minima, a threshold, a gather ---------------------------------------------
// With one to sixteen queries the slice lists + merge tree above are a chain of launches whose latencies (60-110 us) show
// next to a 0.5 ms corpus pass.  Here: (1) one wave per chunk of `chunk` keys writes the chunk's minimum; (2) one workgroup
// per query finds tau = the k-th smallest chunk minimum by an 8-bit radix select in LDS -- at least k keys are <= tau, so
// the k best keys all are -- and gathers the keys <= tau from the chunks whose minimum is <= tau (about k of them), ranks
// the handful of candidates by (key, id) and writes the answer.  Exact for any input: when candidates pile up (ties) the
// list is pruned to its best k and the threshold tightened, the way select_topk_u32 does it.
constexpr uint32_t kPruneMaxChunks = 4096;   // chunk minima per query held in LDS
constexpr uint32_t kPruneMaxQueries = 16;    // one workgroup per query: beyond this the slice lists fill the chip as well (32 / 48 / 64 queries measured level)
constexpr uint32_t kPruneBatch = 4;          // pieces of 1024 keys gathered between two checks of the list
constexpr uint32_t kPruneKeep = 1024;        // the list is pruned to its best k once it holds more than this
constexpr uint32_t kPruneCap = kPruneKeep + kPruneBatch * 1024;   // candidate slots: a trip always fits

__global__ __launch_bounds__(64) void chunk_min_u32(const uint32_t* __restrict__ keys, size_t n, size_t chunk, uint32_t nchunks,
                                                    uint32_t* __restrict__ mins, const uint32_t* __restrict__ run_flag) {
    if (run_flag && *run_flag == 0) return;
    const uint32_t q = blockIdx.y, c = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t* __restrict__ kq = keys + (size_t)q * n;
    const size_t s0 = (size_t)c * chunk, s1 = s0 + chunk < n ? s0 + chunk : n;
    uint32_t m = 0xffffffffu;
    // the key matrix of a query starts at a multiple of 4 keys only if n is one: 16-byte loads when it is
    if ((n & 3) == 0) {
        for (size_t i = s0 + (size_t)lane * 4; i < s1; i += 64 * 4) {
            if (i + 4 <= s1) {
                const uint4 v = *reinterpret_cast<const uint4*>(kq + i);
                m = min(min(m, min(v.x, v.y)), min(v.z, v.w));
            } else {
                for (size_t j = i; j < s1; j++) m = min(m, kq[j]);
            }
        }
    } else {
        for (size_t i = s0 + lane; i < s1; i += 64) m = min(m, kq[i]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off, kWave));
    if (lane == 0) mins[(size_t)q * nchunks + c] = m;
}

__global__ __launch_bounds__(256) void select_pruned_u32(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ ids,
                                                         size_t n, size_t chunk, uint32_t nchunks,
                                                         const uint32_t* __restrict__ mins, uint32_t nq, uint32_t k,
                                                         uint64_t* __restrict__ out_ids, uint32_t* __restrict__ out_key,
                                                         uint32_t* __restrict__ out_cnt, const uint32_t* __restrict__ run_flag) {
    if (run_flag && *run_flag == 0) return;
    __shared__ uint32_t s_min[kPruneMaxChunks];      // chunk minima, then the list of qualifying chunks
    __shared__ uint32_t s_key[kPruneCap];
    __shared__ uint64_t s_id[kPruneCap];
    __shared__ uint32_t s_tkey[UCFP_INDEX_MAX_K];
    __shared__ uint64_t s_tid[UCFP_INDEX_MAX_K];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_sel[2];                    // radix select: chosen bin, rank left inside it
    __shared__ uint32_t s_wtot[4];
    __shared__ uint32_t s_n;                         // candidates in the list
    __shared__ uint32_t s_nqual;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t* __restrict__ kq = keys + (size_t)q * n;

    for (uint32_t c = tid; c < nchunks; c += 256) s_min[c] = mins[(size_t)q * nchunks + c];
    if (tid == 0) s_n = 0, s_nqual = 0;
    __syncthreads();
    // ---- tau = the kk-th smallest chunk minimum, most significant byte first
    uint32_t tau = 0xffffffffu;
    if (nchunks > k) {
        uint32_t prefix = 0, mask = 0, want = k;     // rank (1-based) still to find among the minima matching `prefix`
        for (int shift = 24; shift >= 0; shift -= 8) {
            s_hist[tid] = 0;
            __syncthreads();
            for (uint32_t c = tid; c < nchunks; c += 256) {
                const uint32_t v = s_min[c];
                if ((v & mask) == prefix) atomicAdd(&s_hist[(v >> shift) & 255u], 1u);
            }
            __syncthreads();
            {   // the bin holding rank `want`: inclusive scan of the 256 counts, one per thread
                const uint32_t cntb = s_hist[tid];
                uint32_t incl = cntb;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = (uint32_t)__shfl_up((int)incl, off, kWave);
                    if ((int)(tid & 63) >= off) incl += o;
                }
                if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
                __syncthreads();
                uint32_t before = 0;
                for (uint32_t w = 0; w < (tid >> 6); w++) before += s_wtot[w];
                incl += before;
                if (incl - cntb < want && want <= incl) {      // exactly one thread: the minima matching `prefix` number >= want
                    s_sel[0] = tid;
                    s_sel[1] = want - (incl - cntb);
                }
            }
            __syncthreads();
            prefix |= s_sel[0] << shift;
            mask |= 255u << shift;
            want = s_sel[1];
        }
        tau = prefix;
    }
    // ---- the chunks that can hold a key <= tau, compacted in place behind a barrier (order does not matter)
    uint32_t mine[kPruneMaxChunks / 256];
#pragma unroll
    for (uint32_t u = 0; u < kPruneMaxChunks / 256; u++) {
        const uint32_t c = tid + u * 256;
        mine[u] = c < nchunks ? s_min[c] : 0xffffffffu;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < kPruneMaxChunks / 256; u++) {
        const uint32_t c = tid + u * 256;
        if (c < nchunks && mine[u] <= tau && mine[u] != 0xffffffffu) s_min[atomicAdd(&s_nqual, 1u)] = c;
    }
    __syncthreads();
    const uint32_t nqual = s_nqual;

    // keep the best min(n, k) candidates, sorted by (key, id), at the front of the list; returns their number
    auto prune = [&]() -> uint32_t {
        const uint32_t cn = s_n < kPruneCap ? s_n : kPruneCap;
        // rank of every candidate = the number of candidates before it by (key, id, list position): a permutation even
        // when an APPEND_ONLY shard holds the same id twice
        for (uint32_t e = tid; e < cn; e += 256) {
            const uint32_t dk = s_key[e];
            const uint64_t di = s_id[e];
            uint32_t rank = 0;
            for (uint32_t o = 0; o < cn; o++) {
                const uint32_t ok = s_key[o];
                const uint64_t oi = s_id[o];
                rank += (key_less(ok, oi, dk, di) || (ok == dk && oi == di && o < e)) ? 1u : 0u;
            }
            if (rank < k) {
                s_tkey[rank] = dk;
                s_tid[rank] = di;
            }
        }
        __syncthreads();
        const uint32_t kept = cn < k ? cn : k;
        if (tid < kept) {
            s_key[tid] = s_tkey[tid];
            s_id[tid] = s_tid[tid];
        }
        if (tid == 0) s_n = kept;
        __syncthreads();
        return kept;
    };

    // ---- gather: kPruneBatch x 1024 keys per trip, every thread one group of 4 keys per chunk piece
    uint64_t tau_id = ~0ull;   // with key == tau: ids below this one still count (no id is ~0)
    const uint32_t pieces_per_chunk = (uint32_t)((chunk + 1023) / 1024);
    const uint32_t npieces = nqual * pieces_per_chunk;
    for (uint32_t p0 = 0; p0 < npieces; p0 += kPruneBatch) {
        uint32_t kv[kPruneBatch][4];
        size_t at[kPruneBatch];
#pragma unroll
        for (uint32_t u = 0; u < kPruneBatch; u++) {
            const uint32_t pc = p0 + u;
            at[u] = n;
#pragma unroll
            for (int j = 0; j < 4; j++) kv[u][j] = 0xffffffffu;
            if (pc < npieces) {
                const uint32_t c = s_min[pc / pieces_per_chunk], piece = pc % pieces_per_chunk;
                const size_t c1 = (size_t)c * chunk + chunk < n ? (size_t)c * chunk + chunk : n;
                const size_t i = (size_t)c * chunk + (size_t)piece * 1024 + (size_t)tid * 4;
                at[u] = i;
                if ((n & 3) == 0 && i + 4 <= c1) {
                    const uint4 v = *reinterpret_cast<const uint4*>(kq + i);
                    kv[u][0] = v.x, kv[u][1] = v.y, kv[u][2] = v.z, kv[u][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (i + j < c1) kv[u][j] = kq[i + j];
                }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < kPruneBatch; u++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (kv[u][j] != 0xffffffffu && kv[u][j] <= tau) {
                    const uint64_t id = ids[at[u] + j];
                    // once the list has been pruned, a key equal to the k-th best only matters with a smaller id (a corpus
                    // of copies would otherwise refill the list on every trip)
                    if (kv[u][j] == tau && id >= tau_id) continue;
                    const uint32_t pos = atomicAdd(&s_n, 1u);
                    if (pos < kPruneCap) {
                        s_key[pos] = kv[u][j];
                        s_id[pos] = id;
                    }
                }
        __syncthreads();
        // a trip adds at most kPruneBatch x 1024 candidates: prune while another one might not fit.  The count is
        // snapshotted behind a second barrier: without it a fast wave could fall through, start the next trip and bump
        // s_n past the limit before a slow wave has read it -- the waves would then disagree about the branch (and
        // about the barriers inside prune()).
        const uint32_t cur_n = s_n;
        __syncthreads();
        if (cur_n > kPruneKeep) {                        // block-uniform
            // (an overfull list cannot happen: the check runs after every trip and a trip fits behind the last prune)
            const uint32_t kept = prune();
            if (kept >= k) tau = s_key[k - 1], tau_id = s_id[k - 1];   // the k-th best itself is in the list already
            __syncthreads();
        }
    }
    const uint32_t kept = prune();
    for (uint32_t r = tid; r < k; r += 256) {
        out_ids[(size_t)q * k + r] = r < kept ? s_id[r] : ~0ull;
        out_key[(size_t)q * k + r] = r < kept ? s_key[r] : 0xffffffffu;
    }
    if (tid == 0 && out_cnt) out_cnt[q] = kept;
}

// ---- the same idea without a key matrix (cosine.hip CosinePrune): the keys kernel itself wrote the chunk minima ----
// Three small kernels turn them into thresholds and chunk lists:
//   prune_bound_kernel    per query: the k-th smallest of the keys kernel's per-WAVE minima -- an upper bound on the k-th
//                         smallest chunk minimum (k waves hold a key at most that large)
//   prune_collect_kernel  one coalesced pass over mins[chunk][qpad]: the chunks at or below their query's bound, a handful
//                         each, to cand[q]
//   prune_tau_kernel      per query: tau = the k-th smallest chunk minimum, ranked exactly among the candidates; the chunks
//                         whose minimum is <= tau go to the global list as (query, chunk)
// (A workgroup per query that reads its own column of the minima is bound by its CU's address path, one line per lane: 80 us
// at 16 queries, 165 at 32; a radix select over all minima serialises on one histogram bin, cosine keys share their leading
// bytes: 51 us.)
// Minima that are approximate within eps of the exact score (cosine.hip cosine_mins_f16): a threshold taken from them is
// widened by eps where it bounds EXACT keys from above (k chunks hold a row whose exact score is at least score(threshold) -
// eps) and by 2 eps where it selects chunks by their APPROXIMATE minima (a row that good scores at least score - 2 eps
// approximately).  eps = 0: the minima are exact and nothing moves.  Keys are the inverted order image of the f32 score.
__device__ __forceinline__ uint32_t relax_key(uint32_t key, float d) {
    if (d == 0.f || key == 0xffffffffu) return key;
    uint32_t u = ~key;
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    const float s = __uint_as_float(u) - d;
    uint32_t v = __float_as_uint(s);
    v = (v & 0x80000000u) ? ~v : (v | 0x80000000u);
    const uint32_t r = ~v;
    return r < key ? key : r;      // (never tighter)
}

__global__ __launch_bounds__(256) void prune_bound_kernel(const uint32_t* __restrict__ wmin, uint32_t waves, uint32_t k,
                                                          uint32_t* __restrict__ bound, uint32_t* __restrict__ ccnt, float eps) {
    // 256 threads (k <= 64 < 256): 1024 of them ranking 1024 minima against each other took 46 us on their one CU
    __shared__ __attribute__((aligned(16))) uint32_t s_tmin[256];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    uint32_t m = 0xffffffffu;
    // (eight loads in flight per thread: one after the other the 32 of a 8192-wave launch were most of this kernel's 12 us)
    const uint32_t* __restrict__ wq = wmin + (size_t)q * waves;
    for (uint32_t w0 = tid; w0 < waves; w0 += 8 * 256) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = w0 + u * 256 < waves ? wq[w0 + u * 256] : 0xffffffffu;
#pragma unroll
        for (int u = 0; u < 8; u++) m = min(m, v[u]);
    }
    s_tmin[tid] = m;
    if (tid == 0) bound[q] = 0xffffffffu, ccnt[q] = 0;
    __syncthreads();
    // rank of this thread's minimum among the 256 (ties by thread number): the one of rank k - 1 is the bound
    uint32_t rank = 0;
    for (uint32_t u = 0; u < 256; u += 4) {
        const uint4 o = *reinterpret_cast<const uint4*>(s_tmin + u);
        rank += (o.x < m || (o.x == m && u < tid)) ? 1u : 0u;
        rank += (o.y < m || (o.y == m && u + 1 < tid)) ? 1u : 0u;
        rank += (o.z < m || (o.z == m && u + 2 < tid)) ? 1u : 0u;
        rank += (o.w < m || (o.w == m && u + 3 < tid)) ? 1u : 0u;
    }
    if (rank == k - 1) bound[q] = relax_key(m, 2.f * eps);   // 0xffffffff: fewer than k threads saw a scored row -- no threshold
}

// thread = four consecutive queries of one chunk (a uint4 of mins[chunk][qpad]; qpad is a multiple of 16)
__global__ __launch_bounds__(256) void prune_collect_kernel(const uint32_t* __restrict__ mins, uint32_t nchunks, uint32_t qpad,
                                                            uint32_t nq, const uint32_t* __restrict__ bound,
                                                            uint32_t* __restrict__ ccnt, uint2* __restrict__ cand) {
    __shared__ uint32_t s_b[64];
    if (threadIdx.x < 64) s_b[threadIdx.x] = threadIdx.x < nq ? bound[threadIdx.x] : 0u;
    __syncthreads();
    const uint32_t per = qpad / 4;
    const size_t total = (size_t)nchunks * per;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t c = (uint32_t)(i / per), q0 = (uint32_t)(i % per) * 4;
        const uint4 v = *reinterpret_cast<const uint4*>(mins + i * 4);
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t q = q0 + j;
            if (q < nq && vv[j] <= s_b[q] && vv[j] != 0xffffffffu) {
                const uint32_t pos = atomicAdd(&ccnt[q], 1u);
                if (pos < kPruneCand) cand[(size_t)q * kPruneCand + pos] = make_uint2(vv[j], c);
            }
        }
    }
}

__global__ __launch_bounds__(256) void prune_tau_kernel(const uint32_t* __restrict__ bound, const uint32_t* __restrict__ ccnt,
                                                        const uint2* __restrict__ cand, uint32_t k, uint32_t capq,
                                                        uint32_t* __restrict__ tau_out, uint2* __restrict__ list,
                                                        uint32_t* __restrict__ nlist, uint2* __restrict__ qrange,
                                                        uint32_t* __restrict__ flag, float eps) {
    __shared__ uint32_t s_ck[kPruneCand], s_cc[kPruneCand];
    __shared__ uint32_t s_tau, s_cnt, s_put, s_base;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t cn = ccnt[q];
    // no bound (fewer than k scored waves), ties in bulk at the bound, or -- cannot happen -- fewer than k candidates: dense path
    if (bound[q] == 0xffffffffu || cn > kPruneCand || cn < k) {
        if (tid == 0) {
            *flag = 1;
            tau_out[q] = 0xffffffffu;
            qrange[q] = make_uint2(0, 0);
        }
        return;
    }
    for (uint32_t e = tid; e < cn; e += 256) {
        const uint2 v = cand[(size_t)q * kPruneCand + e];
        s_ck[e] = v.x;
        s_cc[e] = v.y;
    }
    if (tid == 0) s_tau = 0xffffffffu, s_cnt = 0, s_put = 0;
    __syncthreads();
    for (uint32_t e = tid; e < cn; e += 256) {
        const uint32_t v = s_ck[e];
        uint32_t rank = 0;
        for (uint32_t o = 0; o < cn; o++) rank += (s_ck[o] < v || (s_ck[o] == v && o < e)) ? 1u : 0u;
        if (rank == k - 1) s_tau = v;
    }
    __syncthreads();
    const uint32_t tau_exact = relax_key(s_tau, eps);      // bounds the k-th best exact key
    const uint32_t tau = relax_key(s_tau, 2.f * eps);      // selects chunks by their (approximate) minima
    for (uint32_t e = tid; e < cn; e += 256)
        if (s_ck[e] <= tau) atomicAdd(&s_cnt, 1u);
    __syncthreads();
    const uint32_t total = s_cnt;
    if (total > capq) {           // ties in bulk at the threshold: the dense path prunes them while it gathers
        if (tid == 0) {
            *flag = 1;
            tau_out[q] = tau_exact;
            qrange[q] = make_uint2(0, 0);
        }
        return;
    }
    if (tid == 0) {
        s_base = atomicAdd(nlist, total);
        tau_out[q] = tau_exact;
    }
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t e = tid; e < cn; e += 256)
        if (s_ck[e] <= tau) list[base + atomicAdd(&s_put, 1u)] = make_uint2(q, s_cc[e]);
    if (tid == 0) qrange[q] = make_uint2(base, total);
}

// prune_final_kernel: one workgroup per query: the listed chunks' keys <= tau with their ids, ranked by (key, id).
constexpr uint32_t kFinalCap = 4096;
__global__ __launch_bounds__(256) void prune_final_kernel(const uint32_t* __restrict__ ckeys, uint32_t cs_shift,
                                                          const uint2* __restrict__ list, const uint2* __restrict__ qrange,
                                                          const uint32_t* __restrict__ tau_in, const uint64_t* __restrict__ ids,
                                                          size_t n, uint32_t k, uint64_t* __restrict__ out_ids,
                                                          uint32_t* __restrict__ out_key, uint32_t* __restrict__ out_cnt,
                                                          uint32_t* __restrict__ flag) {
    if (*flag) return;
    __shared__ uint32_t s_key[kFinalCap];
    __shared__ uint64_t s_id[kFinalCap];
    __shared__ uint32_t s_tkey[UCFP_INDEX_MAX_K];
    __shared__ uint64_t s_tid[UCFP_INDEX_MAX_K];
    __shared__ uint32_t s_n;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint2 rg = qrange[q];
    const uint32_t tau = tau_in[q];
    if (tid == 0) s_n = 0;
    __syncthreads();
    const uint32_t cs = 1u << cs_shift;
    const size_t total = (size_t)rg.y << cs_shift;
    for (size_t i = tid; i < total; i += 256) {
        const uint32_t e = (uint32_t)(i >> cs_shift), r = (uint32_t)i & (cs - 1);
        const uint32_t key = ckeys[((size_t)(rg.x + e) << cs_shift) + r];
        const size_t row = ((size_t)list[rg.x + e].y << cs_shift) + r;
        if (row < n && key <= tau && key != 0xffffffffu) {
            const uint32_t pos = atomicAdd(&s_n, 1u);
            if (pos < kFinalCap) {
                s_key[pos] = key;
                s_id[pos] = ids[row];
            }
        }
    }
    __syncthreads();
    const uint32_t cn = s_n;
    if (cn > kFinalCap) {          // more keys at or below the threshold than the list holds: the dense path (launched after this
        if (tid == 0) *flag = 1;   // kernel, gated on the flag) answers
        return;
    }
    // rank of every candidate = the number of candidates before it by (key, id, list position): a permutation even when an
    // APPEND_ONLY shard holds the same id twice
    for (uint32_t e = tid; e < cn; e += 256) {
        const uint32_t dk = s_key[e];
        const uint64_t di = s_id[e];
        uint32_t rank = 0;
        for (uint32_t o = 0; o < cn; o++) {
            const uint32_t ok = s_key[o];
            const uint64_t oi = s_id[o];
            rank += (key_less(ok, oi, dk, di) || (ok == dk && oi == di && o < e)) ? 1u : 0u;
        }
        if (rank < k) {
            s_tkey[rank] = dk;
            s_tid[rank] = di;
        }
    }
    __syncthreads();
    const uint32_t kept = cn < k ? cn : k;
    for (uint32_t r = tid; r < k; r += 256) {
        out_ids[(size_t)q * k + r] = r < kept ? s_tid[r] : ~0ull;
        out_key[(size_t)q * k + r] = r < kept ? s_tkey[r] : 0xffffffffu;
    }
    if (tid == 0 && out_cnt) out_cnt[q] = kept;
}

size_t prune_tau_ws_bytes(uint32_t nq) { return (size_t)nq * 8 + 256 + (size_t)nq * kPruneCand * 8; }
int launch_prune_tau(const uint32_t* mins, const uint32_t* wmin, const CosinePrunePlan& p, uint32_t nq, uint32_t k, uint8_t* ws,
                     uint32_t* tau, void* list, uint32_t* nlist, void* qrange, uint32_t* flag, hipStream_t stream, float eps) {
    if (nq == 0) return 0;
    uint32_t* bound = reinterpret_cast<uint32_t*>(ws);
    uint32_t* ccnt = bound + nq;
    uint2* cand = reinterpret_cast<uint2*>(ws + (((size_t)nq * 8 + 255) & ~(size_t)255));
    hipLaunchKernelGGL(prune_bound_kernel, dim3(nq), dim3(256), 0, stream, wmin, p.waves, k, bound, ccnt, eps);
    const size_t units = (size_t)p.nchunks * (p.qpad / 4);
    unsigned grid = (unsigned)((units + 255) / 256);
    if (grid > 256 * 8) grid = 256 * 8;
    hipLaunchKernelGGL(prune_collect_kernel, dim3(grid), dim3(256), 0, stream, mins, p.nchunks, p.qpad, nq,
                       (const uint32_t*)bound, ccnt, cand);
    hipLaunchKernelGGL(prune_tau_kernel, dim3(nq), dim3(256), 0, stream, (const uint32_t*)bound, (const uint32_t*)ccnt,
                       (const uint2*)cand, k, p.capq, tau, reinterpret_cast<uint2*>(list), nlist,
                       reinterpret_cast<uint2*>(qrange), flag, eps);
    return 0;
}
int launch_prune_final(const uint32_t* ckeys, const CosinePrunePlan& p, const void* list, const void* qrange,
                       const uint32_t* tau, const uint64_t* ids, size_t n, uint32_t nq, uint32_t k, uint64_t* out_ids,
                       uint32_t* out_key, uint32_t* out_cnt, uint32_t* flag, hipStream_t stream) {
    if (nq == 0) return 0;
    hipLaunchKernelGGL(prune_final_kernel, dim3(nq), dim3(256), 0, stream, ckeys, p.cs_shift,
                       reinterpret_cast<const uint2*>(list), reinterpret_cast<const uint2*>(qrange), tau, ids, n, k, out_ids,
                       out_key, out_cnt, flag);
    return 0;
}

size_t select_pruned_chunk(size_t n) {
    size_t chunk = 1024;
    while ((n + chunk - 1) / chunk > kPruneMaxChunks) chunk *= 2;
    return chunk;
}
size_t select_pruned_ws_bytes(size_t, uint32_t nq) { return (size_t)kPruneMaxChunks * nq * 4 + 256; }
bool select_pruned_ok(size_t n, uint32_t nq, uint32_t k) { return nq >= 1 && nq <= kPruneMaxQueries && k >= 1 && k <= UCFP_INDEX_MAX_K && n >= 1; }

int launch_select_pruned_u32(const uint32_t* keys, const uint64_t* ids, size_t n, uint32_t nq, uint32_t k, uint32_t* mins,
                             uint64_t* out_ids, uint32_t* out_key, uint32_t* out_cnt, hipStream_t stream,
                             const uint32_t* run_flag) {
    if (nq == 0) return 0;
    const size_t chunk = select_pruned_chunk(n);
    const uint32_t nchunks = (uint32_t)((n + chunk - 1) / chunk);
    hipLaunchKernelGGL(chunk_min_u32, dim3(nchunks, nq), dim3(64), 0, stream, keys, n, chunk, nchunks, mins, run_flag);
    hipLaunchKernelGGL(select_pruned_u32, dim3(nq), dim3(256), 0, stream, keys, ids, n, chunk, nchunks,
                       (const uint32_t*)mins, nq, k, out_ids, out_key, out_cnt, run_flag);
    return 0;
}

SelectPlan select_plan(size_t n, uint32_t nq) {
    SelectPlan p;
    const uint32_t want_waves = 256 * 16;   // measured 8 / 16 / 32 / 64 per CU at 16 and 48 queries over 1 M keys: 16 is fastest (longer slices amortise the list warm-up)
    uint32_t slices = nq ? (want_waves + nq - 1) / nq : 1;
    const size_t max_slices = (n + 511) / 512;
    if (slices > max_slices) slices = (uint32_t)(max_slices ? max_slices : 1);
    if (slices < 1) slices = 1;
    p.per_slice = (n + slices - 1) / slices;
    p.per_slice = (p.per_slice + 63) & ~(size_t)63;
    p.slices = (uint32_t)((n + p.per_slice - 1) / (p.per_slice ? p.per_slice : 1));
    if (p.slices < 1) p.slices = 1;
    return p;
}

int launch_select_topk_u32(const uint32_t* keys, const uint64_t* ids, size_t n, const SelectPlan& p,
                           uint32_t nq, uint32_t k, uint64_t* part_ids, uint32_t* part_key,
                           uint32_t* part_cnt, hipStream_t stream, const uint32_t* run_flag) {
    if (nq == 0) return 0;
    hipLaunchKernelGGL(select_topk_u32, dim3(p.slices, nq), dim3(64), 0, stream, keys, ids, n, p.per_slice, nq,
                       k, part_ids, part_key, part_cnt, run_flag);
    return 0;
}

// One wave per query: the best k by (key, id) of a base list (k entries, invalid ones carry key 0xffffffff) and a
// candidate list of row numbers (at most cap of the cnt[q] produced are stored).  A query whose candidates did not
// all fit raises *overflow (the caller then re-runs the dense path); it still writes its incomplete answer.
constexpr int kListCap = 1024;
__global__ __launch_bounds__(64) void topk_select_lists_u32(const uint64_t* __restrict__ base_ids,
                                                            const uint32_t* __restrict__ base_key,
                                                            const uint32_t* __restrict__ ckey,
                                                            const uint32_t* __restrict__ crow,
                                                            const uint32_t* __restrict__ ccnt, uint32_t cap,
                                                            const uint64_t* __restrict__ ids, uint32_t k,
                                                            uint64_t* __restrict__ out_ids,
                                                            uint32_t* __restrict__ out_key,
                                                            uint32_t* __restrict__ out_cnt,
                                                            uint32_t* __restrict__ overflow) {
    __shared__ uint64_t l_id[kListCap + 64];
    __shared__ uint32_t l_key[kListCap + 64];
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    uint32_t c = ccnt[q];
    if (c > cap) {
        if (lane == 0) atomicOr(overflow, 1u);
        c = cap;
    }
    for (uint32_t e = lane; e < c; e += kWave) {
        l_key[e] = ckey[(size_t)q * cap + e];
        l_id[e] = ids[crow[(size_t)q * cap + e]];
    }
    for (uint32_t e = lane; e < k; e += kWave) {
        l_key[c + e] = base_key[(size_t)q * k + e];
        l_id[c + e] = base_ids[(size_t)q * k + e];
    }
    const uint32_t total = c + k;
    wave_lds_sync();
    uint32_t ld = 0, kept = 0;
    uint64_t li = 0;
    bool first = true;
    for (uint32_t r = 0; r < k; r++) {
        uint32_t bd = 0xffffffffu;
        uint64_t bi = ~0ull;
        for (uint32_t e = lane; e < total; e += kWave) {
            const uint32_t dd = l_key[e];
            const uint64_t ii = l_id[e];
            if (dd != 0xffffffffu && (first || key_less(ld, li, dd, ii)) && key_less(dd, ii, bd, bi)) {
                bd = dd;
                bi = ii;
            }
        }
        wave_argmin(bd, bi);
        if (bd == 0xffffffffu && bi == ~0ull) break;
        if (lane == 0) {
            out_key[(size_t)q * k + r] = bd;
            out_ids[(size_t)q * k + r] = bi;
        }
        ld = bd;
        li = bi;
        first = false;
        kept++;
    }
    for (uint32_t e = kept + lane; e < k; e += kWave) {
        out_key[(size_t)q * k + e] = 0xffffffffu;
        out_ids[(size_t)q * k + e] = ~0ull;
    }
    if (lane == 0) out_cnt[q] = kept;
}

int launch_topk_select_lists_u32(const uint64_t* base_ids, const uint32_t* base_key, const uint32_t* ckey,
                                 const uint32_t* crow, const uint32_t* ccnt, uint32_t cap, const uint64_t* ids,
                                 uint32_t nq, uint32_t k, uint64_t* out_ids, uint32_t* out_key, uint32_t* out_cnt,
                                 uint32_t* overflow, hipStream_t stream) {
    if (nq == 0) return 0;
    hipLaunchKernelGGL(topk_select_lists_u32, dim3(nq), dim3(64), 0, stream, base_ids, base_key, ckey, crow, ccnt,
                       cap < (uint32_t)kListCap ? cap : (uint32_t)kListCap, ids, k, out_ids, out_key, out_cnt, overflow);
    return 0;
}

int launch_topk_merge_u32(const uint64_t* part_ids, const uint32_t* part_key, uint32_t parts, uint32_t nq,
                          uint32_t k, uint64_t* out_ids, uint32_t* out_key, uint32_t* out_cnt,
                          const uint32_t* run_flag, hipStream_t stream, float* hamming_scores) {
    if (nq == 0) return 0;
    hipLaunchKernelGGL(topk_merge_u32<false>, dim3(nq), dim3(64), 0, stream, part_ids, part_key, parts, nq, k, out_ids,
                       out_key, out_cnt, run_flag, parts, hamming_scores);
    return 0;
}

// ---- the sharded search's wire format: one 16-byte entry per (query, place) ----
__global__ void topk_pack_entries_kernel(const uint64_t* __restrict__ ids, const uint32_t* __restrict__ keys, size_t total,
                                         uint4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t id = ids[i];
    out[i] = make_uint4((uint32_t)id, (uint32_t)(id >> 32), keys[i], 0u);
}

int launch_topk_pack_entries(const uint64_t* ids, const uint32_t* keys, size_t total, void* entries, hipStream_t stream) {
    if (total == 0) return 0;
    hipLaunchKernelGGL(topk_pack_entries_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ids, keys,
                       total, reinterpret_cast<uint4*>(entries));
    return 0;
}

// merge `parts` packed lists ([parts][nq][k] entries, as all-gathered) into the final (ids, keys, counts)
int launch_topk_merge_packed(const void* entries, uint32_t parts, uint32_t nq, uint32_t k, uint64_t* out_ids,
                             uint32_t* out_key, uint32_t* out_cnt, hipStream_t stream, uint64_t* missing) {
    if (nq == 0) return 0;
    hipLaunchKernelGGL(topk_merge_u32<true>, dim3(nq), dim3(64), 0, stream, reinterpret_cast<const uint64_t*>(entries),
                       (const uint32_t*)nullptr, parts, nq, k, out_ids, out_key, out_cnt, (const uint32_t*)nullptr, parts, (float*)nullptr,
                       missing);
    return 0;
}

// Tree merge for many parts: groups of kMergeFan parts are merged in parallel into `tmp_*`
// ([groups][nq][k]), then the group lists.  One wave walking thousands of lists serially is the slow part
// of a single-query search otherwise.
constexpr uint32_t kMergeFan = 64;
size_t topk_merge_tmp_entries(uint32_t parts, uint32_t nq, uint32_t k) {
    return parts > kMergeFan ? (size_t)((parts + kMergeFan - 1) / kMergeFan) * nq * k : 0;
}
int launch_topk_merge_tree_u32(const uint64_t* part_ids, const uint32_t* part_key, uint32_t parts, uint32_t nq,
                               uint32_t k, uint64_t* tmp_ids, uint32_t* tmp_key, uint64_t* out_ids, uint32_t* out_key,
                               uint32_t* out_cnt, hipStream_t stream, const uint32_t* run_flag) {
    if (nq == 0) return 0;
    if (parts <= kMergeFan)
        return launch_topk_merge_u32(part_ids, part_key, parts, nq, k, out_ids, out_key, out_cnt, run_flag, stream, nullptr);
    const uint32_t groups = (parts + kMergeFan - 1) / kMergeFan;
    hipLaunchKernelGGL(topk_merge_u32<false>, dim3(nq, groups), dim3(64), 0, stream, part_ids, part_key, parts, nq, k, tmp_ids,
                       tmp_key, (uint32_t*)nullptr, run_flag, kMergeFan, (float*)nullptr);
    return launch_topk_merge_tree_u32(tmp_ids, tmp_key, groups, nq, k, tmp_ids + (size_t)groups * nq * k,
                                      tmp_key + (size_t)groups * nq * k, out_ids, out_key, out_cnt, stream, run_flag);
}

}  // namespace ucfp
