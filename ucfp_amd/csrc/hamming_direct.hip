// hamming_direct.hip -- exact Hamming top-k for ONE to EIGHT queries in a single launch (gfx950).
//
// The reference's /v1/query carries one query per request (src/server/handlers.rs:143-159).  The staged search of
// hamming.hip (sample -> thresholds -> filter stages -> rescan -> select) is built for batches: for a single query its
// chain of ~17 dependent launches costs 75 us over a 12.5 M-code shard whose bytes stream in 12.5 us.  This file is the
// request-shaped path: ONE kernel, no sample, no stages, no workspace traffic besides 256 short lists.
//
//   stream   one workgroup of 16 waves per CU; a wave takes 512-code trips round-robin (4 KiB contiguous per wave,
//            64 KiB per workgroup and round; 16-byte non-temporal loads, the next trip's four loads in flight while this
//            one is evaluated).  lane = 8 codes; the queries sit in SGPRs; a pair costs 2 v_xor + 2 v_bcnt, a trip and
//            query one min tree and ONE compare against the threshold, branch-free over the queries: HBM-bound up to
//            ~8 queries.
//   lists    every wave keeps, per query, a list of <= 64 candidates (d, row) in LDS -- no record id is fetched when a
//            row is listed; the tie-break keys (ids -- or the rows themselves where ids ascend with the row, see `asc`
//            in the kernel) are fetched once, when the wave's stream ends.
//   bound    tau[q], an upper bound of the final k-th distance, lives in LDS, shared by the workgroup's 16 waves.  It is
//            kept by COUNTING: every code at or below the current bound adds one to a 65-bin distance histogram (one
//            ds_add), and the k-th smallest counted distance is the new bound -- exact for everything the workgroup has
//            seen, 16 x tighter than a wave's own view, which is what keeps the listing path off the stream's critical
//            path.  (Measured and dropped: exchanging the counts between workgroups DURING the stream through one global
//            histogram -- 256 workgroups' atomics and reads on three cache lines serialise: 22 -> 52 us at 12.5 M codes.)
//            A list that holds k + 16 entries drops what the bound excludes; a list whose boundary distance alone
//            overfills it (a corpus of copies) is cut to its exact best k by (d, id) and from then on admits d == d* only
//            below the k-th id: results never depend on the row order.
//   merge    workgroup: the waves hand their entries at or below the bound to the query's selection area, one wave per
//            query rank-sorts the few survivors by (d, id), publishes the best k and adds the workgroup's counts to ONE
//            global histogram per query; the LAST workgroup to arrive (one agent-scope atomic ticket) reads d* -- the
//            exact k-th distance -- straight from that histogram (every code at or below d* was counted: each was
//            evaluated against a bound >= d*), takes the published entries at or below it in ONE pass, sorts them and
//            writes ids, distances, scores and counts.
// Visibility inside the launch (MI355X_MICROARCH.md, "inter-workgroup visibility", last-arriver row): the published
// lists are written with agent-scope (sc1, write-through) stores and the histogram by agent-scope atomics, every storing
// wave drains them (s_waitcnt vmcnt(0)) before the workgroup barrier, ONE lane then adds to the ticket with an
// agent-scope atomic, and the last arriver reads lists and histogram with agent-scope (sc1) loads only.  Ticket and
// histogram are put back to zero by the last workgroup, so the state a launch needs is zeroed ONCE, at allocation.
//
// Semantics as everywhere (DESIGN "Hamming"): d = popcount(q ^ x); order (d asc, record_id asc); <= k hits; unused
// places id 2^64-1, distance 2^32-1, score -1.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace ucfp {

namespace {

constexpr int kDW = 16;                 // waves per workgroup
constexpr int kDCap = 64;               // entries per wave list: one per lane when a list is examined
constexpr uint32_t kDKeep = 40;         // a pruned list may keep up to this many entries (all of d <= d*) before ties are cut
constexpr uint32_t kDSlack = 16;        // a list is pruned (and tau tightened) once it holds k + kDSlack entries
constexpr uint32_t kDTrip = 512;        // codes per wave and trip (8 per lane)
constexpr uint32_t kDSel = kDW * kDKeep; // entries per query a merge selects from in LDS (16 pruned lists always fit)

struct DirView {
    uint64_t* q;       // [8]
    uint64_t* wtid;    // [8][16]  per wave: ids admitted at d == wtau (2^64-1: all)
    uint32_t* tau;     // [8]      workgroup-shared bound on the k-th distance
    uint32_t* wcnt;    // [8][16]  entries in a wave's list
    uint32_t* wtau;    // [8][16]  the wave's own k-th distance (>= tau)
    uint32_t* hist;    // [8][80]
    uint32_t* m;       // [8]
    uint32_t* dstar;   // [8]
    uint32_t* ticket;  // [1]
    uint32_t* pend;    // [8][80]  histogram counts not yet moved to the global histogram
    uint32_t* h0;      // [8][80]  histogram of the first trips' LANE MINIMA (the workgroup's first bound; never the exact one)
    uint64_t* s_id;    // [nq][kDSel]     the entries a merge selects from: id ...
    uint32_t* s_d;     // [nq][kDSel]     ... and distance
    uint32_t* l_d;     // [nq][16][64]    wave lists: distance ...
    uint32_t* l_row;   // [nq][16][64]    ... and row
};
constexpr size_t kDirFixed = 4784 + 2560 + 2560;
constexpr size_t kDirPerQuery = (size_t)kDSel * 12 + (size_t)kDW * kDCap * 8;

__device__ __forceinline__ DirView dir_view(uint8_t* base, uint32_t nq) {
    DirView V;
    V.q = reinterpret_cast<uint64_t*>(base);
    V.wtid = reinterpret_cast<uint64_t*>(base + 64);
    V.tau = reinterpret_cast<uint32_t*>(base + 1088);
    V.wcnt = reinterpret_cast<uint32_t*>(base + 1120);
    V.wtau = reinterpret_cast<uint32_t*>(base + 1632);
    V.hist = reinterpret_cast<uint32_t*>(base + 2144);
    V.m = reinterpret_cast<uint32_t*>(base + 4704);
    V.dstar = reinterpret_cast<uint32_t*>(base + 4736);
    V.ticket = reinterpret_cast<uint32_t*>(base + 4768);
    V.pend = reinterpret_cast<uint32_t*>(base + 4784);
    V.h0 = reinterpret_cast<uint32_t*>(base + 4784 + 2560);
    V.s_id = reinterpret_cast<uint64_t*>(base + kDirFixed);
    V.s_d = reinterpret_cast<uint32_t*>(base + kDirFixed + (size_t)nq * kDSel * 8);
    V.l_d = reinterpret_cast<uint32_t*>(base + kDirFixed + (size_t)nq * kDSel * 12);
    V.l_row = reinterpret_cast<uint32_t*>(base + kDirFixed + (size_t)nq * kDSel * 12 + (size_t)nq * kDW * kDCap * 4);
    return V;
}

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t bcast32(uint32_t v, uint32_t src) {   // src wave-uniform
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src);
}
__device__ __forceinline__ uint64_t bcast64(uint64_t v, uint32_t src) {
    return (uint64_t)bcast32((uint32_t)v, src) | ((uint64_t)bcast32((uint32_t)(v >> 32), src) << 32);
}
__device__ __forceinline__ bool dkey_less(uint32_t d1, uint64_t i1, uint32_t d2, uint64_t i2) {
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

// Wave-synchronous.  The wave's list of query j (cnt >= k entries) shrinks to the entries that can still be among the
// best k: all with d <= tau[j] while they are few, otherwise exactly the wave's best k by (d, id).  Reads and writes the
// wave's LDS words.
__device__ __noinline__ void dir_prune(uint8_t* lds_base, uint32_t nq, const uint64_t* __restrict__ ids, bool asc,
                                       uint32_t j, uint32_t wave, int lane, uint32_t k) {
    const DirView V = dir_view(lds_base, nq);   // (a view passed by value would travel through scratch)
    const uint32_t slot = j * kDW + wave;
    const uint32_t cnt = V.wcnt[slot];
    if (cnt < k) return;
    uint32_t* ld = V.l_d + (size_t)slot * kDCap;
    uint32_t* lr = V.l_row + (size_t)slot * kDCap;
    const uint32_t de = (uint32_t)lane < cnt ? ld[lane] : 0xffffffffu;
    const uint32_t re = (uint32_t)lane < cnt ? lr[lane] : 0u;
    // tau[j] is the exact k-th smallest distance among ALL codes the workgroup has evaluated (distance histogram, see the
    // stream): it is <= this list's own k-th distance, so fewer than k entries at d < tau can be in any one list
    const uint32_t dstar = V.tau[j];
    const uint64_t tie = __ballot(de == dstar);
    const uint32_t c1 = (uint32_t)__popcll(__ballot(de < dstar)), nt = (uint32_t)__popcll(tie);
    bool keep = de <= dstar;
    uint64_t tid = ~0ull;
    if (c1 + nt > kDKeep) {
        // the boundary distance alone overfills the list: keep the (k - c1) smallest ids of it (the one place a list
        // needs record ids), and from now on a candidate at d == d* must beat the largest of those
        const uint32_t need = k - c1;
        const uint64_t ie = de != dstar ? ~0ull : asc ? (uint64_t)re : ids[re];
        uint32_t rank = 0;
        for (uint64_t mm = tie; mm;) {
            const uint32_t o = (uint32_t)__builtin_ctzll(mm);
            mm &= mm - 1;
            const uint64_t oi = bcast64(ie, o);
            rank += (oi < ie || (oi == ie && o < (uint32_t)lane)) ? 1u : 0u;
        }
        keep = de < dstar || (de == dstar && rank < need);
        const uint64_t last = __ballot(de == dstar && rank == need - 1);
        tid = bcast64(ie, (uint32_t)__builtin_ctzll(last));
    }
    const uint64_t km = __ballot(keep);
    const uint32_t pos = lane_rank(km);
    if (keep) {
        ld[pos] = de;
        lr[pos] = re;
    }
    if (lane == 0) {
        V.wcnt[slot] = (uint32_t)__popcll(km);
        V.wtau[slot] = dstar;
        V.wtid[slot] = tid;
    }
}

// inclusive scan over the wave by DPP (row shifts, then the row totals passed on): six VALU instructions, no LDS round trips
__device__ __forceinline__ uint32_t dir_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
// k-th smallest bin of a 65-bin histogram, one bin per lane (64 when fewer than k entries were counted)
__device__ __forceinline__ uint32_t dir_kth_of(uint32_t count_of_lane, uint32_t k) {
    const uint64_t mask = __ballot(dir_scan(count_of_lane) >= k);
    return mask ? (uint32_t)__builtin_ctzll(mask) : 64u;
}
__device__ __forceinline__ uint32_t dir_kth_bin(const uint32_t* hist, uint32_t k, int lane) { return dir_kth_of(hist[lane], k); }

// Wave-synchronous: the best min(m, k) of m entries by (d, id), emitted in order.  get(c) -> entry c.
template <class Get, class Emit>
__device__ __forceinline__ uint32_t dir_select_rounds(Get get, uint32_t m, uint32_t k, int lane, Emit emit) {
    uint32_t ld = 0, emitted = 0;
    uint64_t li = 0;
    bool first = true;
    for (uint32_t r = 0; r < k; r++) {
        uint32_t bd = 0xffffffffu;
        uint64_t bi = ~0ull;
        for (uint32_t c = lane; c < m; c += 64) {
            uint32_t dd;
            uint64_t ii;
            get(c, dd, ii);
            if (dd != 0xffffffffu && (first || dkey_less(ld, li, dd, ii)) && dkey_less(dd, ii, bd, bi)) {
                bd = dd;
                bi = ii;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t od = (uint32_t)__shfl_xor((int)bd, off, 64);
            const uint64_t oi = (uint64_t)__shfl_xor((long long)bi, off, 64);
            if (dkey_less(od, oi, bd, bi)) {
                bd = od;
                bi = oi;
            }
        }
        if (bd == 0xffffffffu) break;
        if (lane == 0) emit(r, bd, bi);
        ld = bd;
        li = bi;
        first = false;
        emitted++;
    }
    return emitted;
}

// the same from LDS arrays; up to 64 entries are rank-sorted in registers (one per lane)
template <class Emit>
__device__ __forceinline__ uint32_t dir_select_lds(const uint32_t* sd, const uint64_t* si, uint32_t m, uint32_t k, int lane,
                                                  Emit emit) {
    if (m <= 64) {
        const uint32_t de = (uint32_t)lane < m ? sd[lane] : 0xffffffffu;
        const uint64_t ie = (uint32_t)lane < m ? si[lane] : ~0ull;
        uint32_t rank = 0;
        for (uint32_t o = 0; o < m; o++) {
            const uint32_t od = bcast32(de, o);
            const uint64_t oi = bcast64(ie, o);
            rank += (od < de || (od == de && (oi < ie || (oi == ie && o < (uint32_t)lane)))) ? 1u : 0u;
        }
        if ((uint32_t)lane < m && rank < k) emit(rank, de, ie);
        return m < k ? m : k;
    }
    return dir_select_rounds([&](uint32_t c, uint32_t& dd, uint64_t& ii) { dd = sd[c]; ii = si[c]; }, m, k, lane, emit);
}

// -DUCFP_DIR_PROF: every workgroup leaves 100 MHz timestamps of its phases behind the published lists (tools/prof_direct.py)
#ifdef UCFP_DIR_PROF
#define DIR_STAMP(slot)                                                                                       \
    do {                                                                                                      \
        if (threadIdx.x == 0) prof[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();     \
    } while (0)
#else
#define DIR_STAMP(slot)
#endif
#define UCFP_AGENT_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define UCFP_AGENT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

}  // namespace

// NQ = 1, 2, 4, 8 >= nq (the query loop of the stream is unrolled: queries in SGPRs, no branch per query).
// wg_d / wg_id: [nq][gridDim.x][k] lists published by the workgroups; counter: one word, ghist: [8][80] words, both
// zero between launches.
template <int NQ>
__global__ __launch_bounds__(kDW * 64) void hamming_direct_kernel(
    const uint64_t* __restrict__ codes, const uint64_t* __restrict__ ids, size_t n, const uint64_t* __restrict__ queries,
    uint32_t nq, uint32_t k, const uint32_t* __restrict__ ids_ascending, uint32_t* wg_d, uint64_t* wg_id, uint32_t* counter,
    uint32_t* ghist, uint32_t* ghist2, uint32_t* tctr, uint64_t* __restrict__ out_ids, uint32_t* __restrict__ out_d, float* __restrict__ out_scores,
    uint32_t* __restrict__ out_cnt, uint64_t* prof) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dir_lds[];
    DIR_STAMP(0);
    const DirView V = dir_view(dir_lds, nq);
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Trips (512 codes) -> waves: round-robin over all waves of the grid.  With 5-8 queries over a full grid (256
    // workgroups) the shares are dynamic instead: the corpus is cut into 32 regions, region r belongs to the 8 workgroups
    // 8 r .. 8 r + 7 (one per XCD: the dispatcher deals workgroups round-robin), whose 128 waves take their first two
    // trips by position and every further one from the region's counter (one agent-scope atomic per trip and wave,
    // requested two trips ahead: its answer arrives with the previous trip's codes).  Workgroups progress at visibly
    // different rates (stream ends 6 us apart at 12.5 M codes with fixed shares) and the counter lets the fast ones take
    // more.  Measured on one box, 100 M codes: 8 queries -- the stream is bound by the popcount work -- 262 -> 240 us;
    // 1 query -- its trips are pure latency and the atomic's, longer than a load's, adds to every one -- 134 -> 142 us,
    // hence the NQ == 8 condition.  Each counter sits on a cache line of its own: sharing one line, the ~200 k atomics
    // of a 100 M-code search serialised at ~5 ns each -- 1.1 ms.
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const size_t ntrips = (n + kDTrip - 1) / kDTrip;
    const uint32_t G0 = gridDim.x;
    const bool dyn = NQ == 8 && tctr != nullptr && gridDim.x == 256u && ntrips >= (size_t)4 * 256 * kDW;
    const size_t region = dyn ? (ntrips + 31) / 32 : ntrips;
    const uint32_t grp = dyn ? blockIdx.x >> 3 : 0u;
    const size_t t_lo = (size_t)grp * region, t_hi = t_lo + region < ntrips ? t_lo + region : ntrips;
    const size_t stride = dyn ? (size_t)8 * kDW : (size_t)gridDim.x * kDW;
    const size_t first = t_lo + (dyn ? (size_t)(blockIdx.x & 7u) * kDW + wave : (size_t)blockIdx.x * kDW + wave);
    auto load_trip = [&](uint64_t (&c)[8], size_t t) {
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const size_t row = t * kDTrip + (size_t)h * 128 + 2 * (size_t)lane;
            if (t < t_hi && row + 1 < n) {
                const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(codes + row));
                c[2 * h] = v[0];
                c[2 * h + 1] = v[1];
            } else {
                c[2 * h] = (t < t_hi && row < n) ? codes[row] : 0ull;
                c[2 * h + 1] = 0ull;
            }
        }
    };
    auto grab = [&]() -> uint32_t {          // lane 0's answer counts
        return lane == 0 ? __hip_atomic_fetch_add(tctr + grp * 32u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    };
    uint64_t cn[8];
    load_trip(cn, first);                     // the first trip's loads leave before anything else happens
    uint32_t ga = (dyn && first < t_hi) ? grab() : 0u;
    DIR_STAMP(1);
    if (threadIdx.x < nq) {
        V.q[threadIdx.x] = queries[threadIdx.x];
        V.tau[threadIdx.x] = 64u;
        V.m[threadIdx.x] = 0u;
    }
    for (uint32_t i = threadIdx.x; i < nq * 80; i += kDW * 64) {
        V.hist[i] = 0u;
        V.pend[i] = 0u;
        V.h0[i] = 0u;
    }
    if (threadIdx.x < nq * kDW) {
        V.wcnt[threadIdx.x] = 0u;
        V.wtau[threadIdx.x] = 64u;
        V.wtid[threadIdx.x] = ~0ull;
    }
    uint32_t qlo[NQ], qhi[NQ];          // wave-uniform: scalar loads
#pragma unroll
    for (int j = 0; j < NQ; j++) {
        const uint64_t qv = queries[(uint32_t)j < nq ? j : 0];
        qlo[j] = (uint32_t)qv;
        qhi[j] = (uint32_t)(qv >> 32);
    }
    const uint32_t prune_at = k + kDSlack < (uint32_t)kDCap ? k + kDSlack : (uint32_t)kDCap;
    // Ties are broken by record id.  Where the shard's ids ascend with the row number (an append-only shard keeps that
    // fact on the device, launch_ids_order_update) the ROW is the same order: the lists, both merges and the published
    // entries then carry rows as their keys and the only ids ever fetched are those of the k answers -- otherwise every
    // workgroup's merge waits for a gather (~3 us behind the stream).
    const bool asc = ids_ascending != nullptr && *ids_ascending != 0u;
    __syncthreads();

    // tie-break keys of the wave's listed rows of query j, one per lane
    auto gather_keys = [&](int j) -> uint64_t {
        const uint32_t slot = (uint32_t)((uint32_t)j < nq ? j : 0) * kDW + wave;
        const bool live = (uint32_t)lane < V.wcnt[slot];
        const uint32_t row = live ? V.l_row[(size_t)slot * kDCap + lane] : 0u;
        if (asc) return live ? (uint64_t)row : ~0ull;
        const uint64_t g = ids[row];
        return live ? g : ~0ull;
    };
    // the workgroup's counts -> the global histogram (lane = bin; bin 64 never bounds anything)
    auto flush_counts = [&](int j) {
        const uint32_t pv = atomicExch(&V.pend[j * 80 + lane], 0u);
        if (pv) __hip_atomic_fetch_add(ghist + j * 80 + lane, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- the first bound.  The bound starts at 64, where every code is a candidate -- 512 of them per wave and query through a
    // listing path built for the rare one (PMC, round 3's kernel, 8 queries over 12.5 M codes = six trips per wave: 1250
    // vector instructions per trip on average against ~350 for the distances).  So, before anything is listed: every wave takes
    // the minimum distance each LANE sees in its first trip (64 different codes per wave and query), the workgroup histograms
    // its up to 1024 minima, and the k-th smallest of them bounds the k-th distance (any k of those codes witness it).  The same
    // histogram goes to a second global histogram (ghist2: never the exact one -- these codes are counted again when the stream
    // evaluates them), which every workgroup reads back at its second trip: the k-th smallest of ~260 k minima from all over
    // the corpus is within a bit or two of the final k-th distance, where a workgroup's own bound settles ~6 bits above it (a
    // candidate in every other trip and query).  Nobody waits for anybody: what has been published by then bounds just as well.
    {
        const bool whole = first < t_hi && first * kDTrip + kDTrip <= n;      // (codes past the end read as 0: no witnesses)
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            if ((uint32_t)j < nq && whole) {
                uint32_t best = 64;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t d = (uint32_t)__builtin_popcount((uint32_t)cn[u] ^ qlo[j]) +
                                       (uint32_t)__builtin_popcount((uint32_t)(cn[u] >> 32) ^ qhi[j]);
                    best = d < best ? d : best;
                }
                atomicAdd(&V.h0[j * 80 + best], 1u);
            }
        }
        __syncthreads();
        if (wave < nq) {
            const uint32_t mine = V.h0[wave * 80 + lane];
            const uint32_t t0 = dir_kth_of(mine, k);
            if (lane == 0) V.tau[wave] = t0;                                  // (64 when fewer than k minima were seen)
            if (mine && G0 > 1) __hip_atomic_fetch_add(ghist2 + wave * 80 + lane, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }

    // ---- the stream
    uint32_t it = 0;
    size_t tn = first + stride;
    for (size_t t = first; t < t_hi;) {
        uint64_t c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = cn[u];
        uint32_t gcount = 0;
        const bool read_now = G0 > 1 && wave < nq && it == 1;
        if (read_now) gcount = UCFP_AGENT_LOAD(ghist2 + wave * 80 + lane);     // consumed after this trip's distances
        // the trip after next: by position, or what the region's counter answered (the answer came in with c[])
        const size_t t3 = dyn ? t_lo + 2 * stride + (uint32_t)__builtin_amdgcn_readfirstlane((int)ga) : tn + stride;
        load_trip(cn, tn);
        if (dyn && tn < t_hi) ga = grab();
        const size_t base = t * kDTrip + 2 * (size_t)lane;   // row of c[2 h + e] = base + 128 h + e
        uint32_t tj[NQ];
#pragma unroll
        for (int j = 0; j < NQ; j++) tj[j] = V.tau[j];
        uint32_t hits = 0;
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            if ((uint32_t)j < nq) {
                uint32_t best = 64;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t d = (uint32_t)__builtin_popcount((uint32_t)c[u] ^ qlo[j]) +
                                       (uint32_t)__builtin_popcount((uint32_t)(c[u] >> 32) ^ qhi[j]);
                    best = d < best ? d : best;
                }
                if (__any(best <= tj[j])) hits |= 1u << j;
            }
        }
        if (read_now) {
            const uint32_t tg = dir_kth_of(gcount, k);
            if (lane == 0) atomicMin(&V.tau[wave], tg);
        }
        // ---- rare: some code of this trip may enter the list of query j
        while (hits) {
            const uint32_t j = (uint32_t)__builtin_ctz(hits);
            hits &= hits - 1;
            const uint64_t qv = V.q[j];
            const uint32_t ql = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)qv);
            const uint32_t qh = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(qv >> 32));
            const uint32_t slot = j * kDW + wave;
            uint32_t* ld = V.l_d + (size_t)slot * kDCap;
            uint32_t* lr = V.l_row + (size_t)slot * kDCap;
            uint32_t cnt = V.wcnt[slot], wt = V.wtau[slot];
            uint64_t wi = V.wtid[slot];
            uint32_t* wh = V.hist + j * 80;      // workgroup histogram of every distance <= tau evaluated so far
            uint32_t* wp = V.pend + j * 80;      // ... and the part of it the global histogram has not seen yet
            const uint32_t tnow = V.tau[j];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const size_t row = base + (size_t)(u >> 1) * 128 + (size_t)(u & 1);
                const uint32_t d = (uint32_t)__builtin_popcount((uint32_t)c[u] ^ ql) +
                                   (uint32_t)__builtin_popcount((uint32_t)(c[u] >> 32) ^ qh);
                bool cand = row < n && d <= tnow;
                if (!__any(cand)) continue;
                if (cand) {                               // counted once, whatever happens to it below
                    atomicAdd(&wh[d], 1u);
                    atomicAdd(&wp[d], 1u);
                }
                for (;;) {
                    if (wi != ~0ull) {
                        // this wave's list is full of the boundary distance (a corpus of copies): there the id decides
                        const bool at = cand && d == wt;
                        if (__any(at)) {
                            const uint64_t id = !at ? 0ull : asc ? (uint64_t)row : ids[row];
                            cand = cand && (!at || id < wi);
                        }
                    }
                    const uint64_t mask = __ballot(cand);
                    if (!mask) break;
                    if (cnt == (uint32_t)kDCap) {
                        // full: refresh tau from the histogram, then drop what it excludes
                        wave_lds_sync();
                        const uint32_t tn = dir_kth_bin(wh, k, lane);
                        if (lane == 0) {
                            atomicMin(&V.tau[j], tn);
                            V.wcnt[slot] = cnt;
                        }
                        wave_lds_sync();
                        dir_prune(dir_lds, nq, ids, asc, j, wave, lane, k);
                        wave_lds_sync();
                        cnt = V.wcnt[slot];
                        wt = V.wtau[slot];
                        wi = V.wtid[slot];
                        cand = cand && d <= V.tau[j];
                        continue;
                    }
                    const uint32_t room = (uint32_t)kDCap - cnt, r = lane_rank(mask);
                    const bool take = cand && r < room;
                    if (take) {
                        ld[cnt + r] = d;
                        lr[cnt + r] = (uint32_t)row;
                    }
                    const uint32_t tot = (uint32_t)__popcll(mask);
                    cnt += tot < room ? tot : room;
                    cand = cand && !take;
                    if (tot <= room) break;
                }
            }
            // the k-th smallest distance of everything the workgroup's 16 waves have evaluated: every code at or below the
            // current bound was counted (the bound only falls), so this is exact for the workgroup
            wave_lds_sync();
            const uint32_t tn = dir_kth_bin(wh, k, lane);
            if (lane == 0) {
                atomicMin(&V.tau[j], tn);
                V.wcnt[slot] = cnt;
            }
            wave_lds_sync();
            if (cnt >= prune_at) {
                dir_prune(dir_lds, nq, ids, asc, j, wave, lane, k);
                wave_lds_sync();
            }
        }
        it++;
        t = tn;
        tn = t3;
    }
    // every list down to what the bound admits (<= kDKeep entries: 16 of them fit the merge area), then the keys.
    // (Measured and dropped: fetching the ids when a row is listed, or under the last trip -- the gather sits in the same
    // in-order queue as the stream's loads, so the wave waits for it at its next trip instead: no gain at 12.5 M codes.)
    uint64_t gid[NQ];
#pragma unroll
    for (int j = 0; j < NQ; j++) {
        gid[j] = ~0ull;
        if ((uint32_t)j < nq) {
            if (V.wcnt[(uint32_t)j * kDW + wave] > kDKeep) {
                dir_prune(dir_lds, nq, ids, asc, (uint32_t)j, wave, lane, k);
                wave_lds_sync();
            }
            gid[j] = gather_keys(j);
        }
    }
    __syncthreads();
    DIR_STAMP(2);

    // ---- workgroup merge: every wave hands its entries at or below the bound to the query's selection area ...
#pragma unroll
    for (int j = 0; j < NQ; j++)
        if ((uint32_t)j < nq) {
            const uint32_t slot = (uint32_t)j * kDW + wave;
            const uint32_t de = (uint32_t)lane < V.wcnt[slot] ? V.l_d[(size_t)slot * kDCap + lane] : 0xffffffffu;
            const bool keep = de <= V.tau[j];
            const uint64_t km = __ballot(keep);
            if (km) {
                uint32_t pos = 0;
                if (lane == 0) pos = atomicAdd(&V.m[j], (uint32_t)__popcll(km));
                pos = bcast32(pos, 0) + lane_rank(km);
                if (keep && pos < kDSel) {
                    V.s_d[(size_t)j * kDSel + pos] = de;
                    V.s_id[(size_t)j * kDSel + pos] = gid[j];
                }
            }
        }
    __syncthreads();
    // ... and wave j sorts query j's few survivors by (d, id), publishes the best k, and moves the last counts out
    const uint32_t G = gridDim.x;
    if (wave < nq) {
        const uint32_t j = wave;
        const uint32_t m = V.m[j] < kDSel ? V.m[j] : kDSel;      // (16 lists of <= kDKeep entries: never cut)
        uint32_t* pd = wg_d + ((size_t)j * G + blockIdx.x) * k;
        uint64_t* pi = wg_id + ((size_t)j * G + blockIdx.x) * k;
        const uint32_t got = dir_select_lds(V.s_d + (size_t)j * kDSel, V.s_id + (size_t)j * kDSel, m, k, lane,
                                            [&](uint32_t r, uint32_t dd, uint64_t ii) {
                                                UCFP_AGENT_STORE(pd + r, dd);
                                                UCFP_AGENT_STORE(pi + r, ii);
                                            });
        for (uint32_t r = got + lane; r < k; r += 64) {
            UCFP_AGENT_STORE(pd + r, 0xffffffffu);
            UCFP_AGENT_STORE(pi + r, ~0ull);
        }
        flush_counts((int)j);
    }
    DIR_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its published entries and counts
    __syncthreads();
    DIR_STAMP(4);
    if (threadIdx.x == 0)
        *V.ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    DIR_STAMP(5);
    if (*V.ticket != G - 1) return;

    // ---- the last workgroup to arrive.  The global histogram now holds every workgroup's counts; every code at or below
    // the final k-th distance d* was counted (each was evaluated against a bound >= d*), so d* is simply its k-th bin.
    // The published entries at or below d* are taken in ONE pass: batches of 8 independent loads per thread.
    const uint32_t E = G * k, total = nq * E;
    constexpr uint32_t kT = kDW * 64, kB = 20;      // 20 480 published distances (8 queries x 256 workgroups x k = 10) are ONE round trip
    if (threadIdx.x < nq) V.m[threadIdx.x] = 0;
    // the first batch of published distances is requested BEFORE d* is known (it does not depend on it): the histogram's
    // round trip and the lists' are one.  Loads past the published entries are skipped a whole workgroup at a time (uniform).
    uint32_t v0[kB];
#pragma unroll
    for (uint32_t u = 0; u < kB; u++) {
        v0[u] = 0xffffffffu;
        if (u * kT < total) {
            const uint32_t x = u * kT + threadIdx.x;
            v0[u] = UCFP_AGENT_LOAD(wg_d + (x < total ? x : 0u));
        }
    }
    if (wave < nq) {
        const uint32_t cnt_l = UCFP_AGENT_LOAD(ghist + wave * 80 + lane);
        UCFP_AGENT_STORE(ghist + wave * 80 + lane, 0u);          // zero again for the next launch (bin 64 is never written)
        UCFP_AGENT_STORE(ghist2 + wave * 80 + lane, 0u);
        const uint32_t ds = dir_kth_of(cnt_l, k);
        if (lane == 0) V.dstar[wave] = ds;
    }
    __syncthreads();
    for (uint32_t b0 = 0; b0 < total; b0 += kT * kB) {
        uint32_t v[kB], pos[kB];
        uint64_t w[kB];
#pragma unroll
        for (uint32_t u = 0; u < kB; u++) {      // unconditional within a batch row, so that the loads are ONE round trip
            const uint32_t x = b0 + u * kT + threadIdx.x;
            v[u] = b0 == 0 ? v0[u] : (b0 + u * kT < total ? UCFP_AGENT_LOAD(wg_d + (x < total ? x : 0u)) : 0xffffffffu);
        }
#pragma unroll
        for (uint32_t u = 0; u < kB; u++) {
            const uint32_t x = b0 + u * kT + threadIdx.x, j = x < total ? x / E : 0;
            pos[u] = (x < total && v[u] <= V.dstar[j]) ? atomicAdd(&V.m[j], 1u) : 0xffffffffu;
        }
#pragma unroll
        for (uint32_t u = 0; u < kB; u++) {      // the few entries taken: their ids, again without anything in between
            const uint32_t x = b0 + u * kT + threadIdx.x;
            w[u] = pos[u] < kDSel ? UCFP_AGENT_LOAD(wg_id + x) : 0ull;
        }
#pragma unroll
        for (uint32_t u = 0; u < kB; u++) {
            const uint32_t x = b0 + u * kT + threadIdx.x, j = x < total ? x / E : 0;
            if (pos[u] < kDSel) {
                V.s_d[(size_t)j * kDSel + pos[u]] = v[u];
                V.s_id[(size_t)j * kDSel + pos[u]] = w[u];
            }
        }
    }
    __syncthreads();
    if (wave < nq) {
        const uint32_t j = wave, m = V.m[j];
        auto emit = [&](uint32_t r, uint32_t dd, uint64_t ii) {
            if (asc) ii = ids[(uint32_t)ii];          // keys were rows: the k answers are the only ids ever read
            out_ids[(size_t)j * k + r] = ii;
            out_d[(size_t)j * k + r] = dd;
            if (out_scores) out_scores[(size_t)j * k + r] = 1.0f - (float)dd * (1.0f / 64.0f);
        };
        uint32_t got;
        if (m <= kDSel) {
            got = dir_select_lds(V.s_d + (size_t)j * kDSel, V.s_id + (size_t)j * kDSel, m, k, lane, emit);
        } else {
            // more ties at the k-th distance than LDS holds (published lists of copies): select straight from the lists
            got = dir_select_rounds(
                [&](uint32_t c, uint32_t& dd, uint64_t& ii) {
                    dd = UCFP_AGENT_LOAD(wg_d + (size_t)j * E + c);
                    ii = dd <= 64u ? UCFP_AGENT_LOAD(wg_id + (size_t)j * E + c) : ~0ull;
                },
                E, k, lane, emit);
        }
        for (uint32_t r = got + lane; r < k; r += 64) {
            out_ids[(size_t)j * k + r] = ~0ull;
            out_d[(size_t)j * k + r] = 0xffffffffu;
            if (out_scores) out_scores[(size_t)j * k + r] = -1.0f;
        }
        if (lane == 0) out_cnt[j] = got;
    }
    __syncthreads();
    DIR_STAMP(6);
    if (threadIdx.x == 0) UCFP_AGENT_STORE(counter, 0u);   // every workgroup has arrived: the next launch starts from zero
    if (tctr && threadIdx.x < 32) UCFP_AGENT_STORE(tctr + threadIdx.x * 32u, 0u);
}

bool hamming_direct_ok(size_t n, uint32_t nq, uint32_t k) {
    return n >= 1 && n <= 0xffffffffull && nq >= 1 && nq <= kHammingDirectMaxQ && k >= 1 && k <= kHammingDirectMaxK;
}

// state: kHammingDirectZeroBytes (word 0 = the ticket, from byte 256 the global histogram; zeroed ONCE when the buffer is
// made, the kernel leaves them zero) followed by the published lists
size_t hamming_direct_state_bytes() {
    return kHammingDirectZeroBytes + (size_t)kHammingDirectMaxQ * 256 * kHammingDirectMaxK * 12 +
           256 * 8 * 8 /* UCFP_DIR_PROF stamps */;
}

int launch_hamming_direct(const uint64_t* codes, const uint64_t* ids, size_t n, const uint64_t* queries, uint32_t nq,
                          uint32_t k, uint8_t* state, uint64_t* out_ids, uint32_t* out_dist, float* out_scores,
                          uint32_t* out_cnt, hipStream_t stream, const uint32_t* ids_ascending) {
    if (!hamming_direct_ok(n, nq, k)) return -1;
    static_assert(kHammingDirectMaxK <= kDKeep && kDKeep < (uint32_t)kDCap, "a pruned list (<= max(k, kDKeep) entries) leaves room");
    const size_t ntrips = (n + kDTrip - 1) / kDTrip;
    unsigned G = (unsigned)((ntrips + kDW - 1) / kDW);
    if (G > 256) G = 256;
    if (G < 1) G = 1;
    const size_t lds = kDirFixed + (size_t)nq * kDirPerQuery;
    static_assert(kHammingDirectZeroBytes >= 8192 + kHammingDirectMaxQ * 80 * 4 && 4096 >= 256 + kHammingDirectMaxQ * 80 * 4,
                  "ticket + histogram + 32 trip counters, each on a cache line of its own, + the histogram of first-trip minima");
    uint32_t* counter = reinterpret_cast<uint32_t*>(state);
    uint32_t* ghist = reinterpret_cast<uint32_t*>(state + 256);
    uint32_t* ghist2 = reinterpret_cast<uint32_t*>(state + 8192);
    static const bool fixed_shares = getenv("UCFP_DIRECT_STATIC") != nullptr;      // A/B switch of the trip counters
    uint32_t* tctr = fixed_shares ? nullptr : reinterpret_cast<uint32_t*>(state + 4096);
    uint8_t* lists = state + kHammingDirectZeroBytes;
    uint64_t* wg_id = reinterpret_cast<uint64_t*>(lists);
    uint32_t* wg_d = reinterpret_cast<uint32_t*>(lists + (size_t)kHammingDirectMaxQ * 256 * kHammingDirectMaxK * 8);
    auto go = [&](auto kernel, int nqt) {
        static bool attr_set[9] = {};
        if (!attr_set[nqt]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(kDirFixed + (size_t)nqt * kDirPerQuery));
            attr_set[nqt] = true;
        }
        hipLaunchKernelGGL(kernel, dim3(G), dim3(kDW * 64), lds, stream, codes, ids, n, queries, nq, k, ids_ascending, wg_d,
                           wg_id, counter, ghist, ghist2, tctr, out_ids, out_dist, out_scores, out_cnt,
                           reinterpret_cast<uint64_t*>(lists + (size_t)kHammingDirectMaxQ * 256 * kHammingDirectMaxK * 12));
    };
    if (nq == 1) go(hamming_direct_kernel<1>, 1);
    else if (nq == 2) go(hamming_direct_kernel<2>, 2);
    else if (nq <= 4) go(hamming_direct_kernel<4>, 4);
    else go(hamming_direct_kernel<8>, 8);
    return 0;
}

}  // namespace ucfp
