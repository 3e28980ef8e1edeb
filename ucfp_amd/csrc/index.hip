// index.hip -- GPU-resident shard of the kNN index and its C ABI (ucfp_index_*).
//
// Mirrors `trait IndexBackend` (src/index/mod.rs:17-78) for the kNN part: upsert / delete / knn /
// flush, scoped by tenant.  Layout in HBM: one contiguous range per tenant (the analogue of the
// reference's `(tenant_id, 0)..=(tenant_id, u64::MAX)` key range, src/index/embedded/mod.rs:300-302):
//   ids   u64[n]            record ids, row-aligned with
//   rows  u64[n]            (HAMMING64)   or   f32[n][dim] + norms f32[n]   (COSINE_F32)
// Capacity doubles on growth; delete swaps the last row into the hole, so a tenant scan is
// always one dense stream.  The id -> row map lives on the host (the caller of this ABI is a host
// thread of the ingest route); APPEND_ONLY indexes skip it.

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "../../include/ucfp_hip.h"
#include "common.h"

namespace ucfp {
int capi_fail(int code, const char* fmt, ...);  // capi.hip
int ctx_device(const ucfp_ctx* ctx);            // capi.hip
}  // namespace ucfp

using ucfp::capi_fail;

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return capi_fail(UCFP_E_INDEX, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

namespace {

struct DevBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
    int ensure(size_t need) {
        if (cap >= need) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = need + need / 4 + 4096;
        HIP_TRY(hipMalloc((void**)&p, want));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Shard {
    uint64_t* ids = nullptr;
    uint8_t* rows = nullptr;
    float* norms = nullptr;
    size_t n = 0, cap = 0;
    std::unordered_map<uint64_t, size_t> pos;  // id -> row (unless APPEND_ONLY)
    std::vector<uint64_t> host_ids;            // row -> id (unless APPEND_ONLY)
    uint32_t* order = nullptr;                 // APPEND_ONLY Hamming shards: [0] = 1 while ids ascend with the row, [2..3] = last id
};

__global__ void scatter_rows_kernel(const uint64_t* __restrict__ src_ids, const uint8_t* __restrict__ src_rows,
                                    const uint64_t* __restrict__ dst_row, size_t n, size_t row_bytes,
                                    uint64_t* __restrict__ ids, uint8_t* __restrict__ rows) {
    // one block per staged item; threads copy the row in 4-byte words
    const size_t i = blockIdx.x;
    if (i >= n) return;
    const size_t r = dst_row[i];
    if (threadIdx.x == 0) ids[r] = src_ids[i];
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src_rows + i * row_bytes);
    uint32_t* d = reinterpret_cast<uint32_t*>(rows + r * row_bytes);
    for (size_t w = threadIdx.x; w < row_bytes / 4; w += blockDim.x) d[w] = s[w];
}

__global__ void gather_norms_kernel(const float* __restrict__ src, const uint64_t* __restrict__ dst_row, size_t n,
                                    float* __restrict__ norms) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) norms[dst_row[i]] = src[i];
}

// APPEND_ONLY Hamming shards keep, on the device, whether their ids ascend with the row number (ucfp::launch_ids_order_update).
int ensure_order_state(Shard& s, hipStream_t st) {
    if (s.order) return 0;
    if (hipMalloc((void**)&s.order, 16) != hipSuccess) return -1;
    static const uint32_t init[4] = {1u, 0u, 0u, 0u};
    if (hipMemcpyAsync(s.order, init, 16, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    return 0;
}

}  // namespace

struct ucfp_index {
    ucfp_ctx* ctx = nullptr;
    int device = 0;
    int kind = 0;
    uint32_t dim = 0;
    uint32_t flags = 0;
    size_t row_bytes = 0;
    std::mutex mu;
    std::map<uint32_t, Shard> shards;
    hipStream_t stream = nullptr;  // host-pointer entry points run here
    // search workspaces (partials, keys, ...): Hamming searches alternate between two, so that two batches enqueued on
    // different streams run side by side (the sharded search does that); a cosine search -- its key matrix can be
    // gigabytes -- always takes slot 0.  ws_done[i] orders successive uses of slot i across streams.
    DevBuf ws_slot[2];
    DevBuf direct_state[2];        // per slot: ticket word + published lists of the single-launch search (hamming_direct.hip)
    hipEvent_t ws_done[2] = {nullptr, nullptr};
    unsigned ws_next = 0, ws_cur = 0;
    // Mutations and searches may run on different streams: a mutation first waits for every search enqueued so far
    // (ws_done: none may still be reading rows that are overwritten, swapped or re-allocated) and leaves `data_ready`
    // behind; a search waits for `data_ready` (rows appended on another stream have landed).
    hipEvent_t data_ready = nullptr;
    DevBuf stage;                  // host<->device staging
};

namespace ucfp {
int index_kind(const ucfp_index* ix) { return ix->kind; }
int index_device(const ucfp_index* ix) { return ix->device; }
uint32_t index_dim(const ucfp_index* ix) { return ix->dim; }
}  // namespace ucfp

namespace {

int shard_reserve(ucfp_index* ix, Shard& s, size_t need, hipStream_t st) {
    if (need <= s.cap) return 0;
    size_t ncap = s.cap ? s.cap * 2 : 1024;
    while (ncap < need) ncap *= 2;
    uint64_t* nids = nullptr;
    uint8_t* nrows = nullptr;
    float* nnorms = nullptr;
    HIP_TRY(hipMalloc((void**)&nids, ncap * 8));
    HIP_TRY(hipMalloc((void**)&nrows, ncap * ix->row_bytes));
    if (ix->kind == UCFP_INDEX_COSINE_F32) HIP_TRY(hipMalloc((void**)&nnorms, ncap * 4));
    if (s.n) {
        HIP_TRY(hipMemcpyAsync(nids, s.ids, s.n * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(nrows, s.rows, s.n * ix->row_bytes, hipMemcpyDeviceToDevice, st));
        if (nnorms) HIP_TRY(hipMemcpyAsync(nnorms, s.norms, s.n * 4, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (s.ids) (void)hipFree(s.ids);
    if (s.rows) (void)hipFree(s.rows);
    if (s.norms) (void)hipFree(s.norms);
    s.ids = nids;
    s.rows = nrows;
    s.norms = nnorms;
    s.cap = ncap;
    return 0;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Device-side search over one shard; everything enqueued on `st`. Outputs are device pointers.
// d_keys_out receives the sort key (Hamming distance / inverted score image); may be the
// caller's d_out_dist or an internal buffer.
int search_shard_dev(ucfp_index* ix, const Shard* s, const void* d_queries, size_t nq, uint32_t k,
                     uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_keys, uint32_t* d_out_cnt,
                     hipStream_t st) {
    const size_t n = s ? s->n : 0;
    if (ix->kind == UCFP_INDEX_HAMMING64 && nq <= ucfp::kHammingDirectMaxQ && ucfp::hamming_direct_ok(n, (uint32_t)nq, k) &&
        !getenv("UCFP_HAMMING_NO_DIRECT")) {
        // the request shape of /v1/query (one query; up to 8): ONE launch, no sample, no stages
        DevBuf& stt = ix->direct_state[ix->ws_cur];
        if (!stt.p) {
            int rc = stt.ensure(ucfp::hamming_direct_state_bytes());
            if (rc) return rc;
            HIP_TRY(hipMemsetAsync(stt.p, 0, ucfp::kHammingDirectZeroBytes, st));   // ticket + global histogram: zeroed once, the kernel leaves them zero
        }
        uint32_t* keys = d_out_keys;
        if (!keys) {
            int rc = ix->ws_slot[ix->ws_cur].ensure(nq * k * 4 + 256);
            if (rc) return rc;
            keys = reinterpret_cast<uint32_t*>(ix->ws_slot[ix->ws_cur].p);
        }
        ucfp::launch_hamming_direct(reinterpret_cast<const uint64_t*>(s->rows), s->ids, n,
                                    reinterpret_cast<const uint64_t*>(d_queries), (uint32_t)nq, k, stt.p, d_out_ids, keys,
                                    d_out_scores, d_out_cnt, st, s->order);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (ix->kind == UCFP_INDEX_HAMMING64) {
        // query chunks of kHammingMaxBatch reuse one workspace (candidate lists are nq x cand_cap), stream-ordered
        const size_t chunk = nq < ucfp::kHammingMaxBatch ? nq : ucfp::kHammingMaxBatch;
        ucfp::HammingPlan p = ucfp::hamming_plan(n, (uint32_t)chunk, k);
        size_t need = ucfp::hamming_workspace_bytes(p, (uint32_t)chunk, k) + 4096;
        if (nq % chunk) {   // the tail chunk has its own plan
            const uint32_t tail = (uint32_t)(nq % chunk);
            const size_t nt = ucfp::hamming_workspace_bytes(ucfp::hamming_plan(n, tail, k), tail, k) + 4096;
            need = nt > need ? nt : need;
        }
        uint32_t* keys = d_out_keys;
        size_t keys_off = 0;
        if (!keys) {
            keys_off = align256(need);
            need = keys_off + nq * k * 4;
        }
        int rc = ix->ws_slot[ix->ws_cur].ensure(need);
        if (rc) return rc;
        if (!keys) keys = reinterpret_cast<uint32_t*>(ix->ws_slot[ix->ws_cur].p + keys_off);
        const uint64_t* dq = reinterpret_cast<const uint64_t*>(d_queries);
        for (size_t c0 = 0; c0 < nq; c0 += chunk) {
            const uint32_t cn = (uint32_t)(nq - c0 < chunk ? nq - c0 : chunk);
            if (cn != chunk) p = ucfp::hamming_plan(n, cn, k);
            ucfp::launch_hamming_search(s ? reinterpret_cast<const uint64_t*>(s->rows) : nullptr, s ? s->ids : nullptr,
                                        n, dq + c0, cn, k, ix->ws_slot[ix->ws_cur].p, p, d_out_ids + c0 * k, keys + c0 * k,
                                        d_out_scores ? d_out_scores + c0 * k : nullptr, d_out_cnt + c0, st,
                                        s ? s->order : nullptr);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // ---- cosine ----
    const uint32_t dim = ix->dim;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(d_out_ids, 0xff, nq * k * 8, st));
        if (d_out_keys) HIP_TRY(hipMemsetAsync(d_out_keys, 0xff, nq * k * 4, st));
        HIP_TRY(hipMemsetAsync(d_out_cnt, 0, nq * 4, st));
        if (d_out_scores) {
            int rc = ix->ws_slot[ix->ws_cur].ensure(nq * k * 4 + 256);
            if (rc) return rc;
            HIP_TRY(hipMemsetAsync(ix->ws_slot[ix->ws_cur].p, 0xff, nq * k * 4, st));
            ucfp::launch_cosine_scores_from_keys(reinterpret_cast<uint32_t*>(ix->ws_slot[ix->ws_cur].p), nq * k, d_out_scores, st);
        }
        return 0;
    }
    const int qpp = ucfp::cosine_queries_per_pass(dim, nq);
    if (qpp < 1) return capi_fail(UCFP_E_UNSUPPORTED, "cosine dim %u does not fit one query row in LDS", dim);
    // chunk the query batch so the key matrix stays under ~2 GiB
    size_t chunk = (size_t)2048 * 1024 * 1024 / (4 * n);
    if (chunk < (size_t)qpp) chunk = qpp;
    if (chunk > 32768) chunk = 32768;  // queries ride on gridDim.y of the select kernel
    if (chunk > nq) chunk = nq;
    ucfp::SelectPlan sp = ucfp::select_plan(n, (uint32_t)chunk);
    size_t off = 0;
    const size_t o_qn = off;
    off = align256(off + nq * 4);
    const size_t o_keys = off;
    off = align256(off + chunk * n * 4);
    const size_t o_pid = off;
    off = align256(off + (size_t)sp.slices * chunk * k * 8);
    const size_t o_pk = off;
    off = align256(off + (size_t)sp.slices * chunk * k * 4);
    const size_t o_pc = off;
    off = align256(off + (size_t)sp.slices * chunk * 4);
    const size_t o_ok = off;
    off = align256(off + nq * k * 4);
    const size_t tmp_e = 2 * ucfp::topk_merge_tmp_entries(sp.slices, (uint32_t)chunk, k);   // both tree levels
    const size_t o_tid = off;
    off = align256(off + tmp_e * 8);
    const size_t o_tk = off;
    off = align256(off + tmp_e * 4);
    // filtered batch passes (see below): sample answer, thresholds, candidate lists, overflow flag
    const uint32_t cap = ucfp::kCosineListCap;
    const size_t fq = (size_t)qpp < chunk ? (size_t)qpp : chunk;
    const size_t o_bid = off;
    off = align256(off + fq * k * 8);
    const size_t o_bk = off;
    off = align256(off + fq * k * 4);
    const size_t o_tau = off;
    off = align256(off + fq * 4);
    const size_t o_uq = off;
    off = align256(off + fq * 4);
    const size_t o_cc = off;
    off = align256(off + fq * 4);
    const size_t o_ck = off;
    off = align256(off + fq * cap * 4);
    const size_t o_cr = off;
    off = align256(off + fq * cap * 4);
    const size_t o_flag = off;
    off = align256(off + 256);
    const size_t o_mins = off;
    off = align256(off + ucfp::select_pruned_ws_bytes(n, 16));
    // the pass without a key matrix (5 .. 48 queries): chunk minima, thresholds, listed chunks, their keys
    // (sized for either row-stream kernel: chunks of 16 rows, 48 padded queries)
    // (only when the call can take that pass: one pass of <= 48 queries; 12 bytes per row)
    // round 4: the minima of 5 .. 64 queries come from the f16 matrix pipe where the shape allows (cosine_mins_f16: within
    // cosine_mins_eps of the exact score, the thresholds widened by as much), else from the f32 tile (<= 48 queries)
    // (larger batches: passes of at most 64 queries, each its own chain)
    const bool prune_f16 = !getenv("UCFP_COSINE_NO_F16") &&
                           ucfp::cosine_mins_f16_ok(reinterpret_cast<const float*>(s->rows), dim,
                                                    reinterpret_cast<const float*>(d_queries), (uint32_t)(nq < 64 ? nq : 64), n, k);
    const bool may_prune = prune_f16 || (nq <= 48 && nq <= (size_t)qpp &&
                                         ucfp::cosine_prune_ok(reinterpret_cast<const float*>(s->rows), dim,
                                                               reinterpret_cast<const float*>(d_queries), (uint32_t)nq, n, k));
    const uint32_t p_capq = ucfp::cosine_prune_plan(n, (uint32_t)fq, k, true).capq;
    const size_t o_pmin = off;
    off = align256(off + (may_prune ? (n / 16 + 1) * 64 * 4 + 64 : 0));
    const size_t o_pwmin = off;
    off = align256(off + (may_prune ? (size_t)64 * 256 * 4 * 8 * 4 : 0));   // [qpad <= 64][waves <= 8192]
    const size_t o_qimg = off;
    off = align256(off + (prune_f16 ? (size_t)64 * dim * 2 : 0));
    const size_t o_ptws = off;
    off = align256(off + ucfp::prune_tau_ws_bytes((uint32_t)fq));
    const size_t o_ptau = off;
    off = align256(off + fq * 4);
    const size_t o_plist = off;
    off = align256(off + fq * p_capq * 8);
    const size_t o_prange = off;
    off = align256(off + fq * 8);
    const size_t o_pkeys = off;
    off = align256(off + fq * p_capq * 16 * 4);
    const size_t o_pflag = off;   // [0] fallback flag, [1] listed chunks
    off = align256(off + 256);
    int rc = ix->ws_slot[ix->ws_cur].ensure(off);
    if (rc) return rc;
    uint8_t* w = ix->ws_slot[ix->ws_cur].p;
    float* qn = reinterpret_cast<float*>(w + o_qn);
    uint32_t* keymat = reinterpret_cast<uint32_t*>(w + o_keys);
    uint32_t* okeys = d_out_keys ? d_out_keys : reinterpret_cast<uint32_t*>(w + o_ok);
    const float* q = reinterpret_cast<const float*>(d_queries);
    const float* rows = reinterpret_cast<const float*>(s->rows);
    uint64_t* pid = reinterpret_cast<uint64_t*>(w + o_pid);
    uint32_t* pk = reinterpret_cast<uint32_t*>(w + o_pk);
    uint32_t* pc = reinterpret_cast<uint32_t*>(w + o_pc);
    uint64_t* tid = reinterpret_cast<uint64_t*>(w + o_tid);
    uint32_t* tk = reinterpret_cast<uint32_t*>(w + o_tk);
    // select + merge of `m` rows of keys for `cnt` queries, within the partial space reserved for the chunk
    auto select_merge = [&](size_t m, uint32_t cnt, uint64_t* o_ids, uint32_t* o_keys, uint32_t* o_cnt,
                            const uint32_t* run_flag) {
        if (ucfp::select_pruned_ok(m, cnt, k)) {   // a handful of queries: two launches instead of select + merge tree
            ucfp::launch_select_pruned_u32(keymat, s->ids, m, cnt, k, reinterpret_cast<uint32_t*>(w + o_mins), o_ids, o_keys,
                                           o_cnt, st, run_flag);
            return;
        }
        ucfp::SelectPlan pl = ucfp::select_plan(m, cnt);
        if (pl.slices > sp.slices) {
            pl.per_slice = (((m + sp.slices - 1) / sp.slices) + 63) & ~(size_t)63;
            pl.slices = (uint32_t)((m + pl.per_slice - 1) / pl.per_slice);
        }
        ucfp::launch_select_topk_u32(keymat, s->ids, m, pl, cnt, k, pid, pk, pc, st, run_flag);
        ucfp::launch_topk_merge_tree_u32(pid, pk, pl.slices, cnt, k, tid, tk, o_ids, o_keys, o_cnt, st, run_flag);
    };
    // one pass of 5 .. 64 queries: approximate minima -> widened thresholds -> ~k listed chunks per query -> their EXACT keys (the
    // f32 tile's list pass, in slices of the queries its LDS image holds) -> answer.  The dense pass and its selection follow,
    // gated on the flag.
    auto f16_pass = [&](size_t qoff, uint32_t np) -> int {
        const float* qp = q + qoff * dim;
        float* qnp = qn + qoff;
        uint64_t* oi = d_out_ids + qoff * k;
        uint32_t* ok = okeys + qoff * k;
        uint32_t* oc = d_out_cnt + qoff;
        const ucfp::CosinePrunePlan pp = ucfp::cosine_prune_plan(n, np, k, true);
        const float eps = ucfp::cosine_mins_eps(dim);
        uint32_t* pflag = reinterpret_cast<uint32_t*>(w + o_pflag);
        uint32_t* pmin = reinterpret_cast<uint32_t*>(w + o_pmin);
        uint32_t* pwmin = reinterpret_cast<uint32_t*>(w + o_pwmin);
        uint32_t* ptau = reinterpret_cast<uint32_t*>(w + o_ptau);
        uint32_t* pkeys = reinterpret_cast<uint32_t*>(w + o_pkeys);
        ucfp::launch_cosine_norms_image(qp, np, dim, qnp, w + o_qimg, pflag, st);   // (zeroes the flag and the list counter)
        if (getenv("UCFP_COSINE_PRUNE_FALLBACK")) HIP_TRY(hipMemsetAsync(pflag, 1, 1, st));   // tests: the gated dense pass answers
        ucfp::launch_cosine_mins_f16(rows, s->norms, n, dim, w + o_qimg, qnp, np, pp, pmin, pwmin, pflag, st);
        ucfp::launch_prune_tau(pmin, pwmin, pp, np, k, w + o_ptws, ptau, w + o_plist, pflag + 1, w + o_prange, pflag, st, eps);
        const uint32_t per = ucfp::cosine_list_queries(dim);
        ucfp::launch_cosine_keys_list(rows, s->norms, n, dim, qp, qnp, np, pp, w + o_plist, pflag + 1, pkeys, pflag, st);
        ucfp::launch_prune_final(pkeys, pp, w + o_plist, w + o_prange, ptau, s->ids, n, np, k, oi, ok, oc, pflag, st);
        if (np <= per) ucfp::launch_cosine_keys_dense_mfma(rows, s->norms, n, dim, qp, qnp, np, keymat, pflag, st);
        else ucfp::launch_cosine_keys(rows, s->norms, n, dim, qp, qnp, np, keymat, st, pflag);   // (the GEMM: cosine_mins_f16_ok)
        select_merge(n, np, oi, ok, oc, pflag);
        return 0;
    };
    if (prune_f16) {      // (the image kernel writes each pass's norms)
        const size_t npass = (nq + 63) / 64, per_pass = (nq + npass - 1) / npass;
        for (size_t q0 = 0; q0 < nq; q0 += per_pass) {
            const int rc2 = f16_pass(q0, (uint32_t)(nq - q0 < per_pass ? nq - q0 : per_pass));
            if (rc2) return rc2;
        }
        if (d_out_scores) ucfp::launch_cosine_scores_from_keys(okeys, nq * k, d_out_scores, st);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    ucfp::launch_cosine_norms(q, nq, dim, qn, st);
    for (size_t q0 = 0; q0 < nq; q0 += chunk) {
        const size_t qc = nq - q0 < chunk ? nq - q0 : chunk;
        // Batch passes over a large shard do not write the np x n key matrix: the exact answer over the first
        // kSample rows gives every query a threshold (its k-th best key there), the pass over the other rows keeps
        // only the rows that beat it (about k n / kSample per query) in per-query lists, and a wave per query picks
        // the best k of sample answer + list.  A list that overflows (an adversarial row order) raises a flag and
        // the dense pass, select and merge below -- gated on that flag -- answer instead.
        // sample size: the lists expect k n / kSample entries each; an eighth of their capacity keeps an overflow
        // out of reach for rows in random order
        size_t kSample = ((size_t)8 * k * n / cap + 15) & ~(size_t)15;
        if (kSample < 32768) kSample = 32768;
        const uint32_t np0 = (uint32_t)(qc < (size_t)qpp ? qc : (size_t)qpp);
        if (k >= 1 && n >= 8 * kSample && ucfp::cosine_filter_ok(rows, dim, q + q0 * dim, np0, n)) {
            uint32_t* flag = reinterpret_cast<uint32_t*>(w + o_flag);
            uint64_t* bid = reinterpret_cast<uint64_t*>(w + o_bid);
            uint32_t* bk = reinterpret_cast<uint32_t*>(w + o_bk);
            uint32_t* tau = reinterpret_cast<uint32_t*>(w + o_tau);
            float* uq = reinterpret_cast<float*>(w + o_uq);
            uint32_t* cc = reinterpret_cast<uint32_t*>(w + o_cc);
            uint32_t* ck = reinterpret_cast<uint32_t*>(w + o_ck);
            uint32_t* cr = reinterpret_cast<uint32_t*>(w + o_cr);
            for (size_t p0 = 0; p0 < qc; p0 += qpp) {
                const uint32_t np = (uint32_t)(qc - p0 < (size_t)qpp ? qc - p0 : qpp);
                const float* qp = q + (q0 + p0) * dim;
                const float* qnp = qn + q0 + p0;
                uint64_t* oi = d_out_ids + (q0 + p0) * k;
                uint32_t* ok = okeys + (q0 + p0) * k;
                uint32_t* oc = d_out_cnt + q0 + p0;
                if (!ucfp::cosine_filter_ok(rows, dim, qp, np, n)) {   // a short last pass
                    ucfp::launch_cosine_keys(rows, s->norms, n, dim, qp, qnp, np, keymat, st);
                    select_merge(n, np, oi, ok, oc, nullptr);
                    continue;
                }
                HIP_TRY(hipMemsetAsync(flag, 0, 4, st));
                ucfp::launch_cosine_keys(rows, s->norms, kSample, dim, qp, qnp, np, keymat, st);
                select_merge(kSample, np, bid, bk, cc, nullptr);          // cc: scratch for the sample's counts
                ucfp::launch_cosine_tau(bk, k, qnp, np, tau, uq, cc, st);
                ucfp::launch_cosine_keys_filtered(rows + kSample * dim, s->norms + kSample, n - kSample, kSample, dim, qp,
                                                  qnp, np, tau, uq, cc, ck, cr, cap, st);
                ucfp::launch_topk_select_lists_u32(bid, bk, ck, cr, cc, cap, s->ids, np, k, oi, ok, oc, flag, st);
                ucfp::launch_cosine_keys(rows, s->norms, n, dim, qp, qnp, np, keymat, st, flag);
                select_merge(n, np, oi, ok, oc, flag);
            }
            continue;
        }
        if (may_prune && qc <= (size_t)qpp && !getenv("UCFP_COSINE_NO_PRUNE") &&
            ucfp::cosine_prune_ok(rows, dim, q + q0 * dim, (uint32_t)qc, n, k)) {
            // one pass of 5 .. 48 queries: no key matrix (cosine.hip CosinePrune).  The dense pass and its selection follow,
            // gated on the flag the pruned pass raises when a query's ties outgrow its chunk list.
            const uint32_t np = (uint32_t)qc;
            const ucfp::CosinePrunePlan pp = ucfp::cosine_prune_plan(n, np, k);
            uint32_t* pflag = reinterpret_cast<uint32_t*>(w + o_pflag);
            uint32_t* pmin = reinterpret_cast<uint32_t*>(w + o_pmin);
            uint32_t* ptau = reinterpret_cast<uint32_t*>(w + o_ptau);
            uint32_t* pkeys = reinterpret_cast<uint32_t*>(w + o_pkeys);
            uint64_t* oi = d_out_ids + q0 * k;
            uint32_t* ok = okeys + q0 * k;
            uint32_t* oc = d_out_cnt + q0;
            HIP_TRY(hipMemsetAsync(pflag, 0, 8, st));
            if (getenv("UCFP_COSINE_PRUNE_FALLBACK")) HIP_TRY(hipMemsetAsync(pflag, 1, 1, st));   // tests: the gated dense pass answers
            uint32_t* pwmin = reinterpret_cast<uint32_t*>(w + o_pwmin);
            ucfp::launch_cosine_keys_mins(rows, s->norms, n, dim, q + q0 * dim, qn + q0, np, pp, pmin, pwmin, st);
            ucfp::launch_prune_tau(pmin, pwmin, pp, np, k, w + o_ptws, ptau, w + o_plist, pflag + 1, w + o_prange, pflag, st);
            ucfp::launch_cosine_keys_list(rows, s->norms, n, dim, q + q0 * dim, qn + q0, np, pp, w + o_plist, pflag + 1, pkeys,
                                          pflag, st);
            ucfp::launch_prune_final(pkeys, pp, w + o_plist, w + o_prange, ptau, s->ids, n, np, k, oi, ok, oc, pflag, st);
            ucfp::launch_cosine_keys_dense_mfma(rows, s->norms, n, dim, q + q0 * dim, qn + q0, np, keymat, pflag, st);
            select_merge(n, np, oi, ok, oc, pflag);
            continue;
        }
        for (size_t p0 = 0; p0 < qc; p0 += qpp) {
            const uint32_t np = (uint32_t)(qc - p0 < (size_t)qpp ? qc - p0 : qpp);
            ucfp::launch_cosine_keys(rows, s->norms, n, dim, q + (q0 + p0) * dim, qn + q0 + p0, np, keymat + p0 * n, st);
        }
        select_merge(n, (uint32_t)qc, d_out_ids + q0 * k, okeys + q0 * k, d_out_cnt + q0, nullptr);
    }
    if (d_out_scores) ucfp::launch_cosine_scores_from_keys(okeys, nq * k, d_out_scores, st);
    HIP_TRY(hipGetLastError());
    return 0;
}

int check_search_args(ucfp_index* ix, const void* q, size_t nq, uint32_t k, const void* ids, const void* cnt) {
    if (!ix) return capi_fail(UCFP_E_INVALID, "index is NULL");
    if (k > UCFP_INDEX_MAX_K) return capi_fail(UCFP_E_INVALID, "k = %u exceeds UCFP_INDEX_MAX_K = %u", k, UCFP_INDEX_MAX_K);
    if (nq && k && (!q || !ids || !cnt)) return capi_fail(UCFP_E_INVALID, "queries/out_ids/out_counts is NULL");
    if (nq > 4000000u) return capi_fail(UCFP_E_INVALID, "query batch %zu too large for one call (max 4 000 000)", nq);
    return 0;
}

}  // namespace

extern "C" {

int ucfp_index_create(ucfp_ctx* ctx, int kind, uint32_t dim, uint32_t flags, ucfp_index** out) {
    if (!ctx || !out) return capi_fail(UCFP_E_INVALID, "ctx/out is NULL");
    *out = nullptr;
    if (kind != UCFP_INDEX_HAMMING64 && kind != UCFP_INDEX_COSINE_F32)
        return capi_fail(UCFP_E_UNSUPPORTED, "unknown index kind %d", kind);
    if (kind == UCFP_INDEX_COSINE_F32 && (dim == 0 || dim > 65536))
        return capi_fail(UCFP_E_INVALID, "cosine index needs 1 <= dim <= 65536 (got %u)", dim);
    ucfp_index* ix = new (std::nothrow) ucfp_index();
    if (!ix) return capi_fail(UCFP_E_INDEX, "out of host memory");
    ix->ctx = ctx;
    ix->device = ucfp::ctx_device(ctx);
    ix->kind = kind;
    ix->dim = kind == UCFP_INDEX_HAMMING64 ? 1 : dim;
    ix->flags = flags;
    ix->row_bytes = kind == UCFP_INDEX_HAMMING64 ? 8 : (size_t)dim * 4;
    hipError_t e = hipSetDevice(ix->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ix->ws_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->data_ready, hipEventDisableTiming);
    if (e != hipSuccess) {
        delete ix;
        return capi_fail(UCFP_E_INDEX, "index stream creation failed: %s", hipGetErrorString(e));
    }
    *out = ix;
    return UCFP_OK;
}

void ucfp_index_destroy(ucfp_index* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : ix->shards) {
        if (kv.second.ids) (void)hipFree(kv.second.ids);
        if (kv.second.rows) (void)hipFree(kv.second.rows);
        if (kv.second.norms) (void)hipFree(kv.second.norms);
        if (kv.second.order) (void)hipFree(kv.second.order);
    }
    for (auto& w : ix->ws_slot) w.release();
    for (auto& w : ix->direct_state) w.release();
    ix->stage.release();
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    for (hipEvent_t ev : ix->ws_done)
        if (ev) (void)hipEventDestroy(ev);
    if (ix->data_ready) (void)hipEventDestroy(ix->data_ready);
    delete ix;
}

int ucfp_index_upsert(ucfp_index* ix, uint32_t tenant, const uint64_t* ids, const void* rows, size_t n) {
    if (!ix) return capi_fail(UCFP_E_INVALID, "index is NULL");
    if (n == 0) return UCFP_OK;
    if (!ids || !rows) return capi_fail(UCFP_E_INVALID, "ids/rows is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    Shard& s = ix->shards[tenant];
    hipStream_t st = ix->stream;
    for (hipEvent_t ev : ix->ws_done) HIP_TRY(hipStreamWaitEvent(st, ev, 0));
    HIP_TRY(hipStreamWaitEvent(st, ix->data_ready, 0));
    const bool mapped = !(ix->flags & UCFP_INDEX_APPEND_ONLY);
    // resolve target rows on the host; within a batch the LAST occurrence of an id wins
    std::vector<uint64_t> dst(n);
    std::vector<uint8_t> keep(n, 1);
    size_t new_n = s.n;
    if (mapped) {
        std::unordered_map<uint64_t, size_t> last;
        last.reserve(n * 2);
        for (size_t i = 0; i < n; i++) last[ids[i]] = i;
        for (size_t i = 0; i < n; i++) {
            if (last[ids[i]] != i) {
                keep[i] = 0;
                continue;
            }
            auto it = s.pos.find(ids[i]);
            if (it != s.pos.end()) dst[i] = it->second;
            else dst[i] = new_n++;
        }
    } else {
        for (size_t i = 0; i < n; i++) dst[i] = new_n++;
    }
    int rc = shard_reserve(ix, s, new_n, st);
    if (rc) return rc;
    // compact the kept items into the staging buffer: ids | dst | rows | (norms)
    size_t m = 0;
    for (size_t i = 0; i < n; i++) m += keep[i];
    const size_t o_ids = 0, o_dst = align256(m * 8), o_rows = align256(o_dst + m * 8);
    const size_t o_norm = align256(o_rows + m * ix->row_bytes);
    rc = ix->stage.ensure(o_norm + m * 4 + 256);
    if (rc) return rc;
    std::vector<uint64_t> h_ids(m), h_dst(m);
    std::vector<uint8_t> h_rows(m * ix->row_bytes);
    size_t j = 0;
    for (size_t i = 0; i < n; i++) {
        if (!keep[i]) continue;
        h_ids[j] = ids[i];
        h_dst[j] = dst[i];
        memcpy(h_rows.data() + j * ix->row_bytes, (const uint8_t*)rows + i * ix->row_bytes, ix->row_bytes);
        j++;
    }
    uint8_t* sp = ix->stage.p;
    HIP_TRY(hipMemcpyAsync(sp + o_ids, h_ids.data(), m * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(sp + o_dst, h_dst.data(), m * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(sp + o_rows, h_rows.data(), m * ix->row_bytes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)m), dim3(64), 0, st,
                       reinterpret_cast<const uint64_t*>(sp + o_ids), sp + o_rows,
                       reinterpret_cast<const uint64_t*>(sp + o_dst), m, ix->row_bytes, s.ids, s.rows);
    if (ix->kind == UCFP_INDEX_COSINE_F32) {
        ucfp::launch_cosine_norms(reinterpret_cast<const float*>(sp + o_rows), m, ix->dim,
                                  reinterpret_cast<float*>(sp + o_norm), st);
        hipLaunchKernelGGL(gather_norms_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const float*>(sp + o_norm), reinterpret_cast<const uint64_t*>(sp + o_dst),
                           m, s.norms);
    }
    if (!mapped && ix->kind == UCFP_INDEX_HAMMING64 && m) {      // appended rows [s.n, new_n), in order
        if (ensure_order_state(s, st)) return capi_fail(UCFP_E_INDEX, "out of device memory");
        ucfp::launch_ids_order_update(s.ids + s.n, m, s.n == 0, s.order, st);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ix->data_ready, st));
    HIP_TRY(hipStreamSynchronize(st));  // staging vectors are stack-owned
    if (mapped) {
        s.host_ids.resize(new_n);
        for (size_t t = 0; t < m; t++) {
            s.pos[h_ids[t]] = h_dst[t];
            s.host_ids[h_dst[t]] = h_ids[t];
        }
    }
    s.n = new_n;
    return UCFP_OK;
}

int ucfp_index_append_dev(ucfp_index* ix, uint32_t tenant, const uint64_t* d_ids, const void* d_rows, size_t n,
                          void* stream) {
    if (!ix) return capi_fail(UCFP_E_INVALID, "index is NULL");
    if (!(ix->flags & UCFP_INDEX_APPEND_ONLY))
        return capi_fail(UCFP_E_UNSUPPORTED, "append_dev needs an index created with UCFP_INDEX_APPEND_ONLY");
    if (n == 0) return UCFP_OK;
    if (!d_ids || !d_rows) return capi_fail(UCFP_E_INVALID, "ids/rows is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    Shard& s = ix->shards[tenant];
    hipStream_t st = (hipStream_t)stream;
    for (hipEvent_t ev : ix->ws_done) HIP_TRY(hipStreamWaitEvent(st, ev, 0));
    HIP_TRY(hipStreamWaitEvent(st, ix->data_ready, 0));   // an earlier mutation on another stream (growth copies the rows)
    int rc = shard_reserve(ix, s, s.n + n, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(s.ids + s.n, d_ids, n * 8, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.rows + s.n * ix->row_bytes, d_rows, n * ix->row_bytes, hipMemcpyDeviceToDevice, st));
    if (ix->kind == UCFP_INDEX_HAMMING64) {
        // rows are only ever appended here: keep track, on the device, of whether the ids ascend with the row number
        // (they do for a bulk-loaded corpus); the staged Hamming search then tightens its thresholds (hamming_list_tau)
        if (ensure_order_state(s, st)) return capi_fail(UCFP_E_INDEX, "out of device memory");
        ucfp::launch_ids_order_update(s.ids + s.n, n, s.n == 0, s.order, st);
    }
    if (ix->kind == UCFP_INDEX_COSINE_F32)
        ucfp::launch_cosine_norms(reinterpret_cast<const float*>(s.rows + s.n * ix->row_bytes), n, ix->dim,
                                  s.norms + s.n, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ix->data_ready, st));
    s.n += n;
    return UCFP_OK;
}

int ucfp_index_delete(ucfp_index* ix, uint32_t tenant, const uint64_t* ids, size_t n, size_t* n_removed) {
    if (n_removed) *n_removed = 0;
    if (!ix) return capi_fail(UCFP_E_INVALID, "index is NULL");
    if (ix->flags & UCFP_INDEX_APPEND_ONLY)
        return capi_fail(UCFP_E_UNSUPPORTED, "delete on an APPEND_ONLY index");
    if (n && !ids) return capi_fail(UCFP_E_INVALID, "ids is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    auto sit = ix->shards.find(tenant);
    if (sit == ix->shards.end()) return UCFP_OK;  // deleting from an unknown tenant is a no-op
    Shard& s = sit->second;
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = ix->stream;
    for (hipEvent_t ev : ix->ws_done) HIP_TRY(hipStreamWaitEvent(st, ev, 0));
    HIP_TRY(hipStreamWaitEvent(st, ix->data_ready, 0));
    size_t removed = 0;
    for (size_t i = 0; i < n; i++) {
        auto it = s.pos.find(ids[i]);
        if (it == s.pos.end()) continue;
        const size_t r = it->second, last = s.n - 1;
        if (r != last) {
            HIP_TRY(hipMemcpyAsync(s.ids + r, s.ids + last, 8, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(s.rows + r * ix->row_bytes, s.rows + last * ix->row_bytes, ix->row_bytes,
                                   hipMemcpyDeviceToDevice, st));
            if (s.norms) HIP_TRY(hipMemcpyAsync(s.norms + r, s.norms + last, 4, hipMemcpyDeviceToDevice, st));
            const uint64_t moved = s.host_ids[last];
            s.host_ids[r] = moved;
            s.pos[moved] = r;
        }
        s.pos.erase(it);
        s.host_ids.pop_back();
        s.n--;
        removed++;
    }
    HIP_TRY(hipEventRecord(ix->data_ready, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_removed) *n_removed = removed;
    return UCFP_OK;
}

int ucfp_index_size(ucfp_index* ix, uint32_t tenant, size_t* out) {
    if (!ix || !out) return capi_fail(UCFP_E_INVALID, "index/out is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    auto it = ix->shards.find(tenant);
    *out = it == ix->shards.end() ? 0 : it->second.n;
    return UCFP_OK;
}

int ucfp_index_flush(ucfp_index* ix) {
    if (!ix) return capi_fail(UCFP_E_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());
    return UCFP_OK;
}

// ---- snapshot (the sidecar flat file a restart rebuilds the GPU shard from) ------------------
// File: "UCFPIDX1" | kind u32 | dim u32 | row_bytes u64 | tenants u32 | 0 u32 |
//       per tenant: tenant u32 | 0 u32 | n u64 | ids[n] u64 | rows[n * row_bytes]          (little-endian)
// redb stays the reference's source of truth; this file is the device mirror's own checkpoint.
int ucfp_index_save(ucfp_index* ix, const char* path) {
    if (!ix || !path) return capi_fail(UCFP_E_INVALID, "index/path is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());
    FILE* f = fopen(path, "wb");
    if (!f) return capi_fail(UCFP_E_INDEX, "cannot open %s for writing", path);
    bool ok = fwrite("UCFPIDX1", 1, 8, f) == 8;
    const uint32_t hdr[2] = {(uint32_t)ix->kind, ix->dim};
    const uint64_t rb = ix->row_bytes;
    uint32_t nt[2] = {0, 0};
    for (auto& kv : ix->shards) nt[0] += kv.second.n ? 1u : 0u;
    ok = ok && fwrite(hdr, 4, 2, f) == 2 && fwrite(&rb, 8, 1, f) == 1 && fwrite(nt, 4, 2, f) == 2;
    std::vector<uint8_t> host;
    const size_t chunk_rows = 1u << 20;
    for (auto& kv : ix->shards) {
        const Shard& s = kv.second;
        if (!s.n || !ok) continue;
        const uint32_t th[2] = {kv.first, 0};
        const uint64_t n = s.n;
        ok = ok && fwrite(th, 4, 2, f) == 2 && fwrite(&n, 8, 1, f) == 1;
        for (int pass = 0; pass < 2 && ok; pass++) {   // ids, then rows
            const size_t unit = pass == 0 ? 8 : ix->row_bytes;
            const uint8_t* src = pass == 0 ? reinterpret_cast<const uint8_t*>(s.ids) : s.rows;
            for (size_t r0 = 0; r0 < s.n && ok; r0 += chunk_rows) {
                const size_t cnt = s.n - r0 < chunk_rows ? s.n - r0 : chunk_rows;
                host.resize(cnt * unit);
                if (hipMemcpy(host.data(), src + r0 * unit, cnt * unit, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
                ok = ok && fwrite(host.data(), 1, cnt * unit, f) == cnt * unit;
            }
        }
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? UCFP_OK : capi_fail(UCFP_E_INDEX, "short write to %s", path);
}

int ucfp_index_load(ucfp_index* ix, const char* path) {
    if (!ix || !path) return capi_fail(UCFP_E_INVALID, "index/path is NULL");
    FILE* f = fopen(path, "rb");
    if (!f) return capi_fail(UCFP_E_INDEX, "cannot open %s", path);
    char magic[8];
    uint32_t hdr[2], nt[2];
    uint64_t rb;
    int rc = UCFP_OK;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "UCFPIDX1", 8) != 0 || fread(hdr, 4, 2, f) != 2 ||
        fread(&rb, 8, 1, f) != 1 || fread(nt, 4, 2, f) != 2)
        rc = capi_fail(UCFP_E_INDEX, "%s is not an index snapshot", path);
    else if ((int)hdr[0] != ix->kind || hdr[1] != ix->dim || rb != ix->row_bytes)
        rc = capi_fail(UCFP_E_INVALID, "snapshot kind/dim (%u/%u) does not match the index (%d/%u)", hdr[0], hdr[1],
                       ix->kind, ix->dim);
    std::vector<uint64_t> ids;
    std::vector<uint8_t> rows;
    const size_t chunk_rows = 1u << 20;
    for (uint32_t t = 0; rc == UCFP_OK && t < nt[0]; t++) {
        uint32_t th[2];
        uint64_t n;
        if (fread(th, 4, 2, f) != 2 || fread(&n, 8, 1, f) != 1) {
            rc = capi_fail(UCFP_E_INDEX, "truncated snapshot %s", path);
            break;
        }
        // ids precede rows in the file: remember where each section starts and read both in step
        const long ids_at = ftell(f);
        const long rows_at = ids_at + (long)(n * 8);
        for (uint64_t r0 = 0; rc == UCFP_OK && r0 < n; r0 += chunk_rows) {
            const size_t cnt = n - r0 < chunk_rows ? (size_t)(n - r0) : chunk_rows;
            ids.resize(cnt);
            rows.resize(cnt * rb);
            if (fseek(f, ids_at + (long)(r0 * 8), SEEK_SET) != 0 || fread(ids.data(), 8, cnt, f) != cnt ||
                fseek(f, rows_at + (long)(r0 * rb), SEEK_SET) != 0 || fread(rows.data(), 1, cnt * rb, f) != cnt * rb) {
                rc = capi_fail(UCFP_E_INDEX, "truncated snapshot %s", path);
                break;
            }
            rc = ucfp_index_upsert(ix, th[0], ids.data(), rows.data(), cnt);
        }
        if (rc == UCFP_OK && fseek(f, rows_at + (long)(n * rb), SEEK_SET) != 0)
            rc = capi_fail(UCFP_E_INDEX, "truncated snapshot %s", path);
    }
    fclose(f);
    return rc;
}

int ucfp_index_search_dev(ucfp_index* ix, uint32_t tenant, const void* d_queries, size_t nq, uint32_t k,
                          uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_dist, uint32_t* d_out_counts,
                          void* stream) {
    int rc = check_search_args(ix, d_queries, nq, k, d_out_ids, d_out_counts);
    if (rc) return rc;
    if (nq == 0) return UCFP_OK;
    std::lock_guard<std::mutex> lk(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    if (k == 0) {
        HIP_TRY(hipMemsetAsync(d_out_counts, 0, nq * 4, st));
        return UCFP_OK;
    }
    auto it = ix->shards.find(tenant);
    const Shard* s = it == ix->shards.end() ? nullptr : &it->second;
    // a workspace slot is shared: a search may start only after the previous one in its slot (on any stream) is done
    ix->ws_cur = ix->kind == UCFP_INDEX_HAMMING64 ? (ix->ws_next++ & 1u) : 0u;
    HIP_TRY(hipStreamWaitEvent(st, ix->ws_done[ix->ws_cur], 0));
    HIP_TRY(hipStreamWaitEvent(st, ix->data_ready, 0));
    rc = search_shard_dev(ix, s, d_queries, nq, k, d_out_ids, d_out_scores, d_out_dist, d_out_counts, st);
    HIP_TRY(hipEventRecord(ix->ws_done[ix->ws_cur], st));
    return rc;
}

int ucfp_index_search(ucfp_index* ix, uint32_t tenant, const void* queries, size_t nq, uint32_t k,
                      uint64_t* out_ids, float* out_scores, uint32_t* out_dist, uint32_t* out_counts) {
    int rc = check_search_args(ix, queries, nq, k, out_ids, out_counts);
    if (rc) return rc;
    if (nq == 0) return UCFP_OK;
    if (k == 0) {
        memset(out_counts, 0, nq * 4);
        return UCFP_OK;
    }
    // results buffer (own allocation: the shared staging buffer may be re-grown by search)
    const size_t qbytes = nq * ix->row_bytes;
    const size_t o_q = 0, o_ids = align256(qbytes), o_sc = align256(o_ids + nq * k * 8);
    const size_t o_d = align256(o_sc + nq * k * 4), o_c = align256(o_d + nq * k * 4);
    const size_t total = align256(o_c + nq * 4);
    hipStream_t st;
    uint8_t* buf = nullptr;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        HIP_TRY(hipSetDevice(ix->device));
        st = ix->stream;
        HIP_TRY(hipMalloc((void**)&buf, total));
    }
    hipError_t e = hipMemcpyAsync(buf + o_q, queries, qbytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        rc = ucfp_index_search_dev(ix, tenant, buf + o_q, nq, k, reinterpret_cast<uint64_t*>(buf + o_ids),
                                   reinterpret_cast<float*>(buf + o_sc), reinterpret_cast<uint32_t*>(buf + o_d),
                                   reinterpret_cast<uint32_t*>(buf + o_c), st);
        if (rc == 0) {
            e = hipMemcpyAsync(out_ids, buf + o_ids, nq * k * 8, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && out_scores)
                e = hipMemcpyAsync(out_scores, buf + o_sc, nq * k * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && out_dist)
                e = hipMemcpyAsync(out_dist, buf + o_d, nq * k * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(out_counts, buf + o_c, nq * 4, hipMemcpyDeviceToHost, st);
        }
    }
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(buf);
    if (rc) return rc;
    if (e != hipSuccess) return capi_fail(UCFP_E_INDEX, "search copy failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return capi_fail(UCFP_E_INDEX, "search failed: %s", hipGetErrorString(e2));
    return UCFP_OK;
}

int ucfp_topk_merge_dev(ucfp_ctx* ctx, int kind, const uint64_t* d_part_ids, const uint32_t* d_part_keys,
                        uint32_t parts, size_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_scores,
                        uint32_t* d_out_keys, uint32_t* d_out_counts, void* stream) {
    if (!ctx) return capi_fail(UCFP_E_INVALID, "ctx is NULL");
    if (kind != UCFP_INDEX_HAMMING64 && kind != UCFP_INDEX_COSINE_F32)
        return capi_fail(UCFP_E_UNSUPPORTED, "unknown index kind %d", kind);
    if (nq == 0 || k == 0) return UCFP_OK;
    if (!d_part_ids || !d_part_keys || !d_out_ids || !d_out_keys || !d_out_counts)
        return capi_fail(UCFP_E_INVALID, "merge buffers must not be NULL");
    if (k > UCFP_INDEX_MAX_K) return capi_fail(UCFP_E_INVALID, "k too large");
    hipStream_t st = (hipStream_t)stream;
    ucfp::launch_topk_merge_u32(d_part_ids, d_part_keys, parts, (uint32_t)nq, k, d_out_ids, d_out_keys, d_out_counts,
                                nullptr, st);
    if (d_out_scores) {
        if (kind == UCFP_INDEX_HAMMING64) ucfp::launch_hamming_scores(d_out_keys, nq * k, d_out_scores, st);
        else ucfp::launch_cosine_scores_from_keys(d_out_keys, nq * k, d_out_scores, st);
    }
    HIP_TRY(hipGetLastError());
    return UCFP_OK;
}

}  // extern "C"
