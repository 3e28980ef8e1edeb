// batcher_search.hip -- host micro-batcher in front of IndexBackend::knn (SURVEY 8f N1 for the QUERY route).
//
// The reference answers /v1/query one request at a time (src/server/handlers.rs:143-187: one vector, one knn call per request)
// with up to 512 requests in flight (src/bin/ucfp.rs:267).  One query per launch leaves the GPU idle between requests
// and pays a whole corpus pass per query: 30 k QPS over a 12.5 M-code shard where a launch of 64 queries answers 570 k.
// Here every request thread calls a BLOCKING submit() with its own query and its own k; a worker thread coalesces whatever is
// pending -- up to max_batch queries, or what arrived within max_delay_us of the first -- into ONE H2D copy of the queries,
// ONE ucfp_index_search_dev over the batch with k = the largest k asked for (best-first order is total -- (distance, id) /
// (score, id) -- so a request's answer is the first k entries of its row) and ONE D2H copy of the results, then wakes the
// submitters.  The claim / commit / wake protocol is batch_core.h's (no mutex on the request path).

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <new>

#include "../../include/ucfp_hip.h"
#include "batch_core.h"
#include "common.h"

namespace ucfp {
int capi_fail(int code, const char* fmt, ...);
int index_kind(const ucfp_index* ix);
int index_device(const ucfp_index* ix);
uint32_t index_dim(const ucfp_index* ix);
}  // namespace ucfp
using ucfp::capi_fail;

// Staging of one set: queries [max_batch x row_bytes] in; out = ids [n][kmax] u64 | scores [n][kmax] f32 | dist [n][kmax] u32 |
// counts [n], each region sized for max_batch x UCFP_INDEX_MAX_K.
struct ucfp_search_batcher {
    ucfp_index* idx = nullptr;
    int device = 0, kind = 0;
    uint32_t tenant = 0;
    size_t max_batch = 0, row_bytes = 0;
    size_t o_sc = 0, o_d = 0, o_cnt = 0, out_bytes = 0;
    uint8_t* h_q[2] = {nullptr, nullptr};
    uint8_t* h_out[2] = {nullptr, nullptr};
    uint8_t* d_q = nullptr;
    uint8_t* d_out = nullptr;
    std::atomic<uint32_t> kmax[2];       // largest k asked for in the set being filled
    uint32_t kused[2] = {0, 0};          // ... and the k the set's search ran with (row stride of its results)
    size_t oset[2][3] = {{0, 0, 0}, {0, 0, 0}};   // where the set's scores / distances / counts start in its result block
    hipStream_t stream = nullptr;
    ucfp::BatchCore core;
};

namespace {

int run_set(ucfp_search_batcher* b, int s, size_t n) {
    (void)hipSetDevice(b->device);
    const uint32_t k = b->kmax[s].exchange(0);
    b->kused[s] = k;
    hipError_t e = hipMemcpyAsync(b->d_q, b->h_q[s], n * b->row_bytes, hipMemcpyHostToDevice, b->stream);
    if (e != hipSuccess) return UCFP_E_INDEX;
    // the set's results packed back to back for THIS n and k -- ids [n][k] | scores | distances | counts -- so that they come
    // home in ONE copy (four separate ones cost four launches' worth of host time per flush)
    const size_t e_nk = n * (size_t)k;
    b->oset[s][0] = e_nk * 8;
    b->oset[s][1] = e_nk * 12;
    b->oset[s][2] = e_nk * 16;
    uint64_t* d_ids = reinterpret_cast<uint64_t*>(b->d_out);
    float* d_sc = reinterpret_cast<float*>(b->d_out + b->oset[s][0]);
    uint32_t* d_d = reinterpret_cast<uint32_t*>(b->d_out + b->oset[s][1]);
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(b->d_out + b->oset[s][2]);
    const int rc = ucfp_index_search_dev(b->idx, b->tenant, b->d_q, n, k, d_ids, d_sc, d_d, d_cnt, b->stream);
    if (rc) return rc;
    e = hipMemcpyAsync(b->h_out[s], b->d_out, e_nk * 16 + n * 4, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    return e == hipSuccess ? UCFP_OK : UCFP_E_INDEX;
}

void teardown(ucfp_search_batcher* b) {
    b->core.stop();
    (void)hipSetDevice(b->device);
    for (int s = 0; s < 2; s++) {
        if (b->h_q[s]) (void)hipHostFree(b->h_q[s]);
        if (b->h_out[s]) (void)hipHostFree(b->h_out[s]);
    }
    if (b->d_q) (void)hipFree(b->d_q);
    if (b->d_out) (void)hipFree(b->d_out);
    if (b->stream) (void)hipStreamDestroy(b->stream);
}

}  // namespace

extern "C" {

int ucfp_index_search_batcher_create(ucfp_index* idx, uint32_t tenant, size_t max_batch, uint32_t max_delay_us,
                                     ucfp_search_batcher** out) {
    if (!idx || !out) return capi_fail(UCFP_E_INVALID, "index/out is NULL");
    *out = nullptr;
    if (max_batch == 0 || max_batch > 4096) return capi_fail(UCFP_E_INVALID, "search batcher needs 1 <= max_batch <= 4096");
    ucfp_search_batcher* b = new (std::nothrow) ucfp_search_batcher();
    if (!b) return capi_fail(UCFP_E_INDEX, "out of host memory");
    b->idx = idx;
    b->device = ucfp::index_device(idx);
    b->kind = ucfp::index_kind(idx);
    b->tenant = tenant;
    b->max_batch = max_batch;
    b->row_bytes = b->kind == UCFP_INDEX_HAMMING64 ? 8 : (size_t)ucfp::index_dim(idx) * 4;
    b->kmax[0] = b->kmax[1] = 0;
    const size_t e = max_batch * UCFP_INDEX_MAX_K;
    b->o_sc = e * 8;
    b->o_d = b->o_sc + e * 4;
    b->o_cnt = b->o_d + e * 4;
    b->out_bytes = b->o_cnt + max_batch * 4;
    hipError_t er = hipSetDevice(b->device);
    for (int s = 0; s < 2 && er == hipSuccess; s++) {
        er = hipHostMalloc((void**)&b->h_q[s], max_batch * b->row_bytes, hipHostMallocDefault);
        if (er == hipSuccess) er = hipHostMalloc((void**)&b->h_out[s], b->out_bytes, hipHostMallocDefault);
    }
    if (er == hipSuccess) er = hipMalloc((void**)&b->d_q, max_batch * b->row_bytes);
    if (er == hipSuccess) er = hipMalloc((void**)&b->d_out, b->out_bytes);
    if (er == hipSuccess) er = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (er != hipSuccess) {
        teardown(b);
        delete b;
        return capi_fail(UCFP_E_INDEX, "search batcher allocation failed: %s", hipGetErrorString(er));
    }
    b->core.start(max_batch, max_batch, max_delay_us, [b](int s, size_t n, size_t) { return run_set(b, s, n); });
    *out = b;
    return UCFP_OK;
}

void ucfp_index_search_batcher_destroy(ucfp_search_batcher* b) {
    if (!b) return;
    teardown(b);
    delete b;
}

int ucfp_index_search_batcher_submit(ucfp_search_batcher* b, const void* query, uint32_t k, uint64_t* out_ids, float* out_scores,
                                     uint32_t* out_dist, uint32_t* out_count) {
    if (!b || !query || !out_count) return capi_fail(UCFP_E_INVALID, "batcher/query/out_count is NULL");
    *out_count = 0;
    if (k == 0) return UCFP_OK;                         // the reference returns no hits for k = 0 (embedded/mod.rs:275-277)
    if (k > UCFP_INDEX_MAX_K) return capi_fail(UCFP_E_INVALID, "k = %u exceeds UCFP_INDEX_MAX_K", k);
    if (!out_ids) return capi_fail(UCFP_E_INVALID, "out_ids is NULL");
    ucfp::BatchCore::Ticket t;
    if (!b->core.claim(1, &t)) return capi_fail(UCFP_E_INDEX, "batcher is shutting down");
    memcpy(b->h_q[t.set] + t.slot * b->row_bytes, query, b->row_bytes);
    uint32_t cur = b->kmax[t.set].load();
    while (cur < k && !b->kmax[t.set].compare_exchange_weak(cur, k)) {
    }
    b->core.commit(t);
    const int rc = b->core.wait(t);
    if (rc == UCFP_OK) {
        const uint32_t ks = b->kused[t.set];
        const uint8_t* h = b->h_out[t.set];
        uint32_t cnt = reinterpret_cast<const uint32_t*>(h + b->oset[t.set][2])[t.slot];
        cnt = cnt < k ? cnt : k;
        memcpy(out_ids, h + (t.slot * (size_t)ks) * 8, (size_t)k * 8);
        if (out_scores) memcpy(out_scores, h + b->oset[t.set][0] + (t.slot * (size_t)ks) * 4, (size_t)k * 4);
        if (out_dist) memcpy(out_dist, h + b->oset[t.set][1] + (t.slot * (size_t)ks) * 4, (size_t)k * 4);
        // places past the hits are "unused" in the batch row only beyond ITS count: mark this request's own
        for (uint32_t r = cnt; r < k; r++) {
            out_ids[r] = UCFP_INVALID_ID;
            if (out_scores) out_scores[r] = -1.0f;
            if (out_dist) out_dist[r] = 0xffffffffu;
        }
        *out_count = cnt;
    }
    b->core.release(t);
    if (rc != UCFP_OK) return capi_fail(rc, "batched search failed");
    return UCFP_OK;
}

int ucfp_index_search_batcher_stats(ucfp_search_batcher* b, uint64_t* batches, uint64_t* items) {
    if (!b) return capi_fail(UCFP_E_INVALID, "batcher is NULL");
    b->core.stats(batches, items);
    return UCFP_OK;
}

}  // extern "C"
