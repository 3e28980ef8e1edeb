// shard.hip -- the sharded search of SURVEY 8e behind the C ABI: "Multi-GPU variant ... shards are internal" of the
// `IndexBackend::knn` seam (src/index/mod.rs:29-35).
//
// One process per GPU.  The corpus is range-partitioned over the ranks; a query batch is replicated; each rank
// searches its own shard; the ONLY data-path exchange is ONE ncclAllGather (RCCL over xGMI) of the per-shard top-k as
// packed 16-byte entries {id u64, key u32, pad u32} -- nq x k x 16 B per rank, a few hundred KB: latency-bound on
// xGMI, nowhere near the per-link bandwidth -- and every rank then runs the same deterministic merge
// ((key asc, id asc)) and holds the full answer.  Because the exchange is latency, it is taken off the critical
// path: `submit` runs the shard scan on the caller's stream and the all-gather + merge on this communicator's side
// stream; with two buffer sets the exchange of batch i overlaps the shard scan of batch i + 1.
//
// RCCL is resolved at run time (dlopen "librccl.so.1"): libucfp_hip.so carries no link-time dependency on it, a
// single-GPU host never loads it, and inside a process that already holds RCCL (torch.distributed) the same
// instance is shared.  <rccl/rccl.h> is included for its types only.

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/ucfp_hip.h"
#include "common.h"

namespace ucfp {
int capi_fail(int code, const char* fmt, ...);  // capi.hip
int ctx_device(const ucfp_ctx* ctx);            // capi.hip
int index_kind(const ucfp_index* ix);           // index.hip
int index_device(const ucfp_index* ix);         // index.hip
}  // namespace ucfp

using ucfp::capi_fail;

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return capi_fail(UCFP_E_INDEX, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    char why[256] = "";
};

RcclApi& rccl_state() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char* de = dlerror();
            snprintf(api.why, sizeof api.why, "cannot load RCCL (librccl.so.1): %s", de ? de : "?");
            return;
        }
        api.get_unique_id = (decltype(api.get_unique_id))dlsym(api.handle, "ncclGetUniqueId");
        api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(api.handle, "ncclCommInitRank");
        api.comm_destroy = (decltype(api.comm_destroy))dlsym(api.handle, "ncclCommDestroy");
        api.all_gather = (decltype(api.all_gather))dlsym(api.handle, "ncclAllGather");
        api.error_string = (decltype(api.error_string))dlsym(api.handle, "ncclGetErrorString");
        if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_gather || !api.error_string) {
            snprintf(api.why, sizeof api.why, "the RCCL library found lacks a required symbol");
            api.handle = nullptr;
        }
    });
    return api;
}
RcclApi* rccl() { RcclApi& a = rccl_state(); return a.handle ? &a : nullptr; }
const char* rccl_why() { return rccl_state().why; }

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

struct ucfp_shard_comm {
    ucfp_ctx* ctx = nullptr;
    int device = 0;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t xs = nullptr;       // exchange stream: all-gather + merge
    hipStream_t ls[2] = {nullptr, nullptr};   // scan streams, one per buffer set: the shard scans of two batches in flight overlap
    std::mutex mu;
    struct Set {
        uint8_t* buf = nullptr;
        size_t cap = 0;
        hipEvent_t queued = nullptr, searched = nullptr, done = nullptr;
        uint64_t ticket = 0;
        bool has_mask = false;          // the merge of this set's batch wrote the missing-shard mask at buf + missing_off
        size_t missing_off = 0;
    } sets[2];
    uint64_t next_ticket = 1;
    uint64_t exchanges = 0;         // all-gathers issued (stats / tests)
};

extern "C" {

int ucfp_shard_unique_id(uint8_t uid[UCFP_SHARD_UID_BYTES]) {
    if (!uid) return capi_fail(UCFP_E_INVALID, "uid is NULL");
    RcclApi* api = rccl();
    if (!api) return capi_fail(UCFP_E_UNSUPPORTED, "%s", rccl_why());
    static_assert(sizeof(ncclUniqueId) == UCFP_SHARD_UID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = api->get_unique_id(&id);
    if (r != ncclSuccess) return capi_fail(UCFP_E_INDEX, "ncclGetUniqueId: %s", api->error_string(r));
    memcpy(uid, &id, sizeof id);
    return UCFP_OK;
}

int ucfp_shard_comm_create_ex(ucfp_ctx* ctx, const uint8_t uid[UCFP_SHARD_UID_BYTES], int rank, int world,
                              uint32_t flags, ucfp_shard_comm** out) {
    if (!ctx || !out) return capi_fail(UCFP_E_INVALID, "ctx/out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return capi_fail(UCFP_E_INVALID, "rank %d outside world %d", rank, world);
    if (world > 64) return capi_fail(UCFP_E_UNSUPPORTED, "world %d > 64 shards per search", world);
    if (flags & ~(uint32_t)UCFP_SHARD_FORCE_RCCL) return capi_fail(UCFP_E_INVALID, "unknown flag bits 0x%x", flags);
    // A one-rank communicator is legal in RCCL.  Forcing it makes a single-GPU host run the very code the multi-GPU
    // job runs -- dlopen, ncclCommInitRank, ncclAllGather on the exchange stream, the merge over `recv` -- instead of
    // the local short cut.
    const char* env = getenv("UCFP_SHARD_FORCE_RCCL");
    const bool use_rccl = world > 1 || (flags & UCFP_SHARD_FORCE_RCCL) || (env && env[0] == '1');
    if (use_rccl && !uid) return capi_fail(UCFP_E_INVALID, "uid is NULL (rank 0 makes it with ucfp_shard_unique_id)");
    ucfp_shard_comm* c = new (std::nothrow) ucfp_shard_comm();
    if (!c) return capi_fail(UCFP_E_INDEX, "out of host memory");
    c->ctx = ctx;
    c->device = ucfp::ctx_device(ctx);
    c->rank = rank;
    c->world = world;
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipStreamCreateWithFlags(&c->ls[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->sets[i].queued, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->sets[i].searched, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->sets[i].done, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        ucfp_shard_comm_destroy(c);
        return capi_fail(UCFP_E_INDEX, "shard communicator setup failed: %s", hipGetErrorString(e));
    }
    if (use_rccl) {
        RcclApi* api = rccl();
        if (!api) {
            ucfp_shard_comm_destroy(c);
            return capi_fail(UCFP_E_UNSUPPORTED, "%s", rccl_why());
        }
        ncclUniqueId id;
        memcpy(&id, uid, sizeof id);
        ncclResult_t r = api->comm_init_rank(&c->comm, world, id, rank);   // collective: every rank of the job calls it
        if (r != ncclSuccess) {
            c->comm = nullptr;
            ucfp_shard_comm_destroy(c);
            return capi_fail(UCFP_E_INDEX, "ncclCommInitRank(rank %d of %d): %s", rank, world, api->error_string(r));
        }
    }
    *out = c;
    return UCFP_OK;
}

int ucfp_shard_comm_create(ucfp_ctx* ctx, const uint8_t uid[UCFP_SHARD_UID_BYTES], int rank, int world,
                           ucfp_shard_comm** out) {
    return ucfp_shard_comm_create_ex(ctx, uid, rank, world, 0, out);
}

void ucfp_shard_comm_destroy(ucfp_shard_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (hipStream_t l : c->ls)
        if (l) (void)hipStreamSynchronize(l);
    if (c->xs) (void)hipStreamSynchronize(c->xs);
    if (c->comm) {
        RcclApi* api = rccl();
        if (api) (void)api->comm_destroy(c->comm);
    }
    for (auto& s : c->sets) {
        if (s.buf) (void)hipFree(s.buf);
        if (s.queued) (void)hipEventDestroy(s.queued);
        if (s.searched) (void)hipEventDestroy(s.searched);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (c->xs) (void)hipStreamDestroy(c->xs);
    for (hipStream_t l : c->ls)
        if (l) (void)hipStreamDestroy(l);
    delete c;
}

int ucfp_shard_comm_uses_rccl(ucfp_shard_comm* c) { return c && c->comm ? 1 : 0; }

int ucfp_shard_comm_info(ucfp_shard_comm* c, int* rank, int* world, uint64_t* exchanges) {
    if (!c) return capi_fail(UCFP_E_INVALID, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (exchanges) {
        std::lock_guard<std::mutex> lk(c->mu);
        *exchanges = c->exchanges;
    }
    return UCFP_OK;
}

void ucfp_shard_range(uint64_t n_total, int rank, int world, uint64_t* start, uint64_t* end) {
    const uint64_t w = world > 0 ? (uint64_t)world : 1, r = rank > 0 ? (uint64_t)rank : 0;
    const uint64_t base = n_total / w, rem = n_total % w;
    const uint64_t s = r * base + (r < rem ? r : rem);
    if (start) *start = s;
    if (end) *end = s + base + (r < rem ? 1 : 0);
}

int ucfp_topk_pack_dev(ucfp_ctx* ctx, const uint64_t* d_ids, const uint32_t* d_keys, size_t nq, uint32_t k,
                       void* d_entries, void* stream) {
    if (!ctx) return capi_fail(UCFP_E_INVALID, "ctx is NULL");
    if (nq == 0 || k == 0) return UCFP_OK;
    if (!d_ids || !d_keys || !d_entries) return capi_fail(UCFP_E_INVALID, "pack buffers must not be NULL");
    ucfp::launch_topk_pack_entries(d_ids, d_keys, nq * k, d_entries, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UCFP_OK;
}

int ucfp_topk_merge_packed_dev(ucfp_ctx* ctx, int kind, const void* d_entries, uint32_t parts, size_t nq, uint32_t k,
                               uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_keys, uint32_t* d_out_counts,
                               void* stream) {
    return ucfp_topk_merge_packed_ex_dev(ctx, kind, d_entries, parts, nq, k, d_out_ids, d_out_scores, d_out_keys, d_out_counts,
                                         nullptr, stream);
}

int ucfp_topk_merge_packed_ex_dev(ucfp_ctx* ctx, int kind, const void* d_entries, uint32_t parts, size_t nq, uint32_t k,
                                  uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_keys, uint32_t* d_out_counts,
                                  uint64_t* d_missing, void* stream) {
    if (!ctx) return capi_fail(UCFP_E_INVALID, "ctx is NULL");
    if (kind != UCFP_INDEX_HAMMING64 && kind != UCFP_INDEX_COSINE_F32)
        return capi_fail(UCFP_E_UNSUPPORTED, "unknown index kind %d", kind);
    if (nq == 0 || k == 0) return UCFP_OK;
    if (!d_entries || !d_out_ids || !d_out_keys || !d_out_counts)
        return capi_fail(UCFP_E_INVALID, "merge buffers must not be NULL");
    if (k > UCFP_INDEX_MAX_K) return capi_fail(UCFP_E_INVALID, "k too large");
    if ((size_t)parts * k > 2048 && parts > 64) return capi_fail(UCFP_E_UNSUPPORTED, "more than 64 shards");
    hipStream_t st = (hipStream_t)stream;
    if (d_missing && parts > 64) return capi_fail(UCFP_E_UNSUPPORTED, "the missing-shard mask holds 64 parts");
    ucfp::launch_topk_merge_packed(d_entries, parts, (uint32_t)nq, k, d_out_ids, d_out_keys, d_out_counts, st, d_missing);
    if (d_out_scores) {
        if (kind == UCFP_INDEX_HAMMING64) ucfp::launch_hamming_scores(d_out_keys, nq * k, d_out_scores, st);
        else ucfp::launch_cosine_scores_from_keys(d_out_keys, nq * k, d_out_scores, st);
    }
    HIP_TRY(hipGetLastError());
    return UCFP_OK;
}

int ucfp_index_search_sharded_submit(ucfp_index* idx, ucfp_shard_comm* c, uint32_t tenant, const void* d_queries,
                                     size_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_scores,
                                     uint32_t* d_out_keys, uint32_t* d_out_counts, void* stream, uint64_t* ticket) {
    if (!idx || !c || !ticket) return capi_fail(UCFP_E_INVALID, "index/comm/ticket is NULL");
    *ticket = 0;
    if (ucfp::index_device(idx) != c->device)
        return capi_fail(UCFP_E_INVALID, "index lives on device %d, communicator on device %d", ucfp::index_device(idx),
                         c->device);
    if (k > UCFP_INDEX_MAX_K) return capi_fail(UCFP_E_INVALID, "k = %u exceeds UCFP_INDEX_MAX_K", k);
    if (nq > 1000000u) return capi_fail(UCFP_E_INVALID, "query batch %zu too large for one sharded call", nq);
    if (nq && k && (!d_queries || !d_out_ids || !d_out_counts)) return capi_fail(UCFP_E_INVALID, "NULL buffer");
    const int kind = ucfp::index_kind(idx);
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t t = c->next_ticket++;
    ucfp_shard_comm::Set& S = c->sets[t & 1];
    S.ticket = t;
    *ticket = t;
    S.has_mask = false;
    if (nq == 0 || k == 0) {
        // every rank still takes part in nothing: an empty batch is empty everywhere (queries are replicated)
        HIP_TRY(hipStreamWaitEvent(st, S.done, 0));
        if (nq) HIP_TRY(hipMemsetAsync(d_out_counts, 0, nq * 4, st));
        HIP_TRY(hipEventRecord(S.done, st));
        return UCFP_OK;
    }
    // The shard scan runs on the set's own stream, behind the caller's stream (the queries are written) and behind the
    // exchange that last used this buffer set: the scans of the two batches in flight then overlap -- one batch's short
    // staging kernels (sample, rescans, thresholds, selection) fill the gaps of the other's matrix-core scan.
    hipStream_t ls = c->ls[t & 1];
    HIP_TRY(hipEventRecord(S.queued, st));
    HIP_TRY(hipStreamWaitEvent(ls, S.queued, 0));
    HIP_TRY(hipStreamWaitEvent(ls, S.done, 0));
    const size_t e = nq * k;
    const size_t o_ids = 0, o_keys = align256(e * 8), o_cnt = align256(o_keys + e * 4), o_send = align256(o_cnt + nq * 4);
    const size_t o_recv = align256(o_send + e * 16), o_okeys = align256(o_recv + (size_t)c->world * e * 16);
    const size_t o_missing = align256(o_okeys + e * 4);
    const size_t need = align256(o_missing + 8);
    if (S.cap < need) {
        HIP_TRY(hipStreamSynchronize(c->xs));
        HIP_TRY(hipStreamSynchronize(ls));
        if (S.buf) (void)hipFree(S.buf);
        S.buf = nullptr;
        S.cap = 0;
        HIP_TRY(hipMalloc((void**)&S.buf, need + need / 4));
        S.cap = need + need / 4;
    }
    uint64_t* l_ids = reinterpret_cast<uint64_t*>(S.buf + o_ids);
    uint32_t* l_keys = reinterpret_cast<uint32_t*>(S.buf + o_keys);
    uint32_t* l_cnt = reinterpret_cast<uint32_t*>(S.buf + o_cnt);
    uint8_t* send = S.buf + o_send;
    const bool exchange = c->comm != nullptr;   // world > 1, or a forced one-rank communicator
    uint8_t* recv = exchange ? S.buf + o_recv : send;
    uint32_t* okeys = d_out_keys ? d_out_keys : reinterpret_cast<uint32_t*>(S.buf + o_okeys);
    // 1. this rank's shard, on the set's scan stream
    int rc = ucfp_index_search_dev(idx, tenant, d_queries, nq, k, l_ids, nullptr, l_keys, l_cnt, ls);
    if (rc && !exchange) return rc;
    // From here to the all-gather NOTHING returns on this rank alone: the other ranks are about to enter the collective
    // and would wait in it for ever.  A local failure (the scan could not be enqueued, a stream call failed) is kept in
    // `rc`, the rank joins the all-gather with an all-0xff list -- id 2^64-1 and key 2^32-1 in every place like a shard
    // with no hit, and the PAD word 0xffffffff where a packed entry has 0: the mark of a missing shard, which the merge
    // on every rank turns into the ticket's `missing` mask (ucfp_index_search_sharded_missing) -- and the error is
    // reported after the collective has been issued.
    auto keep = [&](hipError_t e_, const char* what) {
        if (e_ != hipSuccess && rc == UCFP_OK) rc = capi_fail(UCFP_E_INDEX, "%s failed: %s", what, hipGetErrorString(e_));
    };
    if (rc == UCFP_OK) {
        ucfp::launch_topk_pack_entries(l_ids, l_keys, e, send, ls);
        keep(hipGetLastError(), "pack kernel");
    }
    if (rc) (void)hipMemsetAsync(send, 0xff, e * 16, ls);
    // 2. exchange + merge, on the side stream (the scan stream when there is nothing to exchange)
    hipStream_t xs = exchange ? c->xs : ls;
    if (exchange) {
        keep(hipEventRecord(S.searched, ls), "hipEventRecord");
        // (if the event calls failed the all-gather may read a stale send buffer: the rank still JOINS, its error is returned)
        keep(hipStreamWaitEvent(xs, S.searched, 0), "hipStreamWaitEvent");
        RcclApi* api = rccl();
        ncclResult_t r = api->all_gather(send, recv, e * 2, ncclUint64, c->comm, xs);   // 16-byte entries as 2 x u64
        if (r != ncclSuccess) return capi_fail(UCFP_E_INDEX, "ncclAllGather: %s", api->error_string(r));
        c->exchanges++;
    }
    ucfp::launch_topk_merge_packed(recv, (uint32_t)c->world, (uint32_t)nq, k, d_out_ids, okeys, d_out_counts, xs,
                                   reinterpret_cast<uint64_t*>(S.buf + o_missing));
    if (d_out_scores) {
        if (kind == UCFP_INDEX_HAMMING64) ucfp::launch_hamming_scores(okeys, e, d_out_scores, xs);
        else ucfp::launch_cosine_scores_from_keys(okeys, e, d_out_scores, xs);
    }
    keep(hipGetLastError(), "merge kernels");
    keep(hipEventRecord(S.done, xs), "hipEventRecord");
    S.has_mask = true;
    S.missing_off = o_missing;
    return rc;   // non-zero: this rank's shard is missing from the answer every rank now holds (ucfp_last_error says why)
}

int ucfp_index_search_sharded_missing(ucfp_shard_comm* c, uint64_t ticket, uint64_t* missing_mask) {
    if (!c || !missing_mask) return capi_fail(UCFP_E_INVALID, "comm/mask is NULL");
    *missing_mask = 0;
    std::lock_guard<std::mutex> lk(c->mu);
    if (ticket == 0 || ticket >= c->next_ticket) return capi_fail(UCFP_E_INVALID, "unknown ticket %llu", (unsigned long long)ticket);
    ucfp_shard_comm::Set& S = c->sets[ticket & 1];
    if (S.ticket != ticket) return capi_fail(UCFP_E_INVALID, "ticket %llu: its buffers hold a later batch", (unsigned long long)ticket);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(S.done));
    if (!S.has_mask) return UCFP_OK;          // an empty batch: nothing was exchanged
    HIP_TRY(hipMemcpy(missing_mask, S.buf + S.missing_off, 8, hipMemcpyDeviceToHost));
    return UCFP_OK;
}

int ucfp_index_search_sharded_collect(ucfp_shard_comm* c, uint64_t ticket, void* stream) {
    if (!c) return capi_fail(UCFP_E_INVALID, "comm is NULL");
    std::lock_guard<std::mutex> lk(c->mu);
    if (ticket == 0 || ticket >= c->next_ticket) return capi_fail(UCFP_E_INVALID, "unknown ticket %llu",
                                                                  (unsigned long long)ticket);
    HIP_TRY(hipSetDevice(c->device));
    // the set's `done` event is that of `ticket` or of a later batch using the same set: waiting on it is sufficient
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, c->sets[ticket & 1].done, 0));
    return UCFP_OK;
}

int ucfp_index_search_sharded_dev(ucfp_index* idx, ucfp_shard_comm* c, uint32_t tenant, const void* d_queries, size_t nq,
                                  uint32_t k, uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_keys,
                                  uint32_t* d_out_counts, void* stream) {
    uint64_t t = 0;
    int rc = ucfp_index_search_sharded_submit(idx, c, tenant, d_queries, nq, k, d_out_ids, d_out_scores, d_out_keys,
                                              d_out_counts, stream, &t);
    if (rc) return rc;
    return ucfp_index_search_sharded_collect(c, t, stream);
}

}  // extern "C"
