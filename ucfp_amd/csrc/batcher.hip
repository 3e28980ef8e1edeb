// batcher.hip -- host micro-batcher in front of the image entry point.
//
// The reference hashes one item per HTTP request on a tokio worker (src/server/handlers.rs:249-252),
// up to 512 requests in flight (src/bin/ucfp.rs:267).  A GPU wants batches.  This object lets every
// request thread call a BLOCKING submit() with its own frame; a worker thread coalesces whatever is
// pending -- up to `max_batch` frames, or whatever arrived within `max_delay_us` of the first one --
// into ONE pinned-memory H2D copy, ONE kernel launch and ONE D2H copy, then wakes the submitters.
// Each submitter copies its frame into its pinned slot itself, so the host-side memcpy work is
// spread over the request threads.  Fixed geometry per batcher; other geometries use the direct call.

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/ucfp_hip.h"
#include "common.h"

namespace ucfp {
int capi_fail(int code, const char* fmt, ...);
int ctx_device(const ucfp_ctx* ctx);
int image_hash_ordered(ucfp_ctx* ctx, uint32_t algo, const uint8_t* frames, size_t n, uint32_t w, uint32_t h,
                       size_t row_stride, size_t frame_stride, int pixfmt, uint32_t min_dim, uint32_t max_dim,
                       const uint8_t* exact, uint8_t* out, int32_t* status, hipStream_t stream);
}  // namespace ucfp
using ucfp::capi_fail;

struct ucfp_image_batcher {
    ucfp_ctx* ctx = nullptr;
    int device = 0;
    uint32_t algo = 0, width = 0, height = 0;
    int pixfmt = 0;
    ucfp_image_preprocess pre{8192, 32};
    size_t max_batch = 0, row_bytes = 0, d_row = 0, frame_bytes = 0, rec = 0;
    uint32_t max_delay_us = 0;

    // double-buffered pinned + device staging: set s is filled while set s^1 is in flight
    uint8_t* h_in[2] = {nullptr, nullptr};
    uint8_t* h_exact[2] = {nullptr, nullptr};
    uint8_t* h_out[2] = {nullptr, nullptr};
    int32_t* h_status[2] = {nullptr, nullptr};
    uint8_t* d_in = nullptr;
    uint8_t* d_exact = nullptr;
    uint8_t* d_out = nullptr;
    int32_t* d_status = nullptr;
    hipStream_t stream = nullptr;

    std::mutex mu;
    std::condition_variable cv_work, cv_room;
    std::condition_variable cv_done[2];   // per set: a flush wakes only its own submitters
    struct Set {
        size_t pending = 0;       // slots handed out
        size_t copied = 0;        // of which the submitter has finished its memcpy into pinned memory
        size_t readers = 0;       // submitters of the LAST flushed generation that still have to copy out
        uint64_t gen_fill = 1;    // generation being filled
        uint64_t gen_done = 0;    // last flushed generation
        int rc = 0;
        std::chrono::steady_clock::time_point first_arrival;
    } sets[2];
    int fill = 0;                 // set currently accepting submissions
    bool stop = false;
    std::thread worker;
    uint64_t batches = 0, items = 0;
};

namespace {

void worker_loop(ucfp_image_batcher* b) {
    (void)hipSetDevice(b->device);
    std::unique_lock<std::mutex> lk(b->mu);
    for (;;) {
        b->cv_work.wait(lk, [&] { return b->stop || b->sets[b->fill].pending > 0; });
        const int s = b->fill;
        ucfp_image_batcher::Set& S = b->sets[s];
        if (b->stop && S.pending == 0) return;
        // let the batch fill up, but no longer than max_delay_us after its first item
        const auto deadline = S.first_arrival + std::chrono::microseconds(b->max_delay_us);
        b->cv_work.wait_until(lk, deadline, [&] { return b->stop || S.pending >= b->max_batch; });
        // close the set FIRST (later submitters go to the other one), then wait for the memcpys of
        // the slots already handed out and for the previous generation's results to be picked up
        b->fill = s ^ 1;
        const size_t n = S.pending;
        b->cv_room.notify_all();
        b->cv_work.wait(lk, [&] { return S.copied == n && S.readers == 0; });
        lk.unlock();

        int rc = UCFP_OK;
        hipError_t e = hipMemcpyAsync(b->d_in, b->h_in[s], n * b->frame_bytes, hipMemcpyHostToDevice, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(b->d_exact, b->h_exact[s], n * 32, hipMemcpyHostToDevice, b->stream);
        if (e == hipSuccess)
            e = (hipError_t)ucfp::image_hash_ordered(b->ctx, b->algo, b->d_in, n, b->width, b->height, b->d_row,
                                                     b->frame_bytes, b->pixfmt, b->pre.min_dimension,
                                                     b->pre.max_dimension, b->d_exact, b->d_out, b->d_status, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(b->h_out[s], b->d_out, n * b->rec, hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(b->h_status[s], b->d_status, n * 4, hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e != hipSuccess) rc = UCFP_E_INDEX;

        lk.lock();
        S.rc = rc;
        S.readers = n;
        S.gen_done = S.gen_fill;
        S.gen_fill++;
        S.pending = 0;
        S.copied = 0;
        b->batches++;
        b->items += n;
        b->cv_done[s].notify_all();
        b->cv_room.notify_all();
    }
}

}  // namespace

extern "C" {

int ucfp_image_batcher_create(ucfp_ctx* ctx, uint32_t algo, uint32_t width, uint32_t height, int pixfmt,
                              const ucfp_image_preprocess* pre, size_t max_batch, uint32_t max_delay_us,
                              ucfp_image_batcher** out) {
    if (!ctx || !out) return capi_fail(UCFP_E_INVALID, "ctx/out is NULL");
    *out = nullptr;
    const size_t rec = ucfp_image_record_bytes(algo);
    if (!rec) return capi_fail(UCFP_E_UNSUPPORTED, "image algo mask %u", algo);
    if (pixfmt < UCFP_PIX_GRAY8 || pixfmt > UCFP_PIX_RGBA8) return capi_fail(UCFP_E_INVALID, "unknown pixfmt %d", pixfmt);
    if (!width || !height || max_batch == 0 || max_batch > 65536)
        return capi_fail(UCFP_E_INVALID, "batcher needs a geometry and 1 <= max_batch <= 65536");
    ucfp_image_batcher* b = new (std::nothrow) ucfp_image_batcher();
    if (!b) return capi_fail(UCFP_E_INDEX, "out of host memory");
    const size_t bpp = pixfmt == UCFP_PIX_GRAY8 ? 1 : pixfmt == UCFP_PIX_RGB8 ? 3 : 4;
    b->ctx = ctx;
    b->device = ucfp::ctx_device(ctx);
    b->algo = algo;
    b->width = width;
    b->height = height;
    b->pixfmt = pixfmt;
    if (pre) b->pre = *pre;
    b->max_batch = max_batch;
    b->max_delay_us = max_delay_us;
    b->row_bytes = (size_t)width * bpp;
    b->d_row = (b->row_bytes + 15) & ~(size_t)15;
    b->frame_bytes = b->d_row * height;
    b->rec = rec;
    hipError_t e = hipSetDevice(b->device);
    for (int s = 0; s < 2 && e == hipSuccess; s++) {
        e = hipHostMalloc((void**)&b->h_in[s], max_batch * b->frame_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&b->h_exact[s], max_batch * 32, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&b->h_out[s], max_batch * rec, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&b->h_status[s], max_batch * 4, hipHostMallocDefault);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_in, max_batch * b->frame_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_exact, max_batch * 32);
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_out, max_batch * rec);
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_status, max_batch * 4);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        ucfp_image_batcher_destroy(b);
        return capi_fail(UCFP_E_INDEX, "batcher allocation failed: %s", hipGetErrorString(e));
    }
    b->worker = std::thread(worker_loop, b);
    *out = b;
    return UCFP_OK;
}

void ucfp_image_batcher_destroy(ucfp_image_batcher* b) {
    if (!b) return;
    if (b->worker.joinable()) {
        {
            std::lock_guard<std::mutex> lk(b->mu);
            b->stop = true;
        }
        b->cv_work.notify_all();
        b->worker.join();
    }
    (void)hipSetDevice(b->device);
    for (int s = 0; s < 2; s++) {
        if (b->h_in[s]) (void)hipHostFree(b->h_in[s]);
        if (b->h_exact[s]) (void)hipHostFree(b->h_exact[s]);
        if (b->h_out[s]) (void)hipHostFree(b->h_out[s]);
        if (b->h_status[s]) (void)hipHostFree(b->h_status[s]);
    }
    if (b->d_in) (void)hipFree(b->d_in);
    if (b->d_exact) (void)hipFree(b->d_exact);
    if (b->d_out) (void)hipFree(b->d_out);
    if (b->d_status) (void)hipFree(b->d_status);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int ucfp_image_batcher_submit(ucfp_image_batcher* b, const uint8_t* frame, size_t row_stride, const uint8_t* exact,
                              uint8_t* out, int32_t* status) {
    if (!b || !frame || !out) return capi_fail(UCFP_E_INVALID, "batcher/frame/out is NULL");
    if (row_stride < b->row_bytes) return capi_fail(UCFP_E_INVALID, "row_stride %zu < width*bpp", row_stride);
    std::unique_lock<std::mutex> lk(b->mu);
    if (b->stop) return capi_fail(UCFP_E_INDEX, "batcher is shutting down");
    // room in the set being filled? (a full set stays closed until the worker has flushed it)
    b->cv_room.wait(lk, [&] { return b->stop || b->sets[b->fill].pending < b->max_batch; });
    if (b->stop) return capi_fail(UCFP_E_INDEX, "batcher is shutting down");
    const int s = b->fill;
    ucfp_image_batcher::Set& S = b->sets[s];
    const size_t slot = S.pending++;
    const uint64_t my_gen = S.gen_fill;
    if (slot == 0) S.first_arrival = std::chrono::steady_clock::now();
    if (slot == 0 || S.pending >= b->max_batch) b->cv_work.notify_one();
    lk.unlock();

    // copy into the pinned slot outside the lock: request threads share the memcpy work
    uint8_t* dst = b->h_in[s] + slot * b->frame_bytes;
    if (row_stride == b->d_row) {
        memcpy(dst, frame, b->frame_bytes);
    } else {
        for (uint32_t y = 0; y < b->height; y++)
            memcpy(dst + (size_t)y * b->d_row, frame + (size_t)y * row_stride, b->row_bytes);
    }
    if (exact) memcpy(b->h_exact[s] + slot * 32, exact, 32);
    else memset(b->h_exact[s] + slot * 32, 0, 32);

    lk.lock();
    S.copied++;
    b->cv_work.notify_one();
    b->cv_done[s].wait(lk, [&] { return S.gen_done >= my_gen; });
    const int rc = S.rc;
    lk.unlock();
    // the worker does not flush this set again before `readers` drops to zero, so the pinned result
    // slot is stable while we copy it out
    if (rc == UCFP_OK) {
        memcpy(out, b->h_out[s] + slot * b->rec, b->rec);
        if (status) *status = b->h_status[s][slot];
    }
    lk.lock();
    S.readers--;
    if (S.readers == 0) b->cv_work.notify_one();
    lk.unlock();
    if (rc != UCFP_OK) return capi_fail(rc, "batched launch failed");
    return UCFP_OK;
}

int ucfp_image_batcher_stats(ucfp_image_batcher* b, uint64_t* batches, uint64_t* items) {
    if (!b) return capi_fail(UCFP_E_INVALID, "batcher is NULL");
    std::lock_guard<std::mutex> lk(b->mu);
    if (batches) *batches = b->batches;
    if (items) *items = b->items;
    return UCFP_OK;
}

}  // extern "C"
