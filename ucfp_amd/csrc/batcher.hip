// batcher.hip -- host micro-batcher in front of the image entry point.
//
// The reference hashes one item per HTTP request on a tokio worker (src/server/handlers.rs:249-252),
// up to 512 requests in flight (src/bin/ucfp.rs:267).  A GPU wants batches.  This object lets every
// request thread call a BLOCKING submit() with its own frame; a worker thread coalesces whatever is
// pending -- up to `max_batch` frames, or whatever arrived within `max_delay_us` of the first one --
// into ONE pinned-memory H2D copy, ONE kernel launch and ONE D2H copy, then wakes the submitters.
// Each submitter copies its frame into its pinned slot itself, so the host-side memcpy work is
// spread over the request threads.  Fixed geometry per batcher; other geometries use the direct call.
// The claim / commit / wake protocol is batch_core.h's (no mutex on the request path).

#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "../../include/ucfp_hip.h"
#include "batch_core.h"
#include "common.h"

namespace ucfp {
int capi_fail(int code, const char* fmt, ...);
int ctx_device(const ucfp_ctx* ctx);
int image_hash_ordered(ucfp_ctx* ctx, uint32_t algo, const uint8_t* frames, size_t n, uint32_t w, uint32_t h,
                       size_t row_stride, size_t frame_stride, int pixfmt, uint32_t min_dim, uint32_t max_dim,
                       const uint8_t* exact, uint8_t* out, int32_t* status, hipStream_t stream);
}  // namespace ucfp
using ucfp::capi_fail;

// Staging of one set.  Input: [max_batch x 32 exact-hash bytes | n frames], one H2D copy.
// Result: [max_batch status words | n records], one D2H copy.
struct ucfp_image_batcher {
    ucfp_ctx* ctx = nullptr;
    int device = 0;
    uint32_t algo = 0, width = 0, height = 0;
    int pixfmt = 0;
    ucfp_image_preprocess pre{8192, 32};
    size_t max_batch = 0, row_bytes = 0, d_row = 0, frame_bytes = 0, rec = 0, in_head = 0, out_head = 0;

    uint8_t* h_in[2] = {nullptr, nullptr};     // double-buffered pinned staging: set s fills while set s^1 is in flight
    uint8_t* h_out[2] = {nullptr, nullptr};
    uint8_t* d_in = nullptr;
    uint8_t* d_out = nullptr;
    hipStream_t stream = nullptr;
    ucfp::BatchCore core;
};

namespace {

int run_set(ucfp_image_batcher* b, int s, size_t n) {
    (void)hipSetDevice(b->device);
    hipError_t e = hipMemcpyAsync(b->d_in, b->h_in[s], b->in_head + n * b->frame_bytes, hipMemcpyHostToDevice, b->stream);
    if (e != hipSuccess) return UCFP_E_INDEX;
    const int rc = ucfp::image_hash_ordered(b->ctx, b->algo, b->d_in + b->in_head, n, b->width, b->height, b->d_row,
                                            b->frame_bytes, b->pixfmt, b->pre.min_dimension, b->pre.max_dimension, b->d_in,
                                            b->d_out + b->out_head, reinterpret_cast<int32_t*>(b->d_out), b->stream);
    if (rc) return UCFP_E_INDEX;
    e = hipMemcpyAsync(b->h_out[s], b->d_out, b->out_head + n * b->rec, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    return e == hipSuccess ? UCFP_OK : UCFP_E_INDEX;
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
    b->row_bytes = (size_t)width * bpp;
    b->d_row = (b->row_bytes + 15) & ~(size_t)15;
    b->frame_bytes = b->d_row * height;
    b->rec = rec;
    b->in_head = (max_batch * 32 + 255) & ~(size_t)255;
    b->out_head = (max_batch * 4 + 255) & ~(size_t)255;
    const size_t in_bytes = b->in_head + max_batch * b->frame_bytes, out_bytes = b->out_head + max_batch * rec;
    hipError_t e = hipSetDevice(b->device);
    for (int s = 0; s < 2 && e == hipSuccess; s++) {
        e = hipHostMalloc((void**)&b->h_in[s], in_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&b->h_out[s], out_bytes, hipHostMallocDefault);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_in, in_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_out, out_bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        ucfp_image_batcher_destroy(b);
        return capi_fail(UCFP_E_INDEX, "batcher allocation failed: %s", hipGetErrorString(e));
    }
    // one payload unit = one frame
    b->core.start(max_batch, max_batch, max_delay_us, [b](int s, size_t n, size_t) { return run_set(b, s, n); });
    *out = b;
    return UCFP_OK;
}

void ucfp_image_batcher_destroy(ucfp_image_batcher* b) {
    if (!b) return;
    b->core.stop();
    (void)hipSetDevice(b->device);
    for (int s = 0; s < 2; s++) {
        if (b->h_in[s]) (void)hipHostFree(b->h_in[s]);
        if (b->h_out[s]) (void)hipHostFree(b->h_out[s]);
    }
    if (b->d_in) (void)hipFree(b->d_in);
    if (b->d_out) (void)hipFree(b->d_out);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int ucfp_image_batcher_submit(ucfp_image_batcher* b, const uint8_t* frame, size_t row_stride, const uint8_t* exact,
                              uint8_t* out, int32_t* status) {
    if (!b || !frame || !out) return capi_fail(UCFP_E_INVALID, "batcher/frame/out is NULL");
    if (row_stride < b->row_bytes) return capi_fail(UCFP_E_INVALID, "row_stride %zu < width*bpp", row_stride);
    ucfp::BatchCore::Ticket t;
    if (!b->core.claim(1, &t)) return capi_fail(UCFP_E_INDEX, "batcher is shutting down");
    // copy into the pinned slot: request threads share the memcpy work
    uint8_t* dst = b->h_in[t.set] + b->in_head + t.slot * b->frame_bytes;
    if (row_stride == b->d_row) {
        memcpy(dst, frame, b->frame_bytes);
    } else {
        for (uint32_t y = 0; y < b->height; y++)
            memcpy(dst + (size_t)y * b->d_row, frame + (size_t)y * row_stride, b->row_bytes);
    }
    if (exact) memcpy(b->h_in[t.set] + t.slot * 32, exact, 32);
    else memset(b->h_in[t.set] + t.slot * 32, 0, 32);
    b->core.commit(t);
    const int rc = b->core.wait(t);
    // the worker does not flush this set again before every reader has released it: the pinned result slot is stable
    if (rc == UCFP_OK) {
        memcpy(out, b->h_out[t.set] + b->out_head + t.slot * b->rec, b->rec);
        if (status) *status = reinterpret_cast<const int32_t*>(b->h_out[t.set])[t.slot];
    }
    b->core.release(t);
    if (rc != UCFP_OK) return capi_fail(rc, "batched launch failed");
    return UCFP_OK;
}

int ucfp_image_batcher_stats(ucfp_image_batcher* b, uint64_t* batches, uint64_t* items) {
    if (!b) return capi_fail(UCFP_E_INVALID, "batcher is NULL");
    b->core.stats(batches, items);
    return UCFP_OK;
}

}  // extern "C"
