// capi.hip -- extern "C" boundary (include/ucfp_hip.h) over the HIP launchers.
// No torch, no C++ types in signatures. No CPU fallback: every entry point needs a HIP device.

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/ucfp_hip.h"
#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

namespace ucfp {
// shared by every translation unit that implements part of the C ABI
int capi_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace ucfp

#include "ctx.h"
#include "upload_probe.h"

using ucfp::grow;

namespace {
#define fail ucfp::capi_fail

constexpr size_t kNormWsFrames = 2048;  // generic-geometry normalised planes held at once (128 MiB: inside the 256 MB
                                       // Infinity Cache, and enough frames per launch to fill the chip with small frames)

}  // namespace

namespace ucfp {
int ctx_device(const ucfp_ctx* ctx) { return ctx->device; }
int image_any_geometry_ready(ucfp_ctx* ctx, uint32_t w) {
    if (w > 2048) return (int)hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(ctx->geo_mu);
    if (ctx->geo_have[w >> 6] >> (w & 63) & 1) return 0;
    const size_t tb = image_any_geometry_bytes();
    if (!ctx->geo) {
        hipError_t e = hipMalloc((void**)&ctx->geo, 2049 * tb);
        if (e != hipSuccess) return (int)e;
    }
    std::vector<uint32_t> tab(tb / 4);
    image_any_geometry_table(w, tab.data());
    hipError_t e = hipMemcpy(reinterpret_cast<uint8_t*>(ctx->geo) + (size_t)w * tb, tab.data(), tb, hipMemcpyHostToDevice);
    if (e != hipSuccess) return (int)e;
    ctx->geo_have[w >> 6] |= 1ull << (w & 63);
    return 0;
}
// Every image launch of the library goes through here.  The fused kernels touch no shared state; a geometry that
// normalises into ctx->norm_ws first waits (on `stream`) for the previous user of that workspace -- whatever stream
// it ran on -- and leaves its own completion event behind.  The caller has already selected ctx->device.
int image_hash_ordered(ucfp_ctx* ctx, uint32_t algo, const uint8_t* frames, size_t n, uint32_t w, uint32_t h,
                       size_t row_stride, size_t frame_stride, int pixfmt, uint32_t min_dim, uint32_t max_dim,
                       const uint8_t* exact, uint8_t* out, int32_t* status, hipStream_t stream) {
    if (!image_hash_needs_ws(frames, w, h, row_stride, frame_stride, pixfmt, min_dim, max_dim)) {
        launch_image_hash(algo, frames, n, w, h, row_stride, frame_stride, pixfmt, min_dim, max_dim, exact, out, status,
                          ctx->norm_ws, kNormWsFrames, stream);
        return (int)hipGetLastError();
    }
    // a geometry the square kernels do not take: up to any_max_pixels per frame the fused any-geometry kernel (one
    // workgroup per frame, no workspace); larger frames are normalised by many waves each into the shared workspace
    uint32_t cls = 0, magic = 0, shift = 0;
    if (n && (size_t)w * h <= ctx->any_max_pixels && image_any_plan(frames, 0, w, h, row_stride, pixfmt, &cls, &magic, &shift)) {
        const size_t bpp = pixfmt == 0 ? 1 : pixfmt == 1 ? 3 : 4;
        const uint8_t* hi = frames + (n - 1) * frame_stride + (size_t)(h - 1) * row_stride + (size_t)w * bpp;
        const int ge = image_any_geometry_ready(ctx, w);
        if (ge) return ge;
        launch_image_hash_any(algo, frames, nullptr, n, image_any_group(cls), w, h, (uint32_t)row_stride, cls, magic, shift, frame_stride,
                              frames, hi, exact, out, status, ctx->geo, stream);
        return (int)hipGetLastError();
    }
    std::lock_guard<std::mutex> lk(ctx->norm_mu);
    hipError_t e = hipStreamWaitEvent(stream, ctx->norm_done, 0);
    if (e != hipSuccess) return (int)e;
    launch_image_hash(algo, frames, n, w, h, row_stride, frame_stride, pixfmt, min_dim, max_dim, exact, out, status,
                      ctx->norm_ws, kNormWsFrames, stream);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return (int)hipEventRecord(ctx->norm_done, stream);
}

}  // namespace ucfp

extern "C" {

int ucfp_abi_version(void) { return UCFP_ABI_VERSION; }

const char* ucfp_last_error(void) { return g_err; }

int ucfp_ctx_create(int device_id, ucfp_ctx** out) {
    if (!out) return fail(UCFP_E_INVALID, "ucfp_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(UCFP_E_INDEX, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= count)
        return fail(UCFP_E_INVALID, "device %d out of range [0,%d)", device_id, count);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(UCFP_E_UNSUPPORTED, "device %d is %s; kernels are built for gfx950 only",
                    device_id, prop.gcnArchName);
    ucfp_ctx* c = new (std::nothrow) ucfp_ctx();
    if (!c) return fail(UCFP_E_INDEX, "out of host memory");
    c->device = device_id;
    hipError_t e2 = hipMalloc((void**)&c->norm_ws, kNormWsFrames * 65536);
    if (e2 == hipSuccess) e2 = hipStreamCreateWithFlags(&c->host_stream, hipStreamNonBlocking);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&c->audio_done, hipEventDisableTiming);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&c->png_done, hipEventDisableTiming);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&c->norm_done, hipEventDisableTiming);
    if (e2 == hipSuccess) e2 = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&c->side_fork, hipEventDisableTiming);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&c->side_join, hipEventDisableTiming);
    for (int i = 0; i < 2 && e2 == hipSuccess; i++) e2 = hipEventCreateWithFlags(&c->item_used[i], hipEventDisableTiming);
    if (const char* v = getenv("UCFP_IMAGE_ANY_MAX_PIXELS")) c->any_max_pixels = (size_t)strtoull(v, nullptr, 10);   // (tuning)
    if (e2 != hipSuccess) {
        ucfp_ctx_destroy(c);
        return fail(UCFP_E_INDEX, "context allocation failed: %s", hipGetErrorString(e2));
    }
    *out = c;
    return UCFP_OK;
}

void ucfp_ctx_destroy(ucfp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->norm_ws) (void)hipFree(c->norm_ws);
    if (c->stage_in) (void)hipFree(c->stage_in);
    if (c->stage_out) (void)hipFree(c->stage_out);
    if (c->host_stream) (void)hipStreamDestroy(c->host_stream);
    if (c->audio_ws) (void)hipFree(c->audio_ws);
    if (c->audio_done) (void)hipEventDestroy(c->audio_done);
    if (c->png_done) (void)hipEventDestroy(c->png_done);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->side_fork) (void)hipEventDestroy(c->side_fork);
    if (c->side_join) (void)hipEventDestroy(c->side_join);
    if (c->png_ws) (void)hipFree(c->png_ws);
    if (c->b3_ws) (void)hipFree(c->b3_ws);
    if (c->norm_done) (void)hipEventDestroy(c->norm_done);
    if (c->geo) (void)hipFree(c->geo);
    for (int i = 0; i < 2; i++) {
        if (c->item_h[i]) (void)hipHostFree(c->item_h[i]);
        if (c->item_d[i]) (void)hipFree(c->item_d[i]);
        if (c->item_used[i]) (void)hipEventDestroy(c->item_used[i]);
    }
    delete c;
}

size_t ucfp_image_record_bytes(uint32_t algo) {
    if (algo == UCFP_IMG_MULTI) return UCFP_IMAGE_MULTI_BYTES;
    if (algo == UCFP_IMG_AHASH || algo == UCFP_IMG_PHASH || algo == UCFP_IMG_DHASH)
        return UCFP_IMAGE_FP_BYTES;
    return 0;
}

static int image_check(ucfp_ctx* ctx, uint32_t algo, const void* frames, size_t n, uint32_t w,
                       uint32_t h, size_t row_stride, size_t frame_stride, int pixfmt,
                       const void* out) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    if (ucfp_image_record_bytes(algo) == 0)
        return fail(UCFP_E_UNSUPPORTED, "image algo mask %u is not one of ahash|phash|dhash|multi", algo);
    if (pixfmt < UCFP_PIX_GRAY8 || pixfmt > UCFP_PIX_RGBA8)
        return fail(UCFP_E_INVALID, "unknown pixfmt %d", pixfmt);
    if (n && (!frames || !out)) return fail(UCFP_E_INVALID, "frames/out is NULL");
    const size_t bpp = pixfmt == UCFP_PIX_GRAY8 ? 1 : pixfmt == UCFP_PIX_RGB8 ? 3 : 4;
    if (w == 0 || h == 0) return fail(UCFP_E_MODALITY, "empty image %ux%u", w, h);
    if (row_stride < (size_t)w * bpp) return fail(UCFP_E_INVALID, "row_stride %zu < width*bpp", row_stride);
    if (n > 1 && frame_stride < row_stride * (size_t)(h - 1) + (size_t)w * bpp)
        return fail(UCFP_E_INVALID, "frame_stride %zu overlaps frames", frame_stride);
    if (n > 0x7fffffffu) return fail(UCFP_E_INVALID, "batch of %zu frames exceeds one launch", n);
    return UCFP_OK;
}

int ucfp_image_hash_batch_dev(ucfp_ctx* ctx, uint32_t algo, const uint8_t* frames, size_t n,
                              uint32_t width, uint32_t height, size_t row_stride,
                              size_t frame_stride, int pixfmt, const ucfp_image_preprocess* pre,
                              const uint8_t* exact, uint8_t* out, int32_t* status, void* stream) {
    int rc = image_check(ctx, algo, frames, n, width, height, row_stride, frame_stride, pixfmt, out);
    if (rc) return rc;
    const uint32_t min_dim = pre ? pre->min_dimension : 32u;
    const uint32_t max_dim = pre ? pre->max_dimension : 8192u;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY((hipError_t)ucfp::image_hash_ordered(ctx, algo, frames, n, width, height, row_stride, frame_stride, pixfmt,
                                                 min_dim, max_dim, exact, out, status, (hipStream_t)stream));
    return UCFP_OK;
}

// ---------------------------------- BLAKE3 on the device ----------------------------------------

int ucfp_blake3_batch_dev(ucfp_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n, size_t blob_bytes,
                          uint8_t* d_out, void* stream) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    if (n == 0) return UCFP_OK;
    if (!d_offsets || !d_out || (blob_bytes && !d_blob)) return fail(UCFP_E_INVALID, "NULL buffer");
    if (n > 0x7fffffffu) return fail(UCFP_E_INVALID, "batch of %zu inputs exceeds one launch", n);
    if ((uintptr_t)d_out & 3u) return fail(UCFP_E_INVALID, "the digest buffer must be 4-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = grow(&ctx->b3_ws, &ctx->b3_ws_cap, ucfp::blake3_ws_bytes(n, blob_bytes));
    if (rc) return rc;
    HIP_TRY(hipStreamWaitEvent(st, ctx->png_done, 0));
    ucfp::launch_blake3_batch(d_blob, d_offsets, n, ctx->b3_ws, d_out, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->png_done, st));
    return UCFP_OK;
}

// ---------------------------------- PNG front end ----------------------------------------------

int ucfp_png_probe(const uint8_t* png, size_t len, uint32_t* width, uint32_t* height, int* pixfmt) {
    if (!png || !width || !height || !pixfmt) return fail(UCFP_E_INVALID, "NULL argument");
    const int rc = ucfp::png_probe_bytes(png, len, width, height, pixfmt);
    if (rc < 0) return fail(rc, "not a PNG file (or a damaged IHDR)");
    return rc;
}

static int png_check(ucfp_ctx* ctx, const void* d_png, const void* d_offsets, size_t n, size_t png_bytes, uint32_t w, uint32_t h,
                     int pixfmt) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    if (n && (!d_png || !d_offsets)) return fail(UCFP_E_INVALID, "png/offsets is NULL");
    if (pixfmt < UCFP_PIX_GRAY8 || pixfmt > UCFP_PIX_RGBA8) return fail(UCFP_E_INVALID, "unknown pixfmt %d", pixfmt);
    if (w == 0 || h == 0 || w > 16384 || h > 16384) return fail(UCFP_E_MODALITY, "PNG geometry %ux%u outside 1 .. 16384", w, h);
    if (n > 0x7fffffffu || png_bytes >= ((size_t)1 << 32)) return fail(UCFP_E_INVALID, "PNG batch too large for one call");
    const size_t bpp = pixfmt == UCFP_PIX_GRAY8 ? 1 : pixfmt == UCFP_PIX_RGB8 ? 3 : 4;
    if ((size_t)h * ((size_t)w * bpp + 1) >= ((size_t)1 << 31)) return fail(UCFP_E_INVALID, "PNG frame too large for the device decoder");
    if ((size_t)w * bpp > 60000) return fail(UCFP_E_INVALID, "PNG rows of %zu bytes exceed the decoder's row buffer", (size_t)w * bpp);
    return UCFP_OK;
}

// Decode into caller frames (frames != NULL) or into the workspace's own frame area; returns the layout used.
static int png_decode_impl(ucfp_ctx* ctx, const uint8_t* d_png, const uint64_t* d_offsets, size_t n, size_t png_bytes,
                           uint32_t w, uint32_t h, int pixfmt, uint8_t* frames, size_t row_stride, size_t frame_stride,
                           int32_t* d_status, hipStream_t st, ucfp::PngWs* layout, uint8_t** own_frames) {
    // callers hold ctx->mu
    ucfp::PngWs l;
    size_t need = ucfp::png_ws_bytes(n, png_bytes, w, h, pixfmt, &l);
    const size_t frames_off = need;
    if (!frames) need += n * frame_stride;
    int rc = grow(&ctx->png_ws, &ctx->png_ws_cap, need);
    if (rc) return rc;
    HIP_TRY(hipStreamWaitEvent(st, ctx->png_done, 0));
    uint8_t* fr = frames ? frames : ctx->png_ws + frames_off;
    ucfp::launch_png_decode(d_png, d_offsets, n, w, h, pixfmt, ctx->png_ws, l, fr, row_stride, frame_stride, d_status, st, ctx->side,
                            ctx->side_fork, ctx->side_join);
    HIP_TRY(hipGetLastError());
    if (layout) *layout = l;
    if (own_frames) *own_frames = fr;
    return UCFP_OK;
}

int ucfp_image_png_decode_batch_dev(ucfp_ctx* ctx, const uint8_t* d_png, const uint64_t* d_offsets, size_t n,
                                    size_t png_bytes, uint32_t width, uint32_t height, int pixfmt, uint8_t* d_frames,
                                    size_t row_stride, size_t frame_stride, int32_t* d_status, void* stream) {
    int rc = png_check(ctx, d_png, d_offsets, n, png_bytes, width, height, pixfmt);
    if (rc) return rc;
    if (n == 0) return UCFP_OK;
    const size_t bpp = pixfmt == UCFP_PIX_GRAY8 ? 1 : pixfmt == UCFP_PIX_RGB8 ? 3 : 4;
    if (!d_frames) return fail(UCFP_E_INVALID, "frames is NULL");
    if (row_stride < (size_t)width * bpp || (n > 1 && frame_stride < row_stride * (size_t)(height - 1) + (size_t)width * bpp))
        return fail(UCFP_E_INVALID, "row_stride / frame_stride too small for %ux%u", width, height);
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    rc = png_decode_impl(ctx, d_png, d_offsets, n, png_bytes, width, height, pixfmt, d_frames, row_stride, frame_stride,
                         d_status, (hipStream_t)stream, nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->png_done, (hipStream_t)stream));
    return UCFP_OK;
}

int ucfp_image_png_hash_batch_dev(ucfp_ctx* ctx, uint32_t algo, const uint8_t* d_png, const uint64_t* d_offsets, size_t n,
                                  size_t png_bytes, uint32_t width, uint32_t height, int pixfmt,
                                  const ucfp_image_preprocess* pre, const uint8_t* d_exact, uint8_t* d_out,
                                  int32_t* d_status, void* stream) {
    int rc = png_check(ctx, d_png, d_offsets, n, png_bytes, width, height, pixfmt);
    if (rc) return rc;
    const size_t rec = ucfp_image_record_bytes(algo);
    if (!rec) return fail(UCFP_E_UNSUPPORTED, "image algo mask %u is not one of ahash|phash|dhash|multi", algo);
    if (n == 0) return UCFP_OK;
    if (!d_out) return fail(UCFP_E_INVALID, "out is NULL");
    const size_t bpp = pixfmt == UCFP_PIX_GRAY8 ? 1 : pixfmt == UCFP_PIX_RGB8 ? 3 : 4;
    const size_t row = ((size_t)width * bpp + 15) & ~(size_t)15, frame = row * height;   // 16-byte rows: the fused hash path
    const uint32_t min_dim = pre ? pre->min_dimension : 32u, max_dim = pre ? pre->max_dimension : 8192u;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    ucfp::PngWs l;
    uint8_t* fr = nullptr;
    rc = png_decode_impl(ctx, d_png, d_offsets, n, png_bytes, width, height, pixfmt, nullptr, row, frame, nullptr, st, &l, &fr);
    if (rc) return rc;
    if (!d_exact) {
        // the files are here: their BLAKE3 (the records' `exact` field, image.rs:82) is computed on the device too
        const size_t cvb = (ucfp::blake3_ws_bytes(n, png_bytes) + 255) & ~(size_t)255;
        rc = grow(&ctx->b3_ws, &ctx->b3_ws_cap, cvb + n * 32);
        if (rc) return rc;
        ucfp::launch_blake3_batch(d_png, d_offsets, n, ctx->b3_ws, ctx->b3_ws + cvb, st);
        d_exact = ctx->b3_ws + cvb;
    }
    HIP_TRY((hipError_t)ucfp::image_hash_ordered(ctx, algo, fr, n, width, height, row, frame, pixfmt, min_dim, max_dim, d_exact,
                                                 d_out, d_status, st));
    ucfp::launch_png_merge_status(ctx->png_ws, l, n, d_out, (uint32_t)rec, d_status, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->png_done, st));
    return UCFP_OK;
}

// ---------------------------------- JPEG front end ------------------------------------
int ucfp_jpeg_probe(const uint8_t* jpg, size_t len, uint32_t* width, uint32_t* height) {
    if (!jpg || !width || !height) return fail(UCFP_E_INVALID, "NULL argument");
    const int rc = ucfp::jpeg_probe_bytes(jpg, len, width, height);
    if (rc < 0) return fail(rc, "not a JPEG file");
    return rc;
}

static int jpeg_check(ucfp_ctx* ctx, const void* d_jpg, const void* d_offsets, size_t n, size_t jpg_bytes, uint32_t w, uint32_t h) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    if (n && (!d_jpg || !d_offsets)) return fail(UCFP_E_INVALID, "jpg/offsets is NULL");
    if (w == 0 || h == 0 || w > 16384 || h > 16384) return fail(UCFP_E_MODALITY, "JPEG geometry %ux%u outside 1 .. 16384", w, h);
    if (n > 0x7fffffffu || jpg_bytes >= ((size_t)1 << 32)) return fail(UCFP_E_INVALID, "JPEG batch too large for one call");
    return UCFP_OK;
}

// Decode into caller frames (frames != NULL) or into the workspace's own frame area (shares the PNG front end's workspace).
static int jpeg_decode_impl(ucfp_ctx* ctx, const uint8_t* d_jpg, const uint64_t* d_offsets, size_t n, size_t jpg_bytes, uint32_t w,
                            uint32_t h, uint8_t* frames, size_t row_stride, size_t frame_stride, int32_t* d_status,
                            hipStream_t st, ucfp::JpegWs* layout, uint8_t** own_frames) {
    ucfp::JpegWs l;
    size_t need = ucfp::jpeg_ws_bytes(n, jpg_bytes, w, h, &l);
    const size_t frames_off = need;
    if (!frames) need += n * frame_stride;
    int rc = grow(&ctx->png_ws, &ctx->png_ws_cap, need);
    if (rc) return rc;
    HIP_TRY(hipStreamWaitEvent(st, ctx->png_done, 0));
    uint8_t* fr = frames ? frames : ctx->png_ws + frames_off;
    ucfp::launch_jpeg_decode(d_jpg, d_offsets, n, w, h, ctx->png_ws, l, fr, row_stride, frame_stride, d_status, st);
    HIP_TRY(hipGetLastError());
    if (layout) *layout = l;
    if (own_frames) *own_frames = fr;
    return UCFP_OK;
}

int ucfp_image_jpeg_decode_batch_dev(ucfp_ctx* ctx, const uint8_t* d_jpg, const uint64_t* d_offsets, size_t n, size_t jpg_bytes,
                                     uint32_t width, uint32_t height, uint8_t* d_frames, size_t row_stride, size_t frame_stride,
                                     int32_t* d_status, void* stream) {
    int rc = jpeg_check(ctx, d_jpg, d_offsets, n, jpg_bytes, width, height);
    if (rc) return rc;
    if (n == 0) return UCFP_OK;
    if (!d_frames) return fail(UCFP_E_INVALID, "frames is NULL");
    if (row_stride < (size_t)width || (n > 1 && frame_stride < row_stride * (size_t)(height - 1) + (size_t)width))
        return fail(UCFP_E_INVALID, "row_stride / frame_stride too small for %ux%u", width, height);
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    rc = jpeg_decode_impl(ctx, d_jpg, d_offsets, n, jpg_bytes, width, height, d_frames, row_stride, frame_stride, d_status,
                          (hipStream_t)stream, nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->png_done, (hipStream_t)stream));
    return UCFP_OK;
}

int ucfp_image_jpeg_hash_batch_dev(ucfp_ctx* ctx, uint32_t algo, const uint8_t* d_jpg, const uint64_t* d_offsets, size_t n,
                                   size_t jpg_bytes, uint32_t width, uint32_t height, const ucfp_image_preprocess* pre,
                                   const uint8_t* d_exact, uint8_t* d_out, int32_t* d_status, void* stream) {
    int rc = jpeg_check(ctx, d_jpg, d_offsets, n, jpg_bytes, width, height);
    if (rc) return rc;
    const size_t rec = ucfp_image_record_bytes(algo);
    if (!rec) return fail(UCFP_E_UNSUPPORTED, "image algo mask %u is not one of ahash|phash|dhash|multi", algo);
    if (n == 0) return UCFP_OK;
    if (!d_out) return fail(UCFP_E_INVALID, "out is NULL");
    const size_t row = ((size_t)width + 15) & ~(size_t)15, frame = row * height;   // luma planes, 16-byte rows: the fused hash path
    const uint32_t min_dim = pre ? pre->min_dimension : 32u, max_dim = pre ? pre->max_dimension : 8192u;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    ucfp::JpegWs l;
    uint8_t* fr = nullptr;
    rc = jpeg_decode_impl(ctx, d_jpg, d_offsets, n, jpg_bytes, width, height, nullptr, row, frame, nullptr, st, &l, &fr);
    if (rc) return rc;
    if (!d_exact) {
        const size_t cvb = (ucfp::blake3_ws_bytes(n, jpg_bytes) + 255) & ~(size_t)255;
        rc = grow(&ctx->b3_ws, &ctx->b3_ws_cap, cvb + n * 32);
        if (rc) return rc;
        ucfp::launch_blake3_batch(d_jpg, d_offsets, n, ctx->b3_ws, ctx->b3_ws + cvb, st);
        d_exact = ctx->b3_ws + cvb;
    }
    HIP_TRY((hipError_t)ucfp::image_hash_ordered(ctx, algo, fr, n, width, height, row, frame, UCFP_PIX_GRAY8, min_dim, max_dim,
                                                 d_exact, d_out, d_status, st));
    ucfp::launch_jpeg_merge_status(ctx->png_ws, l, n, d_out, (uint32_t)rec, d_status, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->png_done, st));
    return UCFP_OK;
}

int ucfp_image_hash_batch(ucfp_ctx* ctx, uint32_t algo, const uint8_t* frames, size_t n,
                          uint32_t width, uint32_t height, size_t row_stride, size_t frame_stride,
                          int pixfmt, const ucfp_image_preprocess* pre, const uint8_t* exact,
                          uint8_t* out, int32_t* status) {
    int rc = image_check(ctx, algo, frames, n, width, height, row_stride, frame_stride, pixfmt, out);
    if (rc) return rc;
    if (n == 0) return UCFP_OK;
    const size_t bpp = pixfmt == UCFP_PIX_GRAY8 ? 1 : pixfmt == UCFP_PIX_RGB8 ? 3 : 4;
    const size_t rec = ucfp_image_record_bytes(algo);
    // Stage densely packed and 16-byte aligned so the fused path is taken whenever geometry allows.
    const size_t d_row = (((size_t)width * bpp) + 15) & ~(size_t)15;
    const size_t d_frame = d_row * height;
    const size_t in_bytes = d_frame * n;
    const size_t out_bytes = n * rec + n * 32 + n * sizeof(int32_t);
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    rc = grow(&ctx->stage_in, &ctx->stage_in_cap, in_bytes);
    if (rc) return rc;
    rc = grow(&ctx->stage_out, &ctx->stage_out_cap, out_bytes);
    if (rc) return rc;
    hipStream_t st = ctx->host_stream;
    if (frame_stride == row_stride * height || n == 1) {
        // frames are one tall image: a single strided copy
        HIP_TRY(hipMemcpy2DAsync(ctx->stage_in, d_row, frames, row_stride, (size_t)width * bpp,
                                 (size_t)height * n, hipMemcpyHostToDevice, st));
    } else {
        for (size_t i = 0; i < n; i++)
            HIP_TRY(hipMemcpy2DAsync(ctx->stage_in + i * d_frame, d_row, frames + i * frame_stride,
                                     row_stride, (size_t)width * bpp, height, hipMemcpyHostToDevice, st));
    }
    uint8_t* d_out = ctx->stage_out;
    uint8_t* d_exact = ctx->stage_out + n * rec;
    int32_t* d_status = reinterpret_cast<int32_t*>(ctx->stage_out + n * rec + n * 32);
    if (exact) HIP_TRY(hipMemcpyAsync(d_exact, exact, n * 32, hipMemcpyHostToDevice, st));
    const uint32_t min_dim = pre ? pre->min_dimension : 32u;
    const uint32_t max_dim = pre ? pre->max_dimension : 8192u;
    HIP_TRY((hipError_t)ucfp::image_hash_ordered(ctx, algo, ctx->stage_in, n, width, height, d_row, d_frame, pixfmt,
                                                 min_dim, max_dim, exact ? d_exact : nullptr, d_out, d_status, st));
    HIP_TRY(hipMemcpyAsync(out, d_out, n * rec, hipMemcpyDeviceToHost, st));
    if (status) HIP_TRY(hipMemcpyAsync(status, d_status, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return UCFP_OK;
}

// ---------------------------------- audio ---------------------------------------------

static ucfp_wang_config wang_defaults() { return ucfp_wang_config{10u, 63u, 64u, 30u, -50.0f}; }

static int wang_cfg_check(const ucfp_wang_config& c) {
    if (c.fan_out < 1 || c.fan_out > 64 || c.target_zone_t < 1 || c.target_zone_t > 512 || c.target_zone_f < 1 ||
        c.target_zone_f > 1024 || c.peaks_per_sec < 1 || c.peaks_per_sec > 256)
        return fail(UCFP_E_MODALITY, "WangConfig outside the ranges of /v1/algorithms");
    return UCFP_OK;
}

size_t ucfp_audio_wang_max_hashes(size_t n_samples, const ucfp_wang_config* cfg) {
    const ucfp_wang_config c = cfg ? *cfg : wang_defaults();
    const size_t frames = ucfp::audio_stft_frames(n_samples, 1024, 128);
    if (!frames) return 0;
    const size_t n_sec = ((frames - 1) * 128) / 8000 + 1;
    return n_sec * c.peaks_per_sec * c.fan_out;
}

size_t ucfp_audio_wang_batch_max_hashes(size_t n_total, size_t n_clips, uint32_t sample_rate,
                                        const ucfp_wang_config* cfg) {
    if (!sample_rate || !n_clips) return 0;
    const ucfp_wang_config c = cfg ? *cfg : wang_defaults();
    const ucfp::WangWs w = ucfp::wang_ws_layout(n_total, n_clips, sample_rate, c.peaks_per_sec);
    return (size_t)w.n_sec * c.peaks_per_sec * c.fan_out;
}

// shared by the single-stream and the batch entry point
static int wang_batch_impl(ucfp_ctx* ctx, const float* d_pcm, const uint64_t* d_offsets, size_t n_total, size_t n_clips,
                           uint32_t sample_rate, const ucfp_wang_config* cfg, uint8_t* d_out, size_t cap_hashes,
                           uint64_t* d_out_offsets, uint64_t* d_n_hashes, hipStream_t st) {
    const ucfp_wang_config c = cfg ? *cfg : wang_defaults();
    int rc = wang_cfg_check(c);
    if (rc) return rc;
    if (n_clips > 0x7fffffffu || n_total > ((size_t)1 << 46))
        return fail(UCFP_E_INVALID, "audio batch too large for one call");
    const ucfp::WangWs w = ucfp::wang_ws_layout(n_total, n_clips, sample_rate, c.peaks_per_sec);
    if (n_clips == 1 && w.frames >= ((size_t)1 << 23))
        return fail(UCFP_E_INVALID, "a clip of %zu frames exceeds 2^23 (37 h at 8 kHz): split it", w.frames);
    if (cap_hashes && ((uintptr_t)d_out & 7u)) return fail(UCFP_E_INVALID, "the hash buffer must be 8-byte aligned");
    if (w.n_seg > 0x7fffffffu || (size_t)w.n_sec * c.peaks_per_sec > 0x7fffffffu || w.n_sec > 3000000u)   // byte offsets into the candidate arrays are 32-bit
        return fail(UCFP_E_INVALID, "audio batch too large for one call");
    const float floor_p = (float)(65536.0 * pow(10.0, (double)c.min_anchor_mag_db / 10.0));
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    rc = grow(&ctx->audio_ws, &ctx->audio_ws_cap, w.total);
    if (rc) return rc;
    HIP_TRY(hipStreamWaitEvent(st, ctx->audio_done, 0));
    ucfp::launch_wang_batch(d_pcm, d_offsets, n_total, n_clips, sample_rate, c.fan_out, c.target_zone_t,
                            c.target_zone_f, c.peaks_per_sec, floor_p, ctx->audio_ws, w,
                            reinterpret_cast<uint32_t*>(d_out), cap_hashes, d_out_offsets, d_n_hashes, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->audio_done, st));
    return UCFP_OK;
}

int ucfp_audio_wang_dev(ucfp_ctx* ctx, const float* d_pcm, size_t n, uint32_t sample_rate,
                        const ucfp_wang_config* cfg, uint8_t* d_out, size_t cap_hashes, uint64_t* d_n_hashes,
                        void* stream) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    if (!d_n_hashes || (n && !d_pcm) || (cap_hashes && !d_out)) return fail(UCFP_E_INVALID, "NULL buffer");
    if (sample_rate != 8000)
        return fail(UCFP_E_MODALITY, "Wang requires 8 kHz mono input (got %u Hz); resample upstream", sample_rate);
    return wang_batch_impl(ctx, d_pcm, nullptr, n, 1, sample_rate, cfg, d_out, cap_hashes, nullptr, d_n_hashes,
                           (hipStream_t)stream);
}

int ucfp_audio_wang_batch_dev(ucfp_ctx* ctx, const float* d_pcm, const uint64_t* d_offsets, size_t n_total, size_t n_clips,
                              uint32_t sample_rate, const ucfp_wang_config* cfg, uint8_t* d_out, size_t cap_hashes,
                              uint64_t* d_out_offsets, void* stream) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    if (!d_out_offsets || (n_clips && !d_offsets) || (n_total && !d_pcm) || (cap_hashes && !d_out))
        return fail(UCFP_E_INVALID, "NULL buffer");
    if (sample_rate < 1000 || sample_rate > 384000)
        return fail(UCFP_E_MODALITY, "invalid sample rate %u (1 000 .. 384 000 Hz)", sample_rate);
    return wang_batch_impl(ctx, d_pcm, d_offsets, n_total, n_clips, sample_rate, cfg, d_out, cap_hashes, d_out_offsets,
                           nullptr, (hipStream_t)stream);
}

int ucfp_audio_wang(ucfp_ctx* ctx, const float* pcm, size_t n, uint32_t sample_rate, const ucfp_wang_config* cfg,
                    uint8_t* out, size_t cap_hashes, size_t* n_hashes) {
    if (!ctx || !n_hashes) return fail(UCFP_E_INVALID, "ctx/n_hashes is NULL");
    *n_hashes = 0;
    if (n && !pcm) return fail(UCFP_E_INVALID, "pcm is NULL");
    if (sample_rate != 8000)
        return fail(UCFP_E_MODALITY, "Wang requires 8 kHz mono input (got %u Hz); resample upstream", sample_rate);
    HIP_TRY(hipSetDevice(ctx->device));
    float* d_pcm = nullptr;
    uint8_t* d_out = nullptr;
    uint64_t* d_cnt = nullptr;
    HIP_TRY(hipMalloc((void**)&d_pcm, (n ? n : 1) * 4));
    hipError_t e = hipMalloc((void**)&d_out, (cap_hashes ? cap_hashes : 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_cnt, 8);
    int rc = UCFP_OK;
    uint64_t cnt = 0;
    hipStream_t st = ctx->host_stream;
    if (e == hipSuccess && n) e = hipMemcpyAsync(d_pcm, pcm, n * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        rc = ucfp_audio_wang_dev(ctx, d_pcm, n, sample_rate, cfg, d_out, cap_hashes, d_cnt, st);
        if (rc == UCFP_OK) {
            e = hipMemcpyAsync(&cnt, d_cnt, 8, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess && cnt) {
                const size_t m = cnt < cap_hashes ? (size_t)cnt : cap_hashes;
                e = hipMemcpy(out, d_out, m * 8, hipMemcpyDeviceToHost);
            }
        }
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(d_pcm);
    if (d_out) (void)hipFree(d_out);
    if (d_cnt) (void)hipFree(d_cnt);
    if (rc) return rc;
    if (e != hipSuccess) return fail(UCFP_E_INDEX, "wang failed: %s", hipGetErrorString(e));
    *n_hashes = (size_t)cnt;
    if (cnt > cap_hashes) return fail(UCFP_E_INVALID, "output holds %zu hashes, %llu produced", cap_hashes, (unsigned long long)cnt);
    return UCFP_OK;
}

size_t ucfp_audio_resample_len(size_t n, uint32_t sr_in, uint32_t sr_out) {
    return sr_in ? ucfp::audio_resample_len(n, sr_in, sr_out) : 0;
}

int ucfp_audio_resample_linear_dev(ucfp_ctx* ctx, const float* d_in, size_t n, uint32_t sr_in, uint32_t sr_out,
                                   float* d_out, size_t cap, void* stream) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    if (sr_in == 0 || sr_out == 0 || sr_in > 384000 || sr_out > 384000)
        return fail(UCFP_E_MODALITY, "invalid sample rate %u -> %u", sr_in, sr_out);
    const size_t m = ucfp::audio_resample_len(n, sr_in, sr_out);
    if (m > cap) return fail(UCFP_E_INVALID, "resample output needs %zu samples, buffer holds %zu", m, cap);
    if (m && (!d_in || !d_out)) return fail(UCFP_E_INVALID, "NULL buffer");
    ucfp::launch_resample_linear(d_in, n, sr_in, sr_out, d_out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UCFP_OK;
}

static void haitsma_edges(float fmin, float fmax, uint32_t edges[34]) {
    for (int b = 0; b <= 33; b++) {
        const double f = (double)fmin * pow((double)fmax / (double)fmin, (double)b / 33);
        const double bin = f * 2048 / 5000;
        uint32_t e = (uint32_t)ceil(bin);
        if (e > 1024) e = 1024;
        edges[b] = e;
    }
}

size_t ucfp_audio_haitsma_frames(size_t n_samples, uint32_t sample_rate) {
    if (!sample_rate) return 0;
    const size_t n5 = sample_rate == 5000 ? n_samples : ucfp::audio_resample_len(n_samples, sample_rate, 5000);
    return ucfp::audio_stft_frames(n5, 2048, 64);
}

int ucfp_audio_haitsma_dev(ucfp_ctx* ctx, const float* d_pcm5k, size_t n, const ucfp_haitsma_config* cfg,
                           uint32_t* d_out, size_t cap_frames, void* stream) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    const float fmin = cfg ? cfg->fmin : 300.0f, fmax = cfg ? cfg->fmax : 2000.0f;
    if (!(fmin >= 1.0f) || !(fmax > fmin) || !(fmax <= 2500.0f))
        return fail(UCFP_E_MODALITY, "Haitsma band edges must satisfy 1 <= fmin < fmax <= 2500 Hz");
    const size_t frames = ucfp::audio_stft_frames(n, 2048, 64);
    if (frames > cap_frames) return fail(UCFP_E_INVALID, "output holds %zu frames, %zu produced", cap_frames, frames);
    if (frames == 0) return UCFP_OK;
    if (!d_pcm5k || !d_out) return fail(UCFP_E_INVALID, "NULL buffer");
    uint32_t edges[34];
    haitsma_edges(fmin, fmax, edges);
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = grow(&ctx->audio_ws, &ctx->audio_ws_cap, ucfp::haitsma_ws_bytes(n));
    if (rc) return rc;
    HIP_TRY(hipStreamWaitEvent(st, ctx->audio_done, 0));
    // edges are copied with a pageable-memory async copy: keep them alive in the context
    static thread_local uint32_t tl_edges[34];
    memcpy(tl_edges, edges, sizeof edges);
    ucfp::launch_haitsma(d_pcm5k, n, tl_edges, ctx->audio_ws, d_out, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->audio_done, st));
    return UCFP_OK;
}

size_t ucfp_audio_haitsma_batch_max_frames(size_t n_total, size_t n_clips, uint32_t sample_rate) {
    if (sample_rate < 1000 || sample_rate > 384000) return 0;
    return ucfp::haitsma_batch_ws(n_total, n_clips, sample_rate).frames_ub;
}

int ucfp_audio_haitsma_batch_dev(ucfp_ctx* ctx, const float* d_pcm, const uint64_t* d_offsets, size_t n_total, size_t n_clips,
                                 uint32_t sample_rate, const ucfp_haitsma_config* cfg, uint32_t* d_out, size_t cap_frames,
                                 uint64_t* d_out_offsets, void* stream) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    if (!d_out_offsets || (n_clips && !d_offsets) || (n_total && !d_pcm) || (cap_frames && !d_out))
        return fail(UCFP_E_INVALID, "NULL buffer");
    if (sample_rate < 1000 || sample_rate > 384000)
        return fail(UCFP_E_MODALITY, "invalid sample rate %u (1 000 .. 384 000 Hz)", sample_rate);
    if (n_clips > 0x7fffffffu || n_total > ((size_t)1 << 40)) return fail(UCFP_E_INVALID, "audio batch too large for one call");
    const float fmin = cfg ? cfg->fmin : 300.0f, fmax = cfg ? cfg->fmax : 2000.0f;
    if (!(fmin >= 1.0f) || !(fmax > fmin) || !(fmax <= 2500.0f))
        return fail(UCFP_E_MODALITY, "Haitsma band edges must satisfy 1 <= fmin < fmax <= 2500 Hz");
    uint32_t edges[34];
    haitsma_edges(fmin, fmax, edges);
    const ucfp::HaitsmaBatchWs w = ucfp::haitsma_batch_ws(n_total, n_clips, sample_rate);
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = grow(&ctx->audio_ws, &ctx->audio_ws_cap, w.total);
    if (rc) return rc;
    HIP_TRY(hipStreamWaitEvent(st, ctx->audio_done, 0));
    static thread_local uint32_t tl_edges[34];      // copied with an async copy from pageable memory: must outlive the call
    memcpy(tl_edges, edges, sizeof edges);
    ucfp::launch_haitsma_batch(d_pcm, d_offsets, n_total, n_clips, sample_rate, tl_edges, ctx->audio_ws, w, d_out, cap_frames,
                               d_out_offsets, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->audio_done, st));
    return UCFP_OK;
}

int ucfp_audio_haitsma(ucfp_ctx* ctx, const float* pcm, size_t n, uint32_t sample_rate,
                       const ucfp_haitsma_config* cfg, uint32_t* out, size_t cap_frames, size_t* n_frames) {
    if (!ctx || !n_frames) return fail(UCFP_E_INVALID, "ctx/n_frames is NULL");
    *n_frames = 0;
    if (sample_rate == 0 || sample_rate > 384000) return fail(UCFP_E_MODALITY, "invalid sample rate %u", sample_rate);
    if (n && !pcm) return fail(UCFP_E_INVALID, "pcm is NULL");
    const size_t n5 = sample_rate == 5000 ? n : ucfp::audio_resample_len(n, sample_rate, 5000);
    const size_t frames = ucfp::audio_stft_frames(n5, 2048, 64);
    if (frames > cap_frames) return fail(UCFP_E_INVALID, "output holds %zu frames, %zu produced", cap_frames, frames);
    if (frames == 0) return UCFP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    float *d_in = nullptr, *d_5k = nullptr;
    uint32_t* d_out = nullptr;
    hipStream_t st = ctx->host_stream;
    HIP_TRY(hipMalloc((void**)&d_in, n * 4));
    hipError_t e = hipMalloc((void**)&d_out, frames * 4);
    if (e == hipSuccess && sample_rate != 5000) e = hipMalloc((void**)&d_5k, n5 * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, pcm, n * 4, hipMemcpyHostToDevice, st);
    int rc = UCFP_OK;
    if (e == hipSuccess) {
        const float* src = d_in;
        if (sample_rate != 5000) {
            ucfp::launch_resample_linear(d_in, n, sample_rate, 5000, d_5k, st);
            src = d_5k;
        }
        rc = ucfp_audio_haitsma_dev(ctx, src, n5, cfg, d_out, frames, st);
        if (rc == UCFP_OK) e = hipMemcpyAsync(out, d_out, frames * 4, hipMemcpyDeviceToHost, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_in);
    if (d_5k) (void)hipFree(d_5k);
    if (d_out) (void)hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(UCFP_E_INDEX, "haitsma failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    *n_frames = frames;
    return UCFP_OK;
}

// ---------------------------------- text ----------------------------------------------

static int text_check(ucfp_ctx* ctx, const void* utf8, const void* offsets, size_t n, int mode, const void* out) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    if (mode != UCFP_TEXT_RAW_ASCII && mode != UCFP_TEXT_PRETOKENIZED)
        return fail(UCFP_E_INVALID, "unknown text mode %d", mode);
    if (n && (!offsets || !out)) return fail(UCFP_E_INVALID, "offsets/out is NULL");
    if (n > 0x7fffffffu) return fail(UCFP_E_INVALID, "batch of %zu documents exceeds one launch", n);
    (void)utf8;
    return UCFP_OK;
}

int ucfp_text_minhash_batch_dev(ucfp_ctx* ctx, const uint8_t* d_utf8, const uint64_t* d_offsets, size_t n,
                                int mode, uint32_t shingle_k, uint8_t* d_out, int32_t* d_status, void* stream) {
    int rc = text_check(ctx, d_utf8, d_offsets, n, mode, d_out);
    if (rc) return rc;
    if (shingle_k == 0 || shingle_k > 64) return fail(UCFP_E_MODALITY, "shingle k must be in [1, 64] (got %u)", shingle_k);
    ucfp::launch_text_minhash(d_utf8, d_offsets, n, mode, shingle_k, d_out, d_status, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UCFP_OK;
}

int ucfp_text_simhash_batch_dev(ucfp_ctx* ctx, const uint8_t* d_utf8, const uint64_t* d_offsets, size_t n,
                                int mode, uint8_t* d_out, int32_t* d_status, void* stream) {
    int rc = text_check(ctx, d_utf8, d_offsets, n, mode, d_out);
    if (rc) return rc;
    ucfp::launch_text_simhash(d_utf8, d_offsets, n, mode, d_out, d_status, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UCFP_OK;
}

static int text_host(ucfp_ctx* ctx, bool sim, const uint8_t* utf8, const uint64_t* offsets, size_t n, int mode,
                     uint32_t k, uint8_t* out, int32_t* status) {
    int rc = text_check(ctx, utf8, offsets, n, mode, out);
    if (rc) return rc;
    if (n == 0) return UCFP_OK;
    if (!sim && (k == 0 || k > 64)) return fail(UCFP_E_MODALITY, "shingle k must be in [1, 64] (got %u)", k);
    const size_t base = offsets[0], total = offsets[n] - offsets[0];
    for (size_t i = 0; i < n; i++)
        if (offsets[i + 1] < offsets[i]) return fail(UCFP_E_INVALID, "offsets must be non-decreasing");
    const size_t rec = sim ? UCFP_SIMHASH_BYTES : UCFP_MINHASH_BYTES;
    const size_t o_off = (total + 16 + 255) & ~(size_t)255;
    const size_t in_bytes = o_off + (n + 1) * 8;
    const size_t o_st = (n * rec + 255) & ~(size_t)255;
    const size_t out_bytes = o_st + n * 4;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    rc = grow(&ctx->stage_in, &ctx->stage_in_cap, in_bytes);
    if (rc) return rc;
    rc = grow(&ctx->stage_out, &ctx->stage_out_cap, out_bytes);
    if (rc) return rc;
    hipStream_t st = ctx->host_stream;
    std::vector<uint64_t> rel(n + 1);
    for (size_t i = 0; i <= n; i++) rel[i] = offsets[i] - base;
    if (total) HIP_TRY(hipMemcpyAsync(ctx->stage_in, utf8 + base, total, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ctx->stage_in + o_off, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    const uint64_t* d_off = reinterpret_cast<const uint64_t*>(ctx->stage_in + o_off);
    int32_t* d_st = reinterpret_cast<int32_t*>(ctx->stage_out + o_st);
    if (sim) ucfp::launch_text_simhash(ctx->stage_in, d_off, n, mode, ctx->stage_out, d_st, st);
    else ucfp::launch_text_minhash(ctx->stage_in, d_off, n, mode, k, ctx->stage_out, d_st, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, ctx->stage_out, n * rec, hipMemcpyDeviceToHost, st));
    if (status) HIP_TRY(hipMemcpyAsync(status, d_st, n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return UCFP_OK;
}

int ucfp_text_minhash_batch(ucfp_ctx* ctx, const uint8_t* utf8, const uint64_t* offsets, size_t n, int mode,
                            uint32_t shingle_k, uint8_t* out, int32_t* status) {
    return text_host(ctx, false, utf8, offsets, n, mode, shingle_k, out, status);
}

int ucfp_text_simhash_batch(ucfp_ctx* ctx, const uint8_t* utf8, const uint64_t* offsets, size_t n, int mode,
                            uint8_t* out, int32_t* status) {
    return text_host(ctx, true, utf8, offsets, n, mode, 1, out, status);
}

int ucfp_image_record_codes_dev(ucfp_ctx* ctx, const uint8_t* d_records, size_t n, uint32_t algo, uint32_t which,
                                uint64_t* d_codes, void* stream) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    if (n && (!d_records || !d_codes)) return fail(UCFP_E_INVALID, "records/codes is NULL");
    const size_t rec = ucfp_image_record_bytes(algo);
    if (rec == 0) return fail(UCFP_E_INVALID, "algo mask %u is not a record type", algo);
    uint32_t offset = 32;   // single-algorithm record: exact[32] | global_hash
    if (algo == UCFP_IMG_MULTI) {
        if (which == UCFP_IMG_AHASH) offset = 32 + 32;
        else if (which == UCFP_IMG_PHASH) offset = 32 + 168 + 32;
        else if (which == UCFP_IMG_DHASH) offset = 32 + 336 + 32;
        else return fail(UCFP_E_INVALID, "`which` must name one algorithm of the bundle (got %u)", which);
    }
    if ((uintptr_t)d_records & 3) return fail(UCFP_E_INVALID, "records need a 4-byte aligned base");
    ucfp::launch_image_record_codes(d_records, n, (uint32_t)rec, offset, d_codes, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UCFP_OK;
}

int ucfp_image_synth_dev(ucfp_ctx* ctx, uint8_t* frames, size_t n, uint32_t width, uint32_t height,
                         size_t first_index, void* stream) {
    if (!ctx) return fail(UCFP_E_INVALID, "ctx is NULL");
    if (n && !frames) return fail(UCFP_E_INVALID, "frames is NULL");
    if (((size_t)width * height) % 4 != 0 || ((uintptr_t)frames & 3))
        return fail(UCFP_E_INVALID, "synth needs width*height %% 4 == 0 and a 4-byte aligned base");
    ucfp::launch_image_synth(frames, n, width, height, first_index, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UCFP_OK;
}

}  // extern "C"
