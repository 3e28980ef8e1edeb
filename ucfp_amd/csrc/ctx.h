// ctx.h -- the context object behind the C ABI, shared by the translation units that implement it (capi.hip, upload.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/ucfp_hip.h"

namespace ucfp {
int capi_fail(int code, const char* fmt, ...);
}

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return ucfp::capi_fail(UCFP_E_INDEX, "%s failed: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

struct ucfp_ctx {
    int device = 0;
    uint8_t* norm_ws = nullptr;  // kNormWsFrames x 65536: ONE scratch area shared by every generic-geometry launch,
    std::mutex norm_mu;          // so its users are ordered across streams: enqueue under norm_mu, wait on / record
    hipEvent_t norm_done = nullptr;  // norm_done around the launch (ucfp::image_hash_ordered)
    // host-variant staging (grown on demand), guarded by `mu`
    std::mutex mu;
    uint8_t* stage_in = nullptr;
    size_t stage_in_cap = 0;
    uint8_t* stage_out = nullptr;
    size_t stage_out_cap = 0;
    hipStream_t host_stream = nullptr;
    // audio workspace (spilled spectrogram chunk, candidate lists), shared by successive calls
    uint8_t* audio_ws = nullptr;
    size_t audio_ws_cap = 0;
    hipEvent_t audio_done = nullptr;
    // PNG front end: gathered zlib streams, filtered scanlines, decoded frames of the last batch
    uint8_t* png_ws = nullptr;
    size_t png_ws_cap = 0;
    hipEvent_t png_done = nullptr;
    // mixed uploads: the JPEG chain runs on `side` beside the PNG chain (a few hundred one-wave inflates leave most of the
    // chip idle for tens of milliseconds); forked from and joined into the caller's stream with these two events
    hipStream_t side = nullptr;
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    // BLAKE3 chaining values of the last batch (+ the digests when the PNG call computes `exact` itself); ordered by png_done
    uint8_t* b3_ws = nullptr;
    size_t b3_ws_cap = 0;
    // per-frame tables of ragged image batches (ImgItem rows, rejected slots): two pinned + device buffer pairs used in
    // turn; `used[i]` is recorded behind the kernels that read pair i and waited for before the host rewrites it
    std::mutex item_mu;
    uint8_t* item_h[2] = {nullptr, nullptr};
    uint8_t* item_d[2] = {nullptr, nullptr};
    size_t item_cap[2] = {0, 0};
    hipEvent_t item_used[2] = {nullptr, nullptr};
    int item_next = 0;
    size_t any_max_pixels = (size_t)1 << 20;   // uniform batches: frames up to this size take the fused any-geometry kernel
    // column tables of the fused any-geometry kernel, one per width (0 .. 2048), made on first sight of a width
    std::mutex geo_mu;
    uint32_t* geo = nullptr;
    uint64_t geo_have[2049 / 64 + 1] = {0};
};


namespace ucfp {
// (re)allocates *p to at least `need` bytes of device memory (contents are not kept); returns a ucfp_status
inline int grow(uint8_t** p, size_t* cap, size_t need) {
    if (*cap >= need) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4;
    HIP_TRY(hipMalloc((void**)p, want));
    *cap = want;
    return 0;
}
// the context's column table of width w is on the device (first sight: built on the host, copied with a BLOCKING copy)
int image_any_geometry_ready(ucfp_ctx* ctx, uint32_t w);
int image_hash_ordered(ucfp_ctx* ctx, uint32_t algo, const uint8_t* frames, size_t n, uint32_t w, uint32_t h,
                       size_t row_stride, size_t frame_stride, int pixfmt, uint32_t min_dim, uint32_t max_dim,
                       const uint8_t* exact, uint8_t* out, int32_t* status, hipStream_t stream);
}  // namespace ucfp
