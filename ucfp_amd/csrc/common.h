// common.h -- internal declarations shared by the HIP translation units and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ucfp {

// image.hip
int launch_image_hash(uint32_t algo, const uint8_t* frames, size_t n, uint32_t w, uint32_t h,
                      size_t row_stride, size_t frame_stride, int pixfmt, uint32_t min_dim,
                      uint32_t max_dim, const uint8_t* exact, uint8_t* out, int32_t* status,
                      uint8_t* norm_ws, size_t norm_ws_frames, hipStream_t stream);
int launch_image_synth(uint8_t* frames, size_t n, uint32_t w, uint32_t h, size_t first,
                       hipStream_t stream);

}  // namespace ucfp
