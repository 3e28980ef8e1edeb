// Does v_mfma_f32_16x16x32_f16 keep f16 subnormal operands on gfx950?  (cosine_mins_eps counts on it: normalised components
// below 2^-14 reach the matrix pipe as subnormals.)  Prints the product of a subnormal A against 1.0 in B, and the f32 -> f16
// conversion of a value in the subnormal range.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_f16_denorm.hip -o tools/probe_mfma_f16_denorm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float a_val, float* out) {
    const _Float16 a = (_Float16)a_val;           // conversion in the kernel (v_cvt_f16_f32, round to nearest even)
    f16x8 A, B;
    for (int e = 0; e < 8; e++) {
        A[e] = e == 0 ? a : (_Float16)0.f;
        B[e] = (_Float16)1.0f;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = acc[0];
        out[1] = (float)a;
    }
}
int main() {
    float* d;
    hipMalloc(&d, 8);
    const float vals[] = {3.0e-5f, 6.0e-8f, 1.0e-6f, 6.2e-5f};
    for (float v : vals) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, v, d);
        float h[2];
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        // lanes 0..15 of k-group 0 hold a in slot 0: D[0][0] = sum over the 4 k-groups of a * 1 = 4 a
        printf("a = %.9g: as f16 %.9g, mfma sum %.9g (4 a = %.9g) -> %s\n", v, h[1], h[0], 4.0 * h[1],
               h[0] != 0.f ? "subnormals kept" : "FLUSHED");
    }
    return 0;
}
