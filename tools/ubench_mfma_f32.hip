// ubench_mfma_f32.hip -- v_mfma_f32_16x16x4_f32 issue patterns: how the accumulator dependency distance
// limits the rate (the cosine kernel's inner loop).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_f32.hip -o tools/ubench_mfma_f32.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// NACC accumulators used round-robin: dependency distance = NACC.  RANDOM: the A/B operands of consecutive MFMAs
// are different registers of random floats (what a real GEMM feeds the pipes; the chip clocks down under it).
template <int NACC, bool RANDOM = false>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.0f;
    float ra[8], rb[8];
    unsigned h = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u);
    for (int i = 0; i < 8; i++) {
        h = h * 1664525u + 1013904223u;
        ra[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
        h = h * 1664525u + 1013904223u;
        rb[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 12; j++)
            acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(RANDOM ? ra[j % 8] : a, RANDOM ? rb[(j + 3) % 8] : b,
                                                                acc[j % NACC], 0, 0, 0);
    }
    float r = 0.f;
    for (int i = 0; i < NACC; i++) r += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC, bool RANDOM = false>
void run(int waves, int iters = 4000) {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<NACC, RANDOM><<<256, waves * 64>>>(d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NACC, RANDOM><<<256, waves * 64>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * waves * iters * 12 * 2048.0;
    printf("accumulators=%d waves/CU=%d operands=%s  %.3f ms  %.1f TFLOP/s\n", NACC, waves, RANDOM ? "random" : "constant", ms,
           flop / ms / 1e9);
    (void)hipFree(d);
}

int main() {
    run<1>(4); run<2>(4); run<3>(4); run<4>(4); run<6>(4); run<12>(4);
    run<1>(8); run<2>(8); run<3>(8); run<4>(8); run<6>(8); run<12>(8);
    run<1, true>(8); run<4, true>(8); run<12, true>(8);
    // sustained: tens of milliseconds, back to back (power management acts on this scale, not on 1 ms)
    for (int r = 0; r < 3; r++) run<4, true>(8, 60000);
    for (int r = 0; r < 2; r++) run<4, false>(8, 60000);
    return 0;
}
