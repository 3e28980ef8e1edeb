// ubench_pk_f32.hip -- issue rate of the packed-f32 VALU instructions the audio FFT butterflies are made of
// (v_pk_mul_f32 / v_pk_add_f32, no FMA: the oracle rounds every product and sum), beside v_mul_f32 and
// v_pk_fma_f32, so the audio roofline in DESIGN.md rests on a measured ceiling.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_pk_f32.hip -o tools/ubench_pk_f32.bin && tools/ubench_pk_f32.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float s, int iters) {
    f32x2 a[8];
    float b[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        a[j] = f32x2{(float)threadIdx.x + j, 1.0f + j};
        b[j] = (float)threadIdx.x + j;
    }
    const f32x2 m = {s, 1.0f / s};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(m));
            if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(m));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(m));
            if (MODE == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(b[j]) : "v"(s));
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "+v"(a[j]) : "v"(m));
            if (MODE == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(b[j]) : "v"(s));
        }
    }
    float r = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) r += a[j].x + a[j].y + b[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, int iters) {
    float* d;
    const int blocks = 256 * 8, threads = 256;
    hipMalloc(&d, blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(d, 1.0001f, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(d, 1.0002f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * (threads / 64) * iters * 8;   // wave-instructions
    // 1024 SIMDs: cycles per wave-instruction per SIMD at 2.4 GHz
    printf("%-34s %8.3f ms  %7.2f G wave-instr/s  = %5.2f cycles per wave-instr per SIMD @2.4 GHz\n", name, ms,
           winst / (ms * 1e-3) / 1e9, 1024.0 * 2.4e9 * (ms * 1e-3) / winst);
    hipFree(d);
}

int main() {
    run<3>("v_mul_f32", 16000);
    run<5>("v_add_f32", 16000);
    run<0>("v_pk_mul_f32", 16000);
    run<1>("v_pk_add_f32", 16000);
    run<4>("v_pk_mul_f32 op_sel+neg", 16000);
    run<2>("v_pk_fma_f32", 16000);
    return 0;
}
