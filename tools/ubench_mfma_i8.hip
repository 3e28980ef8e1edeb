// ubench_mfma_i8.hip -- issue model behind the Hamming matrix-core filter, int8 and FP4 forms: ns per MFMA
// per SIMD for v_mfma_i32_16x16x64_i8 and v_mfma_i32_32x32x32_i8 with v_max3_i32 fillers in the gaps.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_i8.hip -o tools/ubench_mfma_i8.bin && tools/ubench_mfma_i8.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0..3: 16x16x64 with MODE v_max3 per MFMA (results folded one iteration late)
// MODE 10: 32x32x32, C = 0           MODE 11: 32x32x32, C = previous result (accumulate chain of 2)
// MODE 12: MODE 11 + 8 v_max3 per pair on the OTHER buffer's results (software-pipelined fold)
// MODE 13: pair + 8 v_max3 on its own results right after (what the plain loop does)
template <int MODE>
__global__ __launch_bounds__(1024) void k(int* out, int iters) {
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7}, c = {0, 0, 0, 0};
    const bool sign_random = iters < 0;
    if (iters < 0) {   // negative iters: random 0/1 and +-1 bytes (the data the Hamming filter feeds): toggling -> power -> clock
        iters = -iters;
        uint32_t h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 12345u);
        for (int j = 0; j < 4; j++) {
            h = h * 1664525u + 1013904223u;
            a[j] = (int)(h & 0x01010101u);
            h = h * 1664525u + 1013904223u;
            const uint32_t t = h & 0x01010101u;
            b[j] = (int)(~((t << 8) - t) | 0x01010101u);
        }
    }
    int m0 = -1, m1 = -2, m2 = -3;
    int r = 0;
    if (MODE >= 20) {
        // FP4 (e2m1) operands: nibble 0x2 = 1.0, 0xA = -1.0, 0 = 0; K = 64 in ONE instruction
        if (sign_random) {
            uint32_t h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 777u);
            for (int j = 0; j < 4; j++) {
                h = h * 1664525u + 1013904223u;
                a[j] = (int)((h & 0x11111111u) << 1);
                h = h * 1664525u + 1013904223u;
                b[j] = (int)(0x22222222u | ((h & 0x11111111u) << 3));
            }
        } else {
            for (int j = 0; j < 4; j++) a[j] = 0x22222222, b[j] = 0x22222222;
        }
        f32x16 d[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) d[j][e] = (float)(j + e);
        float f0 = -1.f, f1 = -2.f;
        f32x16 cc;
#pragma unroll
        for (int e = 0; e < 16; e++) cc[e] = 8388608.f + 4194304.f + 1088.f;
        const int sa = 127 + 16, sb = 127;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (MODE == 20)
                    asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, 0 cbsz:4 blgp:4\n\tv_mfma_f32_32x32x64_f8f6f4 %3, %1, %2, 0 cbsz:4 blgp:4"
                                 : "=&v"(d[j]) : "v"(a), "v"(b), "v"(d[(j + 2) & 3]));
                if (MODE == 23) {
                    // two code tiles packed into one result (C = bias, second MFMA block-scaled by 2^16) and folded four
                    // values per instruction with v_pk_maximum3_f16: 8 fold instructions per TWO MFMAs
                    const f32x16& p = d[(j + 2) & 3];
                    asm volatile(
                        "v_mfma_f32_32x32x64_f8f6f4 %0, %3, %4, %21 cbsz:4 blgp:4\n\t"
                        "v_pk_maximum3_f16 %1, %1, %5, %6\n\tv_pk_maximum3_f16 %2, %2, %7, %8\n\tv_pk_maximum3_f16 %1, %1, %9, %10\n\t"
                        "v_pk_maximum3_f16 %2, %2, %11, %12\n\t"
                        "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %3, %4, %0, %22, %23 op_sel_hi:[0,0,0] cbsz:4 blgp:4\n\t"
                        "v_pk_maximum3_f16 %1, %1, %13, %14\n\tv_pk_maximum3_f16 %2, %2, %15, %16\n\tv_pk_maximum3_f16 %1, %1, %17, %18\n\t"
                        "v_pk_maximum3_f16 %2, %2, %19, %20"
                        : "=&v"(d[j]), "+v"(f0), "+v"(f1)
                        : "v"(a), "v"(b), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]),
                          "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]),
                          "v"(p[15]), "v"(cc), "v"(sa), "v"(sb));
                }
                if (MODE == 21 || MODE == 22) {
                    // two tiles per asm block: the fold of the other buffers' results in the MFMA shadows
                    const f32x16& p = d[(j + 2) & 3];
                    asm volatile(
                        "v_mfma_f32_32x32x64_f8f6f4 %0, %3, %4, 0 cbsz:4 blgp:4\n\t"
                        "v_max3_f32 %1, %1, %5, %6\n\tv_max3_f32 %2, %2, %7, %8\n\tv_max3_f32 %1, %1, %9, %10\n\t"
                        "v_max3_f32 %2, %2, %11, %12\n\t"
                        "v_max3_f32 %1, %1, %13, %14\n\tv_max3_f32 %2, %2, %15, %16\n\tv_max3_f32 %1, %1, %17, %18\n\t"
                        "v_max3_f32 %2, %2, %19, %20"
                        : "=&v"(d[j]), "+v"(f0), "+v"(f1)
                        : "v"(a), "v"(b), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]),
                          "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]),
                          "v"(p[15]));
                    if (MODE == 22)   // an i8-rate equivalent: a second MFMA without fold work
                        asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, 0 cbsz:4 blgp:4" : "=&v"(d[(j + 1) & 3]) : "v"(a), "v"(b));
                }
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
#pragma unroll
        for (int j = 0; j < 4; j++) r ^= __float_as_int(d[j][0]) ^ __float_as_int(d[j][15]);
        r ^= __float_as_int(f0) ^ __float_as_int(f1);
    } else if (MODE < 10) {
        i32x4 d[8];
#pragma unroll
        for (int j = 0; j < 8; j++) d[j] = i32x4{j, j, j, j};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (MODE == 0) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "+v"(d[j]) : "v"(a), "v"(b), "v"(c));
                if (MODE == 1)
                    asm volatile("v_max3_i32 %4, %4, %5, %6\n\tv_mfma_i32_16x16x64_i8 %0, %1, %2, %3"
                                 : "+v"(d[j]), "+v"(a), "+v"(b), "+v"(c), "+v"(m0) : "v"(d[j][0]), "v"(d[j][1]));
                if (MODE == 2)
                    asm volatile("v_max3_i32 %4, %4, %6, %7\n\tv_max3_i32 %5, %5, %8, %9\n\tv_mfma_i32_16x16x64_i8 %0, %1, %2, %3"
                                 : "+v"(d[j]), "+v"(a), "+v"(b), "+v"(c), "+v"(m0), "+v"(m1)
                                 : "v"(d[j][0]), "v"(d[j][1]), "v"(d[j][2]), "v"(d[j][3]));
                if (MODE == 3)
                    asm volatile("v_max3_i32 %4, %4, %7, %8\n\tv_max3_i32 %5, %5, %9, %10\n\tv_max3_i32 %6, %6, %7, %9\n\t"
                                 "v_mfma_i32_16x16x64_i8 %0, %1, %2, %3"
                                 : "+v"(d[j]), "+v"(a), "+v"(b), "+v"(c), "+v"(m0), "+v"(m1), "+v"(m2)
                                 : "v"(d[j][0]), "v"(d[j][1]), "v"(d[j][2]), "v"(d[j][3]));
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
#pragma unroll
        for (int j = 0; j < 8; j++) r ^= d[j][0] ^ d[j][3];
    } else {
        i32x16 d[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) d[j][e] = j + e;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (MODE == 10)
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0\n\tv_mfma_i32_32x32x32_i8 %3, %1, %2, 0"
                                 : "=&v"(d[j]) : "v"(a), "v"(b), "v"(d[(j + 2) & 3]));
                if (MODE == 11)
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, %0"
                                 : "=&v"(d[j]) : "v"(a), "v"(b));
                if (MODE == 12) {
                    const i32x16& p = d[(j + 2) & 3];
                    asm volatile(
                        "v_mfma_i32_32x32x32_i8 %0, %3, %4, 0\n\t"
                        "v_max3_i32 %1, %1, %5, %6\n\tv_max3_i32 %2, %2, %7, %8\n\tv_max3_i32 %1, %1, %9, %10\n\t"
                        "v_max3_i32 %2, %2, %11, %12\n\t"
                        "v_mfma_i32_32x32x32_i8 %0, %3, %4, %0\n\t"
                        "v_max3_i32 %1, %1, %13, %14\n\tv_max3_i32 %2, %2, %15, %16\n\tv_max3_i32 %1, %1, %17, %18\n\t"
                        "v_max3_i32 %2, %2, %19, %20"
                        : "=&v"(d[j]), "+v"(m0), "+v"(m1)
                        : "v"(a), "v"(b), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]),
                          "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]),
                          "v"(p[15]));
                }
                if (MODE == 13) {
                    d[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
                    d[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, d[j], 0, 0, 0);
                    int m = d[j][0];
#pragma unroll
                    for (int e = 1; e + 1 < 16; e += 2) m = max(max(m, d[j][e]), d[j][e + 1]);
                    m0 = max(max(m0, m), d[j][15]);
                    a[0] ^= m0 & 1;
                }
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
#pragma unroll
        for (int j = 0; j < 4; j++) r ^= d[j][0] ^ d[j][15];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ m0 ^ m1 ^ m2;
}

template <int MODE>
void run(int waves_per_simd, int sign = 1) {
    int* d;
    const int blocks = 256, threads = 256 * waves_per_simd, iters = 10000;
    (void)hipMalloc(&d, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(d, sign * iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(d, sign * iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_iter = MODE == 21 ? 4 : 8;
    const double mfmas_per_simd = (double)iters * per_iter * waves_per_simd;
    const double pairs = (MODE >= 20 ? 1024.0 : MODE < 10 ? 256.0 : 1024.0 / 2);   // code-query pairs one MFMA completes
    printf("%s mode=%2d waves/SIMD=%d  %.3f ms  %.2f ns per MFMA per SIMD  -> %.1f T pairs/s on 1024 SIMDs\n", sign < 0 ? "random" : "const ", MODE,
           waves_per_simd, ms, ms * 1e6 / mfmas_per_simd, pairs / (ms * 1e6 / mfmas_per_simd) * 1024 / 1e3);
    (void)hipFree(d);
}

int main() {
    for (int w = 1; w <= 4; w++) {
        if (w == 3) continue;
        run<0>(w);
        run<2>(w);
        run<10>(w);
        run<11>(w);
        run<12>(w);
        run<13>(w);
        run<2>(w, -1);
        run<11>(w, -1);
        run<12>(w, -1);
        run<20>(w);
        run<21>(w);
        run<20>(w, -1);
        run<21>(w, -1);
        run<22>(w, -1);
        run<23>(w);
        run<23>(w, -1);
    }
    return 0;
}
