// ubench_valu.hip -- measures the sustained rate of the integer VALU ops the Hamming scan is
// made of (v_xor_b32 with an SGPR operand, v_bcnt_u32_b32, v_min3_i32) on this chip, so the
// ANN roofline in DESIGN.md is a measured ceiling, not a datasheet guess.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t s0, uint32_t s1, int iters) {
    uint32_t q0 = threadIdx.x * 2654435761u, q1 = blockIdx.x * 40503u + threadIdx.x;
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t sa = s0, sb = s1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) {  // xor, xor, bcnt, bcnt (the pair kernel: 4 ops)
                uint32_t a = q0 ^ (sa + j), b = q1 ^ (sb + j);
                acc[j] += __builtin_popcount(a) + __builtin_popcount(b);
            } else if (MODE == 1) {  // xor only (2 ops + add)
                acc[j] += (q0 ^ (sa + j)) ^ (q1 ^ (sb + j));
            } else if (MODE == 3) {  // v_dot4_u32_u8, 8 independent accumulators
                acc[j] = __builtin_amdgcn_udot4(q0 + j, q1, acc[j], false);
            } else if (MODE == 4) {  // v_alignbyte_b32
                acc[j] += __builtin_amdgcn_alignbyte(q0, acc[j], (uint32_t)(j & 3));
            } else if (MODE == 5) {  // v_mad_u64_u32 (64-bit accumulate)
                unsigned long long t = ((unsigned long long)acc[(j + 1) & 7] << 32) | acc[j];
                t = (unsigned long long)(q0 + j) * q1 + t;
                acc[j] = (uint32_t)t ^ (uint32_t)(t >> 32);
            } else if (MODE == 6) {  // v_mul_lo_u32 (32 x 32 -> low 32)
                acc[j] = acc[j] * (q0 | 1u) + j;
            } else if (MODE == 7) {  // v_mul_hi_u32
                acc[j] = __umulhi(acc[j] + q1, q0 | 0x80000001u);
            } else if (MODE == 8) {  // v_mad_u32_u24
                acc[j] = __umul24(acc[j], q0) + q1;
            } else if (MODE == 9) {  // DPP row_shr:1 add (what a wave prefix scan is made of)
                acc[j] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc[j], 0x111, 0xf, 0xf, true);
            } else if (MODE == 10) {  // ds_bpermute shuffle + add (what __shfl_up compiles to)
                acc[j] += (uint32_t)__shfl_up((int)acc[j], 1, 64);
            } else if (MODE == 11) {  // u32 -> f32, multiply, f32 -> u32 (the rounding division's estimate)
                acc[j] = (uint32_t)((float)(acc[j] | 1u) * __uint_as_float(0x3c000000u | (q0 & 0xffffu))) + j;
            } else if (MODE == 12) {  // v_bfe_u32 + v_mad_u32_u24: byte extract and multiply-add
                acc[j] = __umul24((q0 >> (8 * (j & 3))) & 255u, q1 & 0x1ffu) + acc[j];
            } else {  // fma f32 reference: 1 op
                float f = __uint_as_float(acc[j]);
                f = fmaf(f, 1.0001f, 0.5f);
                acc[j] = __float_as_uint(f);
            }
        }
        sa = sa * 1664525u + 1013904223u;  // scalar LCG keeps the operands in SGPRs and non-constant
        sb = sb * 22695477u + 1u;
    }
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) r ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
double run(const char* name, double ops_per_inner, int iters) {
    uint32_t* d;
    const int blocks = 256 * 8, threads = 256;
    hipMalloc(&d, blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(d, 1, 2, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(d, 3, 4, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double laneops = (double)blocks * threads * iters * 8 * ops_per_inner;
    printf("%-28s %8.3f ms  %7.2f T lane-op/s (counting %.0f VALU ops per inner step)\n", name, ms,
           laneops / (ms * 1e-3) / 1e12, ops_per_inner);
    hipFree(d);
    return laneops / (ms * 1e-3);
}

int main() {
    run<0>("xor,xor,bcnt,bcnt(+acc)", 4, 4000);
    run<1>("xor,xor,xor,add", 4, 4000);
    run<2>("fma_f32", 1, 16000);
    run<3>("dot4_u32_u8 (acc chain x8)", 1, 16000);
    run<4>("alignbyte + add", 2, 8000);
    run<5>("mad_u64_u32 + xor", 2, 8000);
    run<6>("mul_lo_u32 + add", 2, 8000);
    run<7>("add + mul_hi_u32", 2, 8000);
    run<8>("mad_u32_u24", 1, 16000);
    run<9>("dpp row_shr:1 + add", 1, 16000);
    run<10>("shfl_up (ds_bpermute) + add", 3, 4000);
    run<11>("or, cvt_f32_u32, mul_f32, cvt_u32_f32, add", 5, 4000);
    run<12>("bfe + mad_u32_u24", 2, 8000);
    return 0;
}
