// ubench_hamming_core.hip -- the INT8 inner loop hamming_scan_mfma had in rounds 1-2 (it now uses FP4 operands: see
// ubench_mfma_i8.hip modes 20-22), in isolation (LDS-resident query tiles,
// 4 code tiles in registers, software-pipelined MFMA + v_max3 fold, never-taken hit branch), with knobs
// to find what separates it from the bare MFMA + v_max3 stream of ubench_mfma_i8.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_hamming_core.hip -o tools/ubench_hamming_core.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// One code tile of the software-pipelined step, as text: MFMA (K half 0), four v_max3 folding the previous
// tile's results, MFMA (K half 1, accumulating), four more.  %0 result, %1 running max, %2/%3 the code
// tile's two K halves (A), %4/%5 the query tile's (B), %6..%21 the 16 previous results of this lane.
#define UCFP_FOLD                                     \
    "v_mfma_i32_32x32x32_i8 %0, %2, %4, 0\n\t"        \
    "v_max3_i32 %1, %6, %7, %8\n\t"                   \
    "v_max3_i32 %1, %1, %9, %10\n\t"                  \
    "v_max3_i32 %1, %1, %11, %12\n\t"                 \
    "v_max3_i32 %1, %1, %13, %14\n\t"                 \
    "v_mfma_i32_32x32x32_i8 %0, %3, %5, %0\n\t"       \
    "v_max3_i32 %1, %1, %15, %16\n\t"                 \
    "v_max3_i32 %1, %1, %17, %18\n\t"                 \
    "v_max3_i32 %1, %1, %19, %20\n\t"                 \
    "v_max_i32 %1, %1, %21"
// the last tile of a step also folds the four running maxima and compares with the lane's threshold:
// %2 scratch, %3 = lane mask of (max >= thr) in an SGPR pair, inputs shifted by two, %24..%26 the other
// three maxima, %27 the threshold
#define UCFP_FOLD_LAST                                \
    "v_mfma_i32_32x32x32_i8 %0, %4, %6, 0\n\t"        \
    "v_max3_i32 %1, %8, %9, %10\n\t"                  \
    "v_max3_i32 %1, %1, %11, %12\n\t"                 \
    "v_max3_i32 %1, %1, %13, %14\n\t"                 \
    "v_max3_i32 %1, %1, %15, %16\n\t"                 \
    "v_mfma_i32_32x32x32_i8 %0, %5, %7, %0\n\t"       \
    "v_max3_i32 %1, %1, %17, %18\n\t"                 \
    "v_max3_i32 %1, %1, %19, %20\n\t"                 \
    "v_max3_i32 %1, %1, %21, %22\n\t"                 \
    "v_max_i32 %1, %1, %23\n\t"                       \
    "v_max3_i32 %2, %1, %24, %25\n\t"                 \
    "v_max_i32 %2, %2, %26\n\t"                       \
    "v_cmp_ge_i32 %3, %2, %27\n\t"                  \
    "s_nop 1"
#define UCFP_FOLD_IN(b)                                                                                           \
    "v"(A[b][0]), "v"(A[b][1]), "v"(b0), "v"(b1), "v"(Dp[b][0]), "v"(Dp[b][1]), "v"(Dp[b][2]), "v"(Dp[b][3]),     \
        "v"(Dp[b][4]), "v"(Dp[b][5]), "v"(Dp[b][6]), "v"(Dp[b][7]), "v"(Dp[b][8]), "v"(Dp[b][9]), "v"(Dp[b][10]), \
        "v"(Dp[b][11]), "v"(Dp[b][12]), "v"(Dp[b][13]), "v"(Dp[b][14]), "v"(Dp[b][15])


// LDSOPS: prefetch the next tile's operands from LDS (else reuse the registers); SAMEA: all four code
// tiles use the same A registers
template <int WAVES, bool LDSOPS, bool SAMEA>
__global__ __launch_bounds__(WAVES * 64) void core(int* out, int ntiles, int supers) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    i32x4* QB = reinterpret_cast<i32x4*>(lds);
    int* THR = reinterpret_cast<int*>(lds + (size_t)(ntiles + 2) * 2048);
    uint32_t h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 12345u);
    for (int s = threadIdx.x; s < (ntiles + 2) * 128; s += WAVES * 64) {
        i32x4 v;
        for (int j = 0; j < 4; j++) {
            h = h * 1664525u + 1013904223u;
            const uint32_t t = h & 0x01010101u;
            v[j] = (int)(~((t << 8) - t) | 0x01010101u);
        }
        QB[s] = v;
    }
    for (int s = threadIdx.x; s < (ntiles + 2) * 32; s += WAVES * 64) THR[s] = 1000;
    __syncthreads();
    const int lane = threadIdx.x & 63, nn = lane & 31;
    int acc = 0;
    for (int st = 0; st < supers; st++) {
        i32x4 A[4][2];
        for (int b = 0; b < 4; b++)
            for (int k = 0; k < 2; k++)
                for (int j = 0; j < 4; j++) {
                    h = h * 1664525u + 1013904223u;
                    A[b][k][j] = (int)(h & 0x01010101u);
                }
        if (SAMEA)
            for (int b = 1; b < 4; b++) {
                A[b][0] = A[0][0];
                A[b][1] = A[0][1];
            }
        i32x16 D0[4], D1[4];
        for (int b = 0; b < 4; b++)
            for (int e = 0; e < 16; e++) D1[b][e] = -100000;
        auto step = [&](int t, i32x16 (&Dn)[4], const i32x16 (&Dp)[4], const i32x4& b0, const i32x4& b1, int thr, i32x4& n0,
                        i32x4& n1, int& nthr) {
            int m0, m1, m2, m3, mm;
            uint64_t hit;
            asm volatile(UCFP_FOLD : "=&v"(Dn[0]), "=&v"(m0) : UCFP_FOLD_IN(0) : "memory");
            if (LDSOPS) {
                n0 = QB[(t + 1) * 128 + lane];
                n1 = QB[(t + 1) * 128 + 64 + lane];
                nthr = THR[(t + 1) * 32 + nn];
            } else {
                n0 = b0;
                n1 = b1;
                nthr = thr;
            }
            asm volatile(UCFP_FOLD : "=&v"(Dn[1]), "=&v"(m1) : UCFP_FOLD_IN(1) : "memory");
            asm volatile(UCFP_FOLD : "=&v"(Dn[2]), "=&v"(m2) : UCFP_FOLD_IN(2) : "memory");
            asm volatile(UCFP_FOLD_LAST
                         : "=&v"(Dn[3]), "=&v"(m3), "=&v"(mm), "=s"(hit)
                         : UCFP_FOLD_IN(3), "v"(m0), "v"(m1), "v"(m2), "v"(thr)
                         : "memory");
            if (__builtin_expect(hit != 0, 0)) acc += mm;   // never taken
        };
        i32x4 p0 = QB[lane], p1 = QB[64 + lane], r0, r1;
        int tp = 0x7fffffff, tc = THR[nn], tn;
        int t = 0;
        for (; t + 2 <= ntiles + 1; t += 2) {
            step(t, D0, D1, p0, p1, tp, r0, r1, tn);
            tp = tc;
            tc = tn;
            step(t + 1, D1, D0, r0, r1, tp, p0, p1, tn);
            tp = tc;
            tc = tn;
        }
        if (t < ntiles + 1) step(t, D0, D1, p0, p1, tp, r0, r1, tn);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        acc += D0[0][0] + D1[3][15];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int WAVES, bool LDSOPS, bool SAMEA>
void run(int ntiles, int supers) {
    int* d;
    (void)hipMalloc(&d, 256 * WAVES * 64 * 4);
    const size_t lds = (size_t)(ntiles + 2) * (2048 + 128);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(core<WAVES, LDSOPS, SAMEA>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    core<WAVES, LDSOPS, SAMEA><<<256, WAVES * 64, lds>>>(d, ntiles, supers);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    core<WAVES, LDSOPS, SAMEA><<<256, WAVES * 64, lds>>>(d, ntiles, supers);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double pairs = 256.0 * WAVES * supers * (double)(ntiles + 1) * 4096.0;
    printf("waves/WG=%d lds_operands=%d same_A=%d ntiles=%d supers=%d  %.3f ms  -> %.1f T pairs/s (%s)\n", WAVES,
           (int)LDSOPS, (int)SAMEA, ntiles, supers, ms, pairs / ms / 1e9, hipGetErrorString(hipGetLastError()));
    (void)hipFree(d);
}

int main() {
    run<8, true, false>(64, 400);
    run<8, false, false>(64, 400);
    run<8, true, true>(64, 400);
    run<8, false, true>(64, 400);
    run<4, true, false>(64, 400);
    run<4, false, true>(64, 400);
    return 0;
}
