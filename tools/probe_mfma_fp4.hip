// probe_mfma_fp4.hip -- checks the operand / result layout of v_mfma_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands as the
// Hamming filter would use it: A = code bits as 0.0 / 1.0 nibbles, B = query bits as -1.0 / +1.0 nibbles; lane l holds
// row (column) l & 31 and the 32 K elements of half l >> 5.  Expected: D[code][query] = popc(c & q) - popc(c & ~q).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_fp4.hip -o tools/probe_mfma_fp4.bin && tools/probe_mfma_fp4.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ uint32_t nib8(uint32_t byte) {   // bit i of byte -> nibble i (0 / 1)
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) r |= ((byte >> i) & 1u) << (4 * i);
    return r;
}

__global__ void k(const uint64_t* codes, const uint64_t* queries, float* out) {
    const int l = threadIdx.x, nn = l & 31, hh = l >> 5;
    const uint32_t cb = (uint32_t)(codes[nn] >> (32 * hh)), qb = (uint32_t)(queries[nn] >> (32 * hh));
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < 4; j++) {
        a[j] = (int)(nib8((cb >> (8 * j)) & 255u) << 1);                      // 1 -> 0x2 (+1.0), 0 -> 0x0
        b[j] = (int)(0x22222222u | (~nib8((qb >> (8 * j)) & 255u) & 0x11111111u) << 3);   // 1 -> 0x2 (+1.0), 0 -> 0xA (-1.0)
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
    for (int i = 0; i < 16; i++) out[l * 16 + i] = c[i];
}

int main() {
    uint64_t hc[32], hq[32];
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < 32; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull; hc[i] = s ^ (s >> 29);
        s = s * 6364136223846793005ull + 1442695040888963407ull; hq[i] = s ^ (s >> 31);
    }
    uint64_t *dc, *dq;
    float* dout;
    (void)hipMalloc(&dc, 256); (void)hipMalloc(&dq, 256); (void)hipMalloc(&dout, 64 * 16 * 4);
    (void)hipMemcpy(dc, hc, 256, hipMemcpyHostToDevice);
    (void)hipMemcpy(dq, hq, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dc, dq, dout);
    float ho[64 * 16];
    (void)hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 16; i++) {
            const int q = l & 31, code = 8 * (i / 4) + 4 * (l >> 5) + (i % 4);
            const int want = __builtin_popcountll(hc[code] & hq[q]) - __builtin_popcountll(hc[code] & ~hq[q]);
            if ((int)ho[l * 16 + i] != want || ho[l * 16 + i] != (float)want) {
                if (bad < 8) printf("lane %d i %d: got %g want %d\n", l, i, ho[l * 16 + i], want);
                bad++;
            }
        }
    printf("fp4 layout probe: %s (%d mismatches of 1024)\n", bad ? "MISMATCH" : "ok", bad);
    return bad != 0;
}
