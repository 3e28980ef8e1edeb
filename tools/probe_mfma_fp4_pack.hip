// probe_mfma_fp4_pack.hip -- two code tiles' sums packed into ONE f32 result by the matrix core itself:
//   D = A_a x B + C                      (C = 2^23 + 2^22 + 1088 in every element)
//   D = (2^16 A_b) x B + D               (v_mfma_scale..., block scale 2^16 on A)
// leaves the float  2^23 + 65536 (64 + s_b) + (1088 + s_a),  whose bit pattern is  0x4B00'0000 | (64 + s_b) << 16 | (1088 + s_a):
// two ordered 16-bit fields (valid, monotone f16 patterns) that v_pk_maximum3_f16 folds four at a time.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_fp4_pack.hip -o tools/probe_mfma_fp4_pack.bin && tools/probe_mfma_fp4_pack.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ i32x8 code_fp4(uint32_t w) {
    i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    v[0] = (int)((w << 1) & 0x22222222u); v[1] = (int)(w & 0x22222222u);
    v[2] = (int)((w >> 1) & 0x22222222u); v[3] = (int)((w >> 2) & 0x22222222u);
    return v;
}
__device__ i32x8 query_fp4(uint32_t w) {
    const uint32_t n = ~w;
    i32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    v[0] = (int)(0x22222222u | ((n << 3) & 0x88888888u)); v[1] = (int)(0x22222222u | ((n << 2) & 0x88888888u));
    v[2] = (int)(0x22222222u | ((n << 1) & 0x88888888u)); v[3] = (int)(0x22222222u | (n & 0x88888888u));
    return v;
}

__global__ void k(const uint64_t* codes, const uint64_t* queries, uint32_t* out, uint32_t* folded) {
    const int l = threadIdx.x, nn = l & 31, hh = l >> 5;
    const i32x8 a = code_fp4((uint32_t)(codes[nn] >> (32 * hh))), a2 = code_fp4((uint32_t)(codes[32 + nn] >> (32 * hh)));
    const i32x8 b = query_fp4((uint32_t)(queries[nn] >> (32 * hh)));
    f32x16 c;
    for (int i = 0; i < 16; i++) c[i] = 8388608.f + 4194304.f + 1088.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a2, b, c, 4, 4, 0, 127 + 16, 0, 127);
    for (int i = 0; i < 16; i++) out[l * 16 + i] = __float_as_uint(c[i]);
    uint32_t m;
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(m) : "v"(c[0]), "v"(c[1]), "v"(c[2]));
    asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(m) : "v"(c[3]), "v"(c[4]));
    folded[l] = m;
}

int main() {
    uint64_t hc[64], hq[32];
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < 64; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; hc[i] = s ^ (s >> 29); }
    for (int i = 0; i < 32; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; hq[i] = s ^ (s >> 31); }
    hq[3] = ~0ull; hc[5] = ~0ull; hc[32 + 9] = ~0ull;   // the extreme: sum +64 (in tile a at row 5, in tile b at row 9)
    hq[4] = 0ull;                                        // and sum -64 against the all-ones codes
    uint64_t *dc, *dq;
    uint32_t *dout, *dfold;
    (void)hipMalloc(&dc, 512); (void)hipMalloc(&dq, 256); (void)hipMalloc(&dout, 64 * 16 * 4); (void)hipMalloc(&dfold, 256);
    (void)hipMemcpy(dc, hc, 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(dq, hq, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dc, dq, dout, dfold);
    uint32_t ho[64 * 16], hf[64];
    (void)hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hf, dfold, sizeof hf, hipMemcpyDeviceToHost);
    int bad = 0, over = 0;
    for (int l = 0; l < 64; l++) {
        uint32_t mh = 0, ml = 0;
        for (int i = 0; i < 16; i++) {
            const int q = l & 31, row = 8 * (i / 4) + 4 * (l >> 5) + (i % 4);
            const int sa = __builtin_popcountll(hc[row] & hq[q]) - __builtin_popcountll(hc[row] & ~hq[q]);
            const int sb = __builtin_popcountll(hc[32 + row] & hq[q]) - __builtin_popcountll(hc[32 + row] & ~hq[q]);
            const uint32_t want = 0x4B000000u + ((uint32_t)(64 + sb) << 16) + (uint32_t)(1088 + sa);
            if (sb == 64) { over++; printf("lane %d i %d: s_b = 64 (field overflow): got %08x, plain sum would be %08x\n", l, i, ho[l * 16 + i], want); continue; }
            if (ho[l * 16 + i] != want) { if (bad < 8) printf("lane %d i %d: got %08x want %08x\n", l, i, ho[l * 16 + i], want); bad++; }
            if (i < 5) { mh = mh > (want >> 16) ? mh : (want >> 16); ml = ml > (want & 0xffff) ? ml : (want & 0xffff); }
        }
        bool skip = false;
        for (int i = 0; i < 5; i++) {
            const int q = l & 31, row = 8 * (i / 4) + 4 * (l >> 5) + (i % 4);
            if (__builtin_popcountll(hc[32 + row] & hq[q]) - __builtin_popcountll(hc[32 + row] & ~hq[q]) == 64) skip = true;
        }
        if (!skip && hf[l] != ((mh << 16) | ml)) { if (bad < 8) printf("lane %d fold: got %08x want %08x\n", l, hf[l], (mh << 16) | ml); bad++; }
    }
    printf("fp4 packed-pair probe: %s (%d mismatches, %d overflow cases shown)\n", bad ? "MISMATCH" : "ok", bad, over);
    return bad != 0;
}
