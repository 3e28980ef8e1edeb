// Layout probe for v_mfma_f32_4x4x1_16B_f32 on gfx950: block (la, lb) sets A = 1 in lane la, B = 1 in lane lb and
// reports where D is non-zero (register r, lane l).  hipcc --offload-arch=gfx950 tools/probe_mfma4x4.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int* out) {
    const int la = blockIdx.x, lb = blockIdx.y, lane = threadIdx.x;
    const float a = lane == la ? 1.f : 0.f, b = lane == lb ? 1.f : 0.f;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++)
        if (c[r] != 0.f) atomicAdd(&out[(la * 64 + lb) * 2], 1), out[(la * 64 + lb) * 2 + 1] = r * 64 + lane;
}
int main() {
    int* d;
    hipMalloc(&d, 64 * 64 * 2 * 4);
    hipMemset(d, 0, 64 * 64 * 2 * 4);
    hipLaunchKernelGGL(probe, dim3(64, 64), dim3(64), 0, 0, d);
    std::vector<int> h(64 * 64 * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; la += 1)
        for (int lb = 0; lb < 64; lb++)
            if (h[(la * 64 + lb) * 2]) printf("A lane %2d  B lane %2d -> hits %d  D reg %d lane %2d\n", la, lb, h[(la * 64 + lb) * 2], h[(la * 64 + lb) * 2 + 1] / 64, h[(la * 64 + lb) * 2 + 1] % 64);
    return 0;
}
