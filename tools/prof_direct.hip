// Phase timing of the single-launch Hamming search (ucfp_amd/csrc/hamming_direct.hip built with -DUCFP_DIR_PROF):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DUCFP_DIR_PROF tools/prof_direct.hip -o tools/prof_direct.bin
//   tools/prof_direct.bin [n = 12500000] [nq = 1] [k = 10]
// Every workgroup stamps the 100 MHz clock at: 0 entry, 1 first loads issued, 2 stream done (after the barrier),
// 3 workgroup merge done, 4 published entries drained, 5 ticket known, 6 (last workgroup) results written.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../ucfp_amd/csrc/hamming_direct.hip"

__global__ void fill(uint64_t* codes, uint64_t* ids, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = 0x9E3779B97F4A7C15ull * (i + 1);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    codes[i] = x;
    ids[i] = i;
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 12500000;
    const uint32_t nq = argc > 2 ? atoi(argv[2]) : 1, k = argc > 3 ? atoi(argv[3]) : 10;
    const bool asc = argc > 4 ? atoi(argv[4]) != 0 : false;   // 1: the shard's ids ascend with the row (rows are the tie-break keys)
    uint64_t *codes, *ids, *q, *out_ids;
    uint32_t *out_d, *out_cnt;
    float* out_sc;
    uint8_t* state;
    hipMalloc(&codes, n * 8);
    hipMalloc(&ids, n * 8);
    hipMalloc(&q, 64);
    hipMalloc(&out_ids, nq * k * 8);
    hipMalloc(&out_d, nq * k * 4);
    hipMalloc(&out_sc, nq * k * 4);
    hipMalloc(&out_cnt, nq * 4);
    const size_t sb = ucfp::hamming_direct_state_bytes();
    hipMalloc(&state, sb);
    hipMemset(state, 0, sb);
    uint32_t* flag;
    hipMalloc(&flag, 16);
    const uint32_t one[4] = {1, 0, 0, 0};
    hipMemcpy(flag, one, 16, hipMemcpyHostToDevice);
    const uint32_t* ascp = asc ? flag : nullptr;
    fill<<<(unsigned)((n + 255) / 256), 256>>>(codes, ids, n);
    uint64_t hq[8];
    for (int j = 0; j < 8; j++) hq[j] = 0x0123456789ABCDEFull * (2 * j + 1) ^ (0xF00Dull << (7 * j));
    hipMemcpy(q, hq, 64, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) ucfp::launch_hamming_direct(codes, ids, n, q, nq, k, state, out_ids, out_d, out_sc, out_cnt, 0, ascp);
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) ucfp::launch_hamming_direct(codes, ids, n, q, nq, k, state, out_ids, out_d, out_sc, out_cnt, 0, ascp);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("n %zu nq %u k %u asc %d: %.2f us per search back to back (%.2f TB/s of codes)\n", n, nq, k, (int)asc, ms / reps * 1e3,
           n * 8.0 / (ms / reps * 1e-3) / 1e12);
    // one isolated launch for the stamps
    hipDeviceSynchronize();
    ucfp::launch_hamming_direct(codes, ids, n, q, nq, k, state, out_ids, out_d, out_sc, out_cnt, 0, ascp);
    hipDeviceSynchronize();
    std::vector<uint64_t> st(256 * 8);
    hipMemcpy(st.data(), state + ucfp::kHammingDirectZeroBytes + (size_t)ucfp::kHammingDirectMaxQ * 256 * ucfp::kHammingDirectMaxK * 12, 256 * 8 * 8,
              hipMemcpyDeviceToHost);
    const size_t ntrips = (n + 511) / 512;
    size_t G = (ntrips + 15) / 16;
    if (G > 256) G = 256;
    uint64_t t0 = ~0ull;
    for (size_t g = 0; g < G; g++) t0 = std::min(t0, st[g * 8]);
    const char* names[7] = {"entry", "loads issued", "stream done", "wg merge done", "drained", "ticket known", "final done"};
    for (int s = 0; s < 7; s++) {
        uint64_t lo = ~0ull, hi = 0;
        double sum = 0;
        int cnt = 0;
        for (size_t g = 0; g < G; g++) {
            const uint64_t v = st[g * 8 + s];
            if (s == 6 && v < st[g * 8 + 5]) continue;   // only the last workgroup writes stamp 6 (stale otherwise)
            lo = std::min(lo, v);
            hi = std::max(hi, v);
            sum += (double)(v - t0);
            cnt++;
        }
        if (cnt) printf("  %-14s first %7.2f us   mean %7.2f us   last %7.2f us   (%d workgroups)\n", names[s], (lo - t0) * 0.01,
                        sum / cnt * 0.01, (hi - t0) * 0.01, cnt);
    }
    uint32_t hd[32];
    hipMemcpy(hd, out_d, std::min<size_t>(nq * k, 32) * 4, hipMemcpyDeviceToHost);
    printf("  best distances of query 0:");
    for (uint32_t r = 0; r < k && r < 12; r++) printf(" %u", hd[r]);
    printf("\n");
    return 0;
}
