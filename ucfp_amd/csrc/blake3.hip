// blake3.hip -- BLAKE3 (default hash mode, 32-byte output) of a batch of byte strings on the GPU.
//
// imgfprint records begin with `exact` = BLAKE3 of the UPLOADED bytes (AlgorithmView.svelte:30-33).  When the
// uploads are already on the device (the PNG front end) hashing them there spares the host a pass over
// every byte.  Written from the BLAKE3 specification; blake3_host.cpp (checked against the official test vectors in
// tests/test_abi.py) is the host statement of the same function and the checker in tests/test_blake3_gpu.py.
//
// One wave per input.  The 1024-byte chunks of an input are independent: lane j compresses chunks j, j + 64, ... (16
// sequential block compressions each) into chaining values; then the binary tree is folded level by level -- adjacent
// pairs become parents, an odd last node moves up unchanged, which is exactly BLAKE3's left-full tree -- with the lanes
// taking pairs.  Chaining values live in a workspace slice of the batch (32 bytes per chunk).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ucfp_hip.h"
#include "common.h"

namespace ucfp {

namespace {

constexpr uint32_t kChunkStart = 1, kChunkEnd = 2, kParent = 4, kRoot = 8;

struct Iv {
    static constexpr uint32_t v[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                      0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
};

// message word order of round r: the permutation 2 6 3 10 7 0 4 13 1 11 12 5 9 14 15 8 applied r times
struct Schedule {
    uint8_t s[7][16];
    constexpr Schedule() : s() {
        constexpr uint8_t perm[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
        for (int i = 0; i < 16; i++) s[0][i] = (uint8_t)i;
        for (int r = 1; r < 7; r++)
            for (int i = 0; i < 16; i++) s[r][i] = s[r - 1][perm[i]];
    }
};
constexpr Schedule kSched{};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }

#define UCFP_B3_G(a, b, c, d, mx, my) \
    a = a + b + (mx);                 \
    d = rotr(d ^ a, 16);              \
    c = c + d;                        \
    b = rotr(b ^ c, 12);              \
    a = a + b + (my);                 \
    d = rotr(d ^ a, 8);               \
    c = c + d;                        \
    b = rotr(b ^ c, 7);

// cv <- first half of compress(cv, m, counter, block_len, flags)
__device__ __forceinline__ void compress(uint32_t (&cv)[8], const uint32_t (&m)[16], uint64_t counter, uint32_t block_len,
                                         uint32_t flags) {
    uint32_t v0 = cv[0], v1 = cv[1], v2 = cv[2], v3 = cv[3], v4 = cv[4], v5 = cv[5], v6 = cv[6], v7 = cv[7];
    uint32_t v8 = Iv::v[0], v9 = Iv::v[1], v10 = Iv::v[2], v11 = Iv::v[3];
    uint32_t v12 = (uint32_t)counter, v13 = (uint32_t)(counter >> 32), v14 = block_len, v15 = flags;
#pragma unroll
    for (int r = 0; r < 7; r++) {
        UCFP_B3_G(v0, v4, v8, v12, m[kSched.s[r][0]], m[kSched.s[r][1]])
        UCFP_B3_G(v1, v5, v9, v13, m[kSched.s[r][2]], m[kSched.s[r][3]])
        UCFP_B3_G(v2, v6, v10, v14, m[kSched.s[r][4]], m[kSched.s[r][5]])
        UCFP_B3_G(v3, v7, v11, v15, m[kSched.s[r][6]], m[kSched.s[r][7]])
        UCFP_B3_G(v0, v5, v10, v15, m[kSched.s[r][8]], m[kSched.s[r][9]])
        UCFP_B3_G(v1, v6, v11, v12, m[kSched.s[r][10]], m[kSched.s[r][11]])
        UCFP_B3_G(v2, v7, v8, v13, m[kSched.s[r][12]], m[kSched.s[r][13]])
        UCFP_B3_G(v3, v4, v9, v14, m[kSched.s[r][14]], m[kSched.s[r][15]])
    }
    cv[0] = v0 ^ v8, cv[1] = v1 ^ v9, cv[2] = v2 ^ v10, cv[3] = v3 ^ v11;
    cv[4] = v4 ^ v12, cv[5] = v5 ^ v13, cv[6] = v6 ^ v14, cv[7] = v7 ^ v15;
}

// Chaining value of chunk `c` of the input p[0 .. len) (little-endian words, zero padding in the last block).
__device__ void chunk_cv(const uint8_t* p, uint64_t len, uint64_t c, bool root, uint32_t (&cv)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = Iv::v[i];
    const uint64_t off = c * 1024;
    const uint32_t cl = (uint32_t)(len - off < 1024 ? len - off : 1024);
    const uint32_t nblocks = cl == 0 ? 1 : (cl + 63) / 64;
    for (uint32_t b = 0; b < nblocks; b++) {
        const uint32_t bl = cl - b * 64 < 64 ? cl - b * 64 : 64;
        // the block as aligned words around it, funnel-shifted into place; bytes past the input are never requested
        const uintptr_t a = reinterpret_cast<uintptr_t>(p) + off + b * 64;
        const uint32_t mis = (uint32_t)(a & 3);
        const uint32_t* w = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
        uint32_t raw[17];
#pragma unroll
        for (int i = 0; i < 17; i++) raw[i] = (4u * i < bl + mis) ? w[i] : 0u;
        uint32_t m[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint32_t x = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], mis);
            const uint32_t have = bl > 4u * i ? bl - 4u * i : 0u;             // bytes of this word inside the block
            if (have < 4) x &= have == 0 ? 0u : (0xffffffffu >> (32 - 8 * have));
            m[i] = x;
        }
        uint32_t flags = 0;
        if (b == 0) flags |= kChunkStart;
        if (b == nblocks - 1) flags |= kChunkEnd | (root ? kRoot : 0u);
        compress(cv, m, c, bl, flags);
    }
}

// One wave per input.  cvs: workspace; input i owns the slice of 8-word entries starting at floor(offsets[i] / 1024) + i
// (at least as long as its chunk count, and disjoint from its neighbours').
__global__ __launch_bounds__(64) void blake3_batch_kernel(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                                         size_t n, uint32_t* __restrict__ cvs, uint8_t* __restrict__ out) {
    const size_t i = blockIdx.x;
    if (i >= n) return;
    const int lane = threadIdx.x;
    const uint8_t* p = blob + offsets[i];
    const uint64_t len = offsets[i + 1] - offsets[i];
    const uint64_t nchunks = len == 0 ? 1 : (len + 1023) / 1024;
    uint32_t* my = cvs + (offsets[i] / 1024 + i) * 8;
    uint32_t cv[8];
    if (nchunks == 1) {
        if (lane == 0) {
            chunk_cv(p, len, 0, true, cv);
#pragma unroll
            for (int k = 0; k < 8; k++) reinterpret_cast<uint32_t*>(out + i * 32)[k] = cv[k];   // little-endian words = the digest bytes
        }
        return;
    }
    for (uint64_t c = lane; c < nchunks; c += 64) {
        chunk_cv(p, len, c, false, cv);
#pragma unroll
        for (int k = 0; k < 8; k++) my[c * 8 + k] = cv[k];
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    uint64_t cnt = nchunks;
    while (cnt > 1) {
        const uint64_t pairs = cnt / 2;
        const bool root = cnt == 2;
        for (uint64_t b0 = 0; b0 < pairs; b0 += 64) {          // a batch of 64 pairs: read both children, then write the parents
            const uint64_t pi = b0 + lane;
            uint32_t m[16];
            if (pi < pairs) {
#pragma unroll
                for (int k = 0; k < 16; k++) m[k] = my[pi * 16 + k];
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            if (pi < pairs) {
#pragma unroll
                for (int k = 0; k < 8; k++) cv[k] = Iv::v[k];
                compress(cv, m, 0, 64, kParent | (root ? kRoot : 0u));
#pragma unroll
                for (int k = 0; k < 8; k++) my[pi * 8 + k] = cv[k];
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
        }
        if (cnt & 1) {                                           // the odd last node moves up unchanged
            uint32_t t = 0;
            if (lane < 8) t = my[(cnt - 1) * 8 + lane];
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            if (lane < 8) my[pairs * 8 + lane] = t;
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
        }
        cnt = pairs + (cnt & 1);
    }
    if (lane < 8) reinterpret_cast<uint32_t*>(out + i * 32)[lane] = my[lane];
}

}  // namespace

size_t blake3_ws_bytes(size_t n, size_t blob_bytes) { return (blob_bytes / 1024 + n + 2) * 32; }

int launch_blake3_batch(const uint8_t* blob, const uint64_t* offsets, size_t n, uint8_t* ws, uint8_t* out, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(blake3_batch_kernel, dim3((unsigned)n), dim3(64), 0, stream, blob, offsets, n,
                       reinterpret_cast<uint32_t*>(ws), out);
    return 0;
}

}  // namespace ucfp
