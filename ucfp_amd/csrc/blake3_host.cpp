// blake3_host.cpp -- portable BLAKE3 (default hash mode, 32-byte output) for the `exact`
// field of imgfprint records: the reference stores BLAKE3 of the uploaded image bytes in
// ImageFingerprint.exact / MultiHashFingerprint.exact (AlgorithmView.svelte:30-33).
// The upload lives on the host, so this is host code; written from the BLAKE3 specification.

#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../../include/ucfp_hip.h"

namespace {

constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                            0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
constexpr uint8_t PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum : uint32_t { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

inline void g(uint32_t* v, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    v[a] = v[a] + v[b] + mx;
    v[d] = rotr(v[d] ^ v[a], 16);
    v[c] = v[c] + v[d];
    v[b] = rotr(v[b] ^ v[c], 12);
    v[a] = v[a] + v[b] + my;
    v[d] = rotr(v[d] ^ v[a], 8);
    v[c] = v[c] + v[d];
    v[b] = rotr(v[b] ^ v[c], 7);
}

// out[0..7] = new chaining value (first half of the compression output)
void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len,
              uint32_t flags, uint32_t out[8]) {
    uint32_t v[16], m[16], t[16];
    for (int i = 0; i < 8; i++) v[i] = cv[i];
    for (int i = 0; i < 4; i++) v[8 + i] = IV[i];
    v[12] = (uint32_t)counter;
    v[13] = (uint32_t)(counter >> 32);
    v[14] = block_len;
    v[15] = flags;
    memcpy(m, block, sizeof m);
    for (int r = 0; r < 7; r++) {
        g(v, 0, 4, 8, 12, m[0], m[1]);
        g(v, 1, 5, 9, 13, m[2], m[3]);
        g(v, 2, 6, 10, 14, m[4], m[5]);
        g(v, 3, 7, 11, 15, m[6], m[7]);
        g(v, 0, 5, 10, 15, m[8], m[9]);
        g(v, 1, 6, 11, 12, m[10], m[11]);
        g(v, 2, 7, 8, 13, m[12], m[13]);
        g(v, 3, 4, 9, 14, m[14], m[15]);
        if (r < 6) {
            for (int i = 0; i < 16; i++) t[i] = m[PERM[i]];
            memcpy(m, t, sizeof m);
        }
    }
    for (int i = 0; i < 8; i++) out[i] = v[i] ^ v[i + 8];
}

inline void load_block(const uint8_t* p, size_t len, uint32_t w[16]) {
    uint8_t buf[64];
    memset(buf, 0, 64);
    memcpy(buf, p, len);
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) |
               ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
}

// Chaining value of one chunk (<= 1024 bytes); `root` marks a single-chunk input.
void chunk_cv(const uint8_t* p, size_t len, uint64_t chunk_index, bool root, uint32_t out[8]) {
    uint32_t cv[8];
    memcpy(cv, IV, sizeof cv);
    size_t nblocks = len == 0 ? 1 : (len + 63) / 64;
    for (size_t b = 0; b < nblocks; b++) {
        size_t off = b * 64;
        size_t bl = len - off < 64 ? len - off : 64;
        if (len == 0) bl = 0;
        uint32_t w[16];
        load_block(p + off, bl, w);
        uint32_t flags = 0;
        if (b == 0) flags |= CHUNK_START;
        if (b == nblocks - 1) flags |= CHUNK_END | (root ? ROOT : 0);
        compress(cv, w, chunk_index, (uint32_t)bl, flags, cv);
    }
    memcpy(out, cv, sizeof cv);
}

void parent_cv(const uint32_t l[8], const uint32_t r[8], bool root, uint32_t out[8]) {
    uint32_t w[16];
    memcpy(w, l, 32);
    memcpy(w + 8, r, 32);
    compress(IV, w, 0, 64, PARENT | (root ? ROOT : 0), out);
}

}  // namespace

extern "C" int ucfp_blake3(const uint8_t* data, size_t len, uint8_t out[32]) {
    if (!out || (len && !data)) return UCFP_E_INVALID;
    uint32_t stack[64][8];
    int sp = 0;
    uint32_t cv[8];
    const size_t nchunks = len == 0 ? 1 : (len + 1023) / 1024;
    if (nchunks == 1) {
        chunk_cv(data, len, 0, true, cv);
    } else {
        for (size_t c = 0; c < nchunks; c++) {
            size_t off = c * 1024;
            size_t cl = len - off < 1024 ? len - off : 1024;
            chunk_cv(data + off, cl, c, false, cv);
            if (c == nchunks - 1) break;  // last chunk: merged below with ROOT handling
            // merge completed subtrees: one merge per trailing one-bit of the chunk count so far
            uint64_t total = c + 1;
            while ((total & 1) == 0) {
                parent_cv(stack[--sp], cv, false, cv);
                total >>= 1;
            }
            memcpy(stack[sp++], cv, 32);
        }
        // fold the stack right-to-left; the final merge carries ROOT
        while (sp > 0) {
            parent_cv(stack[sp - 1], cv, sp == 1, cv);
            sp--;
        }
    }
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)cv[i];
        out[4 * i + 1] = (uint8_t)(cv[i] >> 8);
        out[4 * i + 2] = (uint8_t)(cv[i] >> 16);
        out[4 * i + 3] = (uint8_t)(cv[i] >> 24);
    }
    return UCFP_OK;
}
