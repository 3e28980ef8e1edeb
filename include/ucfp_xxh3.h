/*
 * ucfp_xxh3.h -- XXH3_64bits (seed 0, default secret) written from the published xxHash
 * specification, usable from C (oracle), C++ host code and HIP device code.
 *
 * The reference's text SDK hashes shingles/tokens with the Xxh3_64 family
 * (src/server/tests.rs:1126-1127; xxhash-rust 0.8.15 in Cargo.lock:6144).  Both the CPU oracle and
 * the HIP kernels include this one header, and tests/test_oracle_spec.py checks it against the
 * independent `xxhash` Python module for every length class (0, 1-3, 4-8, 9-16, 17-128, 129-240,
 * > 240 bytes).
 *
 * Input access goes through the macro UCFP_XXH3_BYTE(p, i) so a caller can hash from any byte
 * source (plain memory by default; the HIP text kernel reads LDS).
 */
#ifndef UCFP_XXH3_H
#define UCFP_XXH3_H

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
/* device-only in HIP translation units: the secret lives in constant memory */
#define UCFP_XXH3_FN __device__ static inline __attribute__((always_inline))
#define UCFP_XXH3_CONST __constant__ const
#else
#define UCFP_XXH3_FN static inline
#define UCFP_XXH3_CONST static const
#endif

#define UCFP_XXH_PRIME32_1 0x9E3779B1u
#define UCFP_XXH_PRIME32_2 0x85EBCA77u
#define UCFP_XXH_PRIME32_3 0xC2B2AE3Du
#define UCFP_XXH_PRIME64_1 0x9E3779B185EBCA87ull
#define UCFP_XXH_PRIME64_2 0xC2B2AE3D27D4EB4Full
#define UCFP_XXH_PRIME64_3 0x165667B19E3779F9ull
#define UCFP_XXH_PRIME64_4 0x85EBCA77C2B2AE63ull
#define UCFP_XXH_PRIME64_5 0x27D4EB2F165667C5ull
#define UCFP_XXH_PRIME_MX1 0x165667919E3779F9ull
#define UCFP_XXH_PRIME_MX2 0x9FB21C651E98DF25ull

/* the 192-byte default secret as 24 little-endian u64 words at byte offsets 0, 8, ... */
#define UCFP_XXH3_SECRET_BYTES                                                                     \
    {0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c, \
     0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f, \
     0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21, \
     0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c, \
     0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3, \
     0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8, \
     0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d, \
     0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64, \
     0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb, \
     0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e, \
     0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce, \
     0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e}

UCFP_XXH3_CONST uint8_t ucfp_xxh3_secret[192] = UCFP_XXH3_SECRET_BYTES;

UCFP_XXH3_FN uint64_t ucfp_xxh_sec64(int off) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | ucfp_xxh3_secret[off + i];
    return v;
}
UCFP_XXH3_FN uint32_t ucfp_xxh_sec32(int off) {
    uint32_t v = 0;
    for (int i = 3; i >= 0; i--) v = (v << 8) | ucfp_xxh3_secret[off + i];
    return v;
}

UCFP_XXH3_FN uint64_t ucfp_xxh_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
UCFP_XXH3_FN uint64_t ucfp_xxh_swap64(uint64_t x) {
    x = ((x & 0x00ff00ff00ff00ffull) << 8) | ((x >> 8) & 0x00ff00ff00ff00ffull);
    x = ((x & 0x0000ffff0000ffffull) << 16) | ((x >> 16) & 0x0000ffff0000ffffull);
    return (x << 32) | (x >> 32);
}
UCFP_XXH3_FN uint64_t ucfp_xxh_mul128_fold64(uint64_t a, uint64_t b) {
    /* 64x64 -> 128 by 32-bit limbs (portable: device code has no __int128) */
    const uint64_t a0 = (uint32_t)a, a1 = a >> 32, b0 = (uint32_t)b, b1 = b >> 32;
    const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
    const uint64_t mid = (p00 >> 32) + (uint32_t)p01 + (uint32_t)p10;
    const uint64_t lo = (mid << 32) | (uint32_t)p00;
    const uint64_t hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
    return lo ^ hi;
}
UCFP_XXH3_FN uint64_t ucfp_xxh64_avalanche(uint64_t h) {
    h ^= h >> 33;
    h *= UCFP_XXH_PRIME64_2;
    h ^= h >> 29;
    h *= UCFP_XXH_PRIME64_3;
    h ^= h >> 32;
    return h;
}
UCFP_XXH3_FN uint64_t ucfp_xxh3_avalanche(uint64_t h) {
    h ^= h >> 37;
    h *= UCFP_XXH_PRIME_MX1;
    h ^= h >> 32;
    return h;
}

/* Generic body: RD8(i) must yield input byte i as an integer. */
/* UCFP_XXH3_DEFINE = the byte-assembling little-endian readers + the hash body.  A caller with a faster way to
 * read unaligned words (e.g. aligned LDS dwords + v_alignbyte) defines NAME_rd64 / NAME_rd32 itself and uses
 * UCFP_XXH3_DEFINE_BODY alone. */
#define UCFP_XXH3_DEFINE_READERS(NAME, SRC_T, RD8)                                                  \
    UCFP_XXH3_FN uint64_t NAME##_rd64(SRC_T src, size_t o) {                                        \
        uint64_t v = 0;                                                                             \
        for (int i = 7; i >= 0; i--) v = (v << 8) | (uint64_t)(RD8(src, o + (size_t)i));            \
        return v;                                                                                   \
    }                                                                                               \
    UCFP_XXH3_FN uint32_t NAME##_rd32(SRC_T src, size_t o) {                                        \
        uint32_t v = 0;                                                                             \
        for (int i = 3; i >= 0; i--) v = (v << 8) | (uint32_t)(RD8(src, o + (size_t)i));            \
        return v;                                                                                   \
    }
#define UCFP_XXH3_DEFINE(NAME, SRC_T, RD8)                                                          \
    UCFP_XXH3_DEFINE_READERS(NAME, SRC_T, RD8)                                                      \
    UCFP_XXH3_DEFINE_BODY(NAME, SRC_T, RD8)
#define UCFP_XXH3_DEFINE_BODY(NAME, SRC_T, RD8)                                                     \
    UCFP_XXH3_FN uint64_t NAME##_mix16(SRC_T src, size_t o, int so) {                               \
        return ucfp_xxh_mul128_fold64(NAME##_rd64(src, o) ^ ucfp_xxh_sec64(so),                     \
                                      NAME##_rd64(src, o + 8) ^ ucfp_xxh_sec64(so + 8));            \
    }                                                                                               \
    UCFP_XXH3_FN void NAME##_acc512(uint64_t* acc, SRC_T src, size_t o, int so) {                   \
        for (int i = 0; i < 8; i++) {                                                               \
            const uint64_t dv = NAME##_rd64(src, o + 8 * (size_t)i);                                \
            const uint64_t dk = dv ^ ucfp_xxh_sec64(so + 8 * i);                                    \
            acc[i ^ 1] += dv;                                                                       \
            acc[i] += (dk & 0xffffffffull) * (dk >> 32);                                            \
        }                                                                                           \
    }                                                                                               \
    UCFP_XXH3_FN uint64_t NAME(SRC_T src, size_t len) {                                             \
        if (len == 0) return ucfp_xxh64_avalanche(ucfp_xxh_sec64(56) ^ ucfp_xxh_sec64(64));         \
        if (len <= 3) {                                                                             \
            const uint32_t c1 = (uint32_t)(RD8(src, 0)), c2 = (uint32_t)(RD8(src, len >> 1)),       \
                           c3 = (uint32_t)(RD8(src, len - 1));                                      \
            const uint32_t comb = (c1 << 16) | (c2 << 24) | c3 | ((uint32_t)len << 8);              \
            const uint64_t flip = (uint64_t)(ucfp_xxh_sec32(0) ^ ucfp_xxh_sec32(4));                \
            return ucfp_xxh64_avalanche((uint64_t)comb ^ flip);                                     \
        }                                                                                           \
        if (len <= 8) {                                                                             \
            const uint32_t in1 = NAME##_rd32(src, 0), in2 = NAME##_rd32(src, len - 4);              \
            const uint64_t flip = ucfp_xxh_sec64(8) ^ ucfp_xxh_sec64(16);                           \
            const uint64_t in64 = (uint64_t)in2 + ((uint64_t)in1 << 32);                            \
            uint64_t h = in64 ^ flip;                                                               \
            h ^= ucfp_xxh_rotl64(h, 49) ^ ucfp_xxh_rotl64(h, 24);                                   \
            h *= UCFP_XXH_PRIME_MX2;                                                                \
            h ^= (h >> 35) + (uint64_t)len;                                                         \
            h *= UCFP_XXH_PRIME_MX2;                                                                \
            return h ^ (h >> 28);                                                                   \
        }                                                                                           \
        if (len <= 16) {                                                                            \
            const uint64_t f1 = ucfp_xxh_sec64(24) ^ ucfp_xxh_sec64(32);                            \
            const uint64_t f2 = ucfp_xxh_sec64(40) ^ ucfp_xxh_sec64(48);                            \
            const uint64_t lo = NAME##_rd64(src, 0) ^ f1, hi = NAME##_rd64(src, len - 8) ^ f2;      \
            const uint64_t acc = (uint64_t)len + ucfp_xxh_swap64(lo) + hi + ucfp_xxh_mul128_fold64(lo, hi); \
            return ucfp_xxh3_avalanche(acc);                                                        \
        }                                                                                           \
        if (len <= 128) {                                                                           \
            uint64_t acc = (uint64_t)len * UCFP_XXH_PRIME64_1;                                      \
            if (len > 32) {                                                                         \
                if (len > 64) {                                                                     \
                    if (len > 96) {                                                                 \
                        acc += NAME##_mix16(src, 48, 96);                                           \
                        acc += NAME##_mix16(src, len - 64, 112);                                    \
                    }                                                                               \
                    acc += NAME##_mix16(src, 32, 64);                                               \
                    acc += NAME##_mix16(src, len - 48, 80);                                         \
                }                                                                                   \
                acc += NAME##_mix16(src, 16, 32);                                                   \
                acc += NAME##_mix16(src, len - 32, 48);                                             \
            }                                                                                       \
            acc += NAME##_mix16(src, 0, 0);                                                         \
            acc += NAME##_mix16(src, len - 16, 16);                                                 \
            return ucfp_xxh3_avalanche(acc);                                                        \
        }                                                                                           \
        if (len <= 240) {                                                                           \
            uint64_t acc = (uint64_t)len * UCFP_XXH_PRIME64_1;                                      \
            const int rounds = (int)(len / 16);                                                     \
            for (int i = 0; i < 8; i++) acc += NAME##_mix16(src, 16 * (size_t)i, 16 * i);           \
            acc = ucfp_xxh3_avalanche(acc);                                                         \
            for (int i = 8; i < rounds; i++) acc += NAME##_mix16(src, 16 * (size_t)i, 16 * (i - 8) + 3); \
            acc += NAME##_mix16(src, len - 16, 136 - 17);                                           \
            return ucfp_xxh3_avalanche(acc);                                                        \
        }                                                                                           \
        {                                                                                           \
            uint64_t acc[8] = {UCFP_XXH_PRIME32_3, UCFP_XXH_PRIME64_1, UCFP_XXH_PRIME64_2,          \
                               UCFP_XXH_PRIME64_3, UCFP_XXH_PRIME64_4, UCFP_XXH_PRIME32_2,          \
                               UCFP_XXH_PRIME64_5, UCFP_XXH_PRIME32_1};                             \
            const size_t nb_blocks = (len - 1) / 1024;                                              \
            for (size_t n = 0; n < nb_blocks; n++) {                                                \
                for (int s = 0; s < 16; s++) NAME##_acc512(acc, src, n * 1024 + 64 * (size_t)s, 8 * s); \
                for (int i = 0; i < 8; i++) {                                                       \
                    uint64_t a = acc[i];                                                            \
                    a ^= a >> 47;                                                                   \
                    a ^= ucfp_xxh_sec64(128 + 8 * i);                                               \
                    a *= UCFP_XXH_PRIME32_1;                                                        \
                    acc[i] = a;                                                                     \
                }                                                                                   \
            }                                                                                       \
            const size_t nb_stripes = ((len - 1) - 1024 * nb_blocks) / 64;                          \
            for (size_t s = 0; s < nb_stripes; s++)                                                 \
                NAME##_acc512(acc, src, nb_blocks * 1024 + 64 * s, (int)(8 * s));                   \
            NAME##_acc512(acc, src, len - 64, 192 - 64 - 7);                                        \
            uint64_t r = (uint64_t)len * UCFP_XXH_PRIME64_1;                                        \
            for (int i = 0; i < 4; i++)                                                             \
                r += ucfp_xxh_mul128_fold64(acc[2 * i] ^ ucfp_xxh_sec64(11 + 16 * i),               \
                                            acc[2 * i + 1] ^ ucfp_xxh_sec64(11 + 16 * i + 8));      \
            return ucfp_xxh3_avalanche(r);                                                          \
        }                                                                                           \
    }

#define UCFP_XXH3_RD8_MEM(p, i) ((p)[(i)])
UCFP_XXH3_DEFINE(ucfp_xxh3_64, const uint8_t*, UCFP_XXH3_RD8_MEM)

#endif /* UCFP_XXH3_H */
