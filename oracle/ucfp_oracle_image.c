/*
 * ucfp_oracle_image.c -- CPU restatement of the image hot path.  TEST INFRASTRUCTURE ONLY:
 * may be called from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
 * never from the product path (ucfp_amd/).
 *
 * PARITY UNPINNED.  The reference delegates this arithmetic to the crates.io package
 * `imgfprint 0.4.1` (Cargo.lock:1863-1866), whose source is not under /root/reference and
 * not available offline.  The reference's own tests pin only the record LENGTHS (536 B
 * bundle src/server/tests.rs:1206; 168 B single web/.../algorithmView.ts:11-17) and the
 * layout (web/.../ImageHashView.svelte:2-5, AlgorithmView.svelte:30-37).  This file restates
 * the published construction (normalise -> global hash + 4x4 block hashes; aHash 8x8 vs mean,
 * dHash 9x8 left>right, pHash 32x32 DCT-II low 8x8 vs median: REPORT.md:764-824,
 * src/modality/image.rs:265-270,310-318) and fixes every choice the reference leaves open
 * (DESIGN.md "Image spec").  Bit-exactness claims are GPU-vs-this-file.
 *
 * Written for clarity, not speed: every resample goes through one general exact-integer
 * area filter, so the HIP kernel's hierarchical partial sums are checked against an
 * independent formulation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/ucfp_dct32.h"

#define NORM 256

static const float DCT_LO[8][32] = UCFP_DCT32_LO_INIT;

/* I1: luma of one pixel. GRAY8: the byte. RGB8/RGBA8: BT.601 integer, weights sum 256. */
static inline uint8_t luma_at(const uint8_t* row, uint32_t x, int pixfmt) {
    if (pixfmt == 0) return row[x];
    const uint8_t* p = row + (size_t)x * (pixfmt == 1 ? 3 : 4);
    return (uint8_t)((77u * p[0] + 150u * p[1] + 29u * p[2] + 128u) >> 8);
}

/* overlap of source cell s (covering [dn*s, dn*s+dn)) with dest cell d (covering
 * [sn*d, sn*d+sn)) on the common lattice of length sn*dn; sn = source count, dn = dest count */
static inline uint64_t overlap(uint32_t s, uint32_t d, uint32_t sn, uint32_t dn) {
    uint64_t s0 = (uint64_t)dn * s, s1 = s0 + dn;
    uint64_t d0 = (uint64_t)sn * d, d1 = d0 + sn;
    uint64_t lo = s0 > d0 ? s0 : d0, hi = s1 < d1 ? s1 : d1;
    return hi > lo ? hi - lo : 0;
}

/* I3/I5: exact-integer area resample of a u8 plane (sw x sh, row stride `stride`) to
 * dw x dh.  out = floor((2*sum + D) / (2*D)), D = sw*sh  (round half up). */
static void area_resample(const uint8_t* src, size_t stride, uint32_t sw, uint32_t sh,
                          uint8_t* dst, uint32_t dw, uint32_t dh) {
    const uint64_t D = (uint64_t)sw * sh;
    for (uint32_t j = 0; j < dh; j++) {
        uint32_t y0 = (uint32_t)(((uint64_t)sh * j) / dh);
        uint32_t y1 = (uint32_t)(((uint64_t)sh * (j + 1) + dh - 1) / dh);
        for (uint32_t i = 0; i < dw; i++) {
            uint32_t x0 = (uint32_t)(((uint64_t)sw * i) / dw);
            uint32_t x1 = (uint32_t)(((uint64_t)sw * (i + 1) + dw - 1) / dw);
            uint64_t acc = 0;
            for (uint32_t y = y0; y < y1 && y < sh; y++) {
                uint64_t wy = overlap(y, j, sh, dh);
                if (!wy) continue;
                uint64_t racc = 0;
                for (uint32_t x = x0; x < x1 && x < sw; x++)
                    racc += overlap(x, i, sw, dw) * src[(size_t)y * stride + x];
                acc += wy * racc;
            }
            dst[(size_t)j * dw + i] = (uint8_t)((2 * acc + D) / (2 * D));
        }
    }
}

/* I6 */
static uint64_t ahash_region(const uint8_t* reg, size_t stride, uint32_t rs) {
    uint8_t g[64];
    area_resample(reg, stride, rs, rs, g, 8, 8);
    uint32_t sum = 0;
    for (int i = 0; i < 64; i++) sum += g[i];
    uint32_t mean = sum / 64; /* integer mean, as src/modality/image.rs:317-318 */
    uint64_t h = 0;
    for (int i = 0; i < 64; i++)
        if (g[i] > mean) h |= 1ull << i;
    return h;
}

/* I7 */
static uint64_t dhash_region(const uint8_t* reg, size_t stride, uint32_t rs) {
    uint8_t g[72];
    area_resample(reg, stride, rs, rs, g, 9, 8);
    uint64_t h = 0;
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++)
            if (g[r * 9 + c] > g[r * 9 + c + 1]) h |= 1ull << (r * 8 + c);
    return h;
}

/* I8: coefficients of the low 8x8 block, row-major i = 8*v + u. */
static void phash_coefs(const uint8_t g[1024], float coef[64]) {
    float P[32][8]; /* P[y][u] = sum_x g[y][x] * C[u][x], fmaf chain x ascending from +0 */
    for (int y = 0; y < 32; y++)
        for (int u = 0; u < 8; u++) {
            float acc = 0.0f;
            for (int x = 0; x < 32; x++) acc = fmaf((float)g[y * 32 + x], DCT_LO[u][x], acc);
            P[y][u] = acc;
        }
    for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++) {
            float acc = 0.0f;
            for (int y = 0; y < 32; y++) acc = fmaf(DCT_LO[v][y], P[y][u], acc);
            coef[v * 8 + u] = acc;
        }
}

static int cmp_f32(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

static uint64_t phash_region(const uint8_t* reg, size_t stride, uint32_t rs) {
    uint8_t g[1024];
    float coef[64], ac[63];
    area_resample(reg, stride, rs, rs, g, 32, 32);
    phash_coefs(g, coef);
    memcpy(ac, coef + 1, sizeof ac);
    qsort(ac, 63, sizeof(float), cmp_f32);
    float med = ac[31]; /* 32nd smallest of the 63 AC coefficients */
    uint64_t h = 0;
    for (int i = 0; i < 64; i++)
        if (coef[i] > med) h |= 1ull << i;
    return h;
}

static void put_u64(uint8_t* p, uint64_t v) {
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}

/* I3: normalise one frame to 256x256 luma. Exposed for stage-level tests. */
void ucfp_oracle_image_normalize(const uint8_t* frame, uint32_t w, uint32_t h, size_t row_stride,
                                 int pixfmt, uint8_t* norm /* 256*256 */) {
    uint8_t* luma = (uint8_t*)malloc((size_t)w * h);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++)
            luma[(size_t)y * w + x] = luma_at(frame + (size_t)y * row_stride, x, pixfmt);
    area_resample(luma, w, w, h, norm, NORM, NORM);
    free(luma);
}

/* which: 1 = ahash, 2 = phash, 4 = dhash. hashes[0] = global, hashes[1..16] = blocks row-major. */
void ucfp_oracle_image_hashes17(const uint8_t* norm, int which, uint64_t hashes[17]) {
    for (int r = 0; r < 17; r++) {
        const uint8_t* reg = norm;
        uint32_t rs = NORM;
        if (r > 0) {
            int by = (r - 1) / 4, bx = (r - 1) % 4;
            reg = norm + (size_t)by * 64 * NORM + bx * 64;
            rs = 64;
        }
        hashes[r] = which == 1   ? ahash_region(reg, NORM, rs)
                    : which == 2 ? phash_region(reg, NORM, rs)
                                 : dhash_region(reg, NORM, rs);
    }
}

/* Stage probes for tests: 32x32 gray and DCT coefficients of region r. */
void ucfp_oracle_image_region_gray32(const uint8_t* norm, int r, uint8_t g[1024]) {
    const uint8_t* reg = norm;
    uint32_t rs = NORM;
    if (r > 0) {
        int by = (r - 1) / 4, bx = (r - 1) % 4;
        reg = norm + (size_t)by * 64 * NORM + bx * 64;
        rs = 64;
    }
    area_resample(reg, NORM, rs, rs, g, 32, 32);
}
void ucfp_oracle_image_phash_coefs(const uint8_t g[1024], float coef[64]) { phash_coefs(g, coef); }

static void write_fp(uint8_t* out, const uint8_t* exact, const uint64_t hs[17]) {
    if (exact) memcpy(out, exact, 32);
    else memset(out, 0, 32);
    for (int r = 0; r < 17; r++) put_u64(out + 32 + 8 * r, hs[r]);
}

/* Batch entry mirroring ucfp_image_hash_batch. Returns 0; status[i] = 0 / -1. */
int ucfp_oracle_image_hash_batch(uint32_t algo, const uint8_t* frames, size_t n, uint32_t w,
                                 uint32_t h, size_t row_stride, size_t frame_stride, int pixfmt,
                                 uint32_t min_dim, uint32_t max_dim, const uint8_t* exact,
                                 uint8_t* out, int32_t* status) {
    size_t rec = algo == 7 ? 536 : 168;
    int bad = (w < min_dim || h < min_dim || w > max_dim || h > max_dim);
    /* frames are independent: OpenMP over the batch for the timed CPU baseline */
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t i = 0; i < n; i++) {
        uint8_t* norm = (uint8_t*)malloc(NORM * NORM);
        uint8_t* o = out + i * rec;
        if (bad) {
            memset(o, 0, rec);
            if (status) status[i] = -1;
            free(norm);
            continue;
        }
        if (status) status[i] = 0;
        const uint8_t* ex = exact ? exact + 32 * i : NULL;
        ucfp_oracle_image_normalize(frames + i * frame_stride, w, h, row_stride, pixfmt, norm);
        uint64_t hs[17];
        if (algo == 7) {
            /* MultiHashFingerprint: exact | ahash | phash | dhash
             * (web/src/lib/components/charts/AlgorithmView.svelte:30-37) */
            if (ex) memcpy(o, ex, 32);
            else memset(o, 0, 32);
            ucfp_oracle_image_hashes17(norm, 1, hs);
            write_fp(o + 32, ex, hs);
            ucfp_oracle_image_hashes17(norm, 2, hs);
            write_fp(o + 32 + 168, ex, hs);
            ucfp_oracle_image_hashes17(norm, 4, hs);
            write_fp(o + 32 + 336, ex, hs);
        } else {
            ucfp_oracle_image_hashes17(norm, (int)algo, hs);
            write_fp(o, ex, hs);
        }
        free(norm);
    }
    return 0;
}

/* Synthetic workload generator, identical to ucfp_image_synth_dev. */
static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void ucfp_oracle_image_synth(uint8_t* frames, size_t n, uint32_t w, uint32_t h, size_t first) {
#pragma omp parallel for
    for (size_t k = 0; k < n; k++) {
        uint64_t idx = first + k;
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t x = 0; x < w; x++) {
                uint64_t pix = (idx * h + y) * w + x;
                uint8_t ramp = (uint8_t)((x + y + 17 * idx) & 255);
                frames[(k * h + y) * (size_t)w + x] = ramp ^ (uint8_t)(mix64(pix) >> 60);
            }
    }
}

int ucfp_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void ucfp_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
