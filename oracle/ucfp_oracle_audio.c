/*
 * ucfp_oracle_audio.c -- CPU restatement of the audio hot path (Wang landmarks, Haitsma-Kalker).
 * TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The reference delegates to `audiofp 0.3.0` (Cargo.lock:209-212), absent from
 * /root/reference.  Pinned by the reference and honoured here: WangHash = 8 bytes
 * {u32 LE f_a(9)|f_b(9)|dt(14), u32 LE t_anchor} with the anchor frequency in bits 31..23
 * (web/.../LandmarkScatter.svelte:4,31-37); 62.5 frames/s at 8 kHz => hop 128, 512 frequency
 * buckets => n_fft 1024 (LandmarkScatter.svelte:9-10,25); Wang needs 8 kHz input
 * (src/modality/audio.rs:422-430); defaults fan_out 10, target_zone_t 63, target_zone_f 64,
 * peaks_per_sec 30, min_anchor_mag_db -50 (src/server/algorithms_manifest.rs:553-592); the
 * anchor->target pairing rule, restated in-tree at src/modality/audio.rs:965-1003; Haitsma:
 * linear resample to 5 kHz (audio.rs:194-200), one u32 per frame (audio.rs:208-209), 300-2000 Hz
 * (manifest :655-672), 312 B/s => hop 64 (manifest :654).  Everything else (window, framing, dB
 * reference, peak neighbourhood, tie rules, band edges) is fixed by DESIGN.md "Audio spec".
 *
 * Every float operation below is a single IEEE f32 operation in a fixed order (build with
 * -ffp-contract=off), and the HIP kernels perform the same operations in the same order, so the
 * integer outputs can be compared bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ucfp_fft_tw.h"

static const float TW[1024][2] = UCFP_FFT_TW_INIT;

/* A1: linear resampler. out_len = floor(n * sr_out / sr_in). */
size_t ucfp_oracle_resample_len(size_t n, uint32_t sr_in, uint32_t sr_out) {
    return (size_t)(((unsigned __int128)n * sr_out) / sr_in);
}
void ucfp_oracle_resample_linear(const float* in, size_t n, uint32_t sr_in, uint32_t sr_out, float* out) {
    size_t m = ucfp_oracle_resample_len(n, sr_in, sr_out);
    for (size_t i = 0; i < m; i++) {
        unsigned __int128 num = (unsigned __int128)i * sr_in;
        size_t idx = (size_t)(num / sr_out);
        uint32_t rem = (uint32_t)(num % sr_out);
        float frac = (float)((double)rem / (double)sr_out);
        float x0 = in[idx], x1 = in[idx + 1 < n ? idx + 1 : n - 1];
        float d = x1 - x0;
        float mm = d * frac;
        out[i] = x0 + mm;
    }
}

static uint32_t bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

/* A2/A3 (round 2): one frame -> power spectrum P[0..N/2), N in {1024, 2048}, by the real-input route:
 *   window  w[m] = 0.5 - 0.5*cos(2*pi*m/N) (periodic Hann), applied HALVED: wh[m] = 0.5f * w[m]  (an exact scaling;
 *           it absorbs the factor 1/2 of the untangling step, so P is the plain |X[k]|^2 of the windowed frame)
 *   pack    z[n] = (x[2n]*wh[2n], x[2n+1]*wh[2n+1]),  n = 0..M-1,  M = N/2
 *   FFT     in-place radix-2 DIT over z, M points; in stages 1..3 a twiddle 1 is a pure add/subtract and a twiddle
 *           -i is (xi, -xr); every other butterfly (and EVERY butterfly from stage 4 on, whatever its twiddle) is the
 *           general one  t1 = xr c, t2 = xi s, t3 = xr s, t4 = xi c, v = (t1 - t2, t3 + t4), u' = u + v, x' = u - v
 *   untangle for k = 0..M/2-1, with Y = Z[(M-k) mod M] and W = TW_N^k = (c, s)  (bin M/2 pairs with itself:
 *           P[M/2] = (2 Zr)^2 + (2 Zi)^2):
 *           A = (Zr + Yr, Zi - Yi)   B = (Zr - Yr, Zi + Yi)   T = (Br c - Bi s, Br s + Bi c)
 *           X[k]   = (Ar + Ti, Ai - Tr)            P[k]   = Xr*Xr + Xi*Xi
 *           X[M-k] = (Ar - Ti, Ai + Tr) (conj.)    P[M-k] = ...           for k > 0
 * Half the butterflies of a complex N-point transform; the HIP kernels (audio.hip, wave_rfft_power) perform the
 * same single f32 operations in the same order.  The order is OURS (audiofp's is unknown: parity unpinned). */
static float half_hann(int m, int N) {
    const int tws = 2048 / N; /* table stride */
    float c = TW[(m * tws) & 1023][0];
    if (m * tws >= 1024) c = -c; /* cos(theta + pi) = -cos(theta) */
    const float w = 0.5f - 0.5f * c;
    return 0.5f * w;
}
static void frame_power(const float* x, int N, float* P, float* re, float* im) {
    const int M = N / 2;
    const int bits = N == 1024 ? 9 : 10;
    const int tws = 2048 / N;
    for (int n = 0; n < M; n++) {
        const uint32_t r = bitrev((uint32_t)n, bits);
        re[r] = x[2 * n] * half_hann(2 * n, N);
        im[r] = x[2 * n + 1] * half_hann(2 * n + 1, N);
    }
    for (int s = 1; s <= bits; s++) {
        const int m = 1 << s, half = m >> 1, tstep = 2048 / m;
        for (int b = 0; b < M; b += m)
            for (int j = 0; j < half; j++) {
                const int ti = j * tstep;
                const int i0 = b + j, i1 = b + j + half;
                const float xr = re[i1], xi = im[i1];
                float vr, vi;
                if (s <= 3 && ti == 0) {
                    vr = xr;
                    vi = xi;
                } else if (s <= 3 && ti == 512) {
                    vr = xi;
                    vi = -xr;
                } else {
                    const float c = TW[ti][0], sn = TW[ti][1];
                    const float t1 = xr * c, t2 = xi * sn, t3 = xr * sn, t4 = xi * c;
                    vr = t1 - t2;
                    vi = t3 + t4;
                }
                const float ur = re[i0], ui = im[i0];
                re[i0] = ur + vr;
                im[i0] = ui + vi;
                re[i1] = ur - vr;
                im[i1] = ui - vi;
            }
    }
    {   /* the self-paired bin M/2: A = (2 Zr, 0), B = (0, 2 Zi), W = -i  =>  X = (2 Zr, -2 Zi) */
        const float xr = re[M / 2] + re[M / 2], xi = im[M / 2] + im[M / 2];
        const float p1 = xr * xr, p2 = xi * xi;
        P[M / 2] = p1 + p2;
    }
    for (int k = 0; k < M / 2; k++) {
        const int kk = (M - k) & (M - 1);
        const float zr = re[k], zi = im[k], yr = re[kk], yi = im[kk];
        const float ar = zr + yr, ai = zi - yi, br = zr - yr, bi = zi + yi;
        const float c = TW[k * tws][0], sn = TW[k * tws][1];
        const float t1 = br * c, t2 = bi * sn, t3 = br * sn, t4 = bi * c;
        const float tr = t1 - t2, ti = t3 + t4;
        const float xr = ar + ti, xi = ai - tr;
        const float p1 = xr * xr, p2 = xi * xi;
        P[k] = p1 + p2;
        if (k > 0) {
            const float ur = ar - ti, ui = ai + tr;
            const float q1 = ur * ur, q2 = ui * ui;
            P[M - k] = q1 + q2;
        }
    }
}

size_t ucfp_oracle_stft_frames(size_t n, int N, int hop) { return n >= (size_t)N ? 1 + (n - N) / hop : 0; }

/* power spectrogram [frames][N/2] */
void ucfp_oracle_stft_power(const float* x, size_t n, int N, int hop, float* P) {
    size_t T = ucfp_oracle_stft_frames(n, N, hop);
#pragma omp parallel
    {
        float* re = (float*)malloc(sizeof(float) * N);
        float* im = (float*)malloc(sizeof(float) * N);
#pragma omp for
        for (size_t t = 0; t < T; t++) frame_power(x + t * hop, N, P + t * (N / 2), re, im);
        free(re);
        free(im);
    }
}

typedef struct {
    uint32_t fan_out, target_zone_t, target_zone_f, peaks_per_sec;
    float min_anchor_mag_db;
} wang_cfg;

typedef struct {
    uint32_t t, k;
    float p;
} peak_t;

static int cmp_peak_strength(const void* a, const void* b) {
    const peak_t* x = (const peak_t*)a;
    const peak_t* y = (const peak_t*)b;
    if (x->p != y->p) return x->p > y->p ? -1 : 1;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return (x->k > y->k) - (x->k < y->k);
}
static int cmp_peak_time(const void* a, const void* b) {
    const peak_t* x = (const peak_t*)a;
    const peak_t* y = (const peak_t*)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return (x->k > y->k) - (x->k < y->k);
}

#define WANG_N 1024
#define WANG_HOP 128
#define WANG_BINS 512
#define WANG_RT 7
#define WANG_RK 15
#define WANG_SR 8000

float ucfp_oracle_wang_floor_power(float db) { return (float)(65536.0 * pow(10.0, (double)db / 10.0)); }

/* A5: peaks (time-sorted). Returns count; out may be NULL to count only. */
size_t ucfp_oracle_wang_peaks(const float* P, size_t T, uint32_t peaks_per_sec, uint32_t* out_t, uint32_t* out_k,
                              float* out_p) {
    /* candidate flags in parallel over frames (the test of a cell is independent of the others),
     * then a sequential gather in (t, k) order */
    uint8_t* flag = (uint8_t*)calloc(T * WANG_BINS + 1, 1);
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t t = 0; t < T; t++)
        for (int k = 0; k < WANG_BINS; k++) {
            const float v = P[t * WANG_BINS + k];
            if (!(v > 0.0f)) continue;
            int ok = 1;
            const long t0 = (long)t - WANG_RT < 0 ? 0 : (long)t - WANG_RT;
            const long t1 = t + WANG_RT >= T ? (long)T - 1 : (long)t + WANG_RT;
            const int k0 = k - WANG_RK < 0 ? 0 : k - WANG_RK, k1 = k + WANG_RK > WANG_BINS - 1 ? WANG_BINS - 1 : k + WANG_RK;
            for (long tt = t0; tt <= t1 && ok; tt++)
                for (int kk = k0; kk <= k1; kk++) {
                    if (tt == (long)t && kk == k) continue;
                    const float o = P[tt * WANG_BINS + kk];
                    const int before = tt < (long)t || (tt == (long)t && kk < k);
                    if (o > v || (before && o == v)) {
                        ok = 0;
                        break;
                    }
                }
            flag[t * WANG_BINS + k] = (uint8_t)ok;
        }
    peak_t* cand = (peak_t*)malloc(sizeof(peak_t) * (T * 40 + 64));
    size_t nc = 0;
    for (size_t t = 0; t < T; t++)
        for (int k = 0; k < WANG_BINS; k++)
            if (flag[t * WANG_BINS + k]) {
                cand[nc].t = (uint32_t)t;
                cand[nc].k = (uint32_t)k;
                cand[nc].p = P[t * WANG_BINS + k];
                nc++;
            }
    free(flag);
    /* per-second cap: second = floor(t * hop / sr) */
    size_t np = 0, i = 0;
    peak_t* sel = (peak_t*)malloc(sizeof(peak_t) * (nc + 1));
    while (i < nc) {
        const uint32_t sec = (uint32_t)(((uint64_t)cand[i].t * WANG_HOP) / WANG_SR);
        size_t j = i;
        while (j < nc && (uint32_t)(((uint64_t)cand[j].t * WANG_HOP) / WANG_SR) == sec) j++;
        qsort(cand + i, j - i, sizeof(peak_t), cmp_peak_strength);
        size_t keep = j - i < peaks_per_sec ? j - i : peaks_per_sec;
        qsort(cand + i, keep, sizeof(peak_t), cmp_peak_time);
        for (size_t q = 0; q < keep; q++) sel[np++] = cand[i + q];
        i = j;
    }
    if (out_t)
        for (size_t q = 0; q < np; q++) {
            out_t[q] = sel[q].t;
            out_k[q] = sel[q].k;
            out_p[q] = sel[q].p;
        }
    free(cand);
    free(sel);
    return np;
}

/* A6: pairing, src/modality/audio.rs:965-1003. out: pairs of u32 (hash, t_anchor). */
size_t ucfp_oracle_wang_pairs(const uint32_t* pt, const uint32_t* pk, const float* pp, size_t np, const wang_cfg* cfg,
                              uint32_t* out, size_t cap) {
    const float floor_p = ucfp_oracle_wang_floor_power(cfg->min_anchor_mag_db);
    size_t n = 0;
    for (size_t i = 0; i < np; i++) {
        if (!(pp[i] >= floor_p)) continue;
        uint32_t taken = 0;
        for (size_t j = i + 1; j < np && taken < cfg->fan_out; j++) {
            const int32_t dt = (int32_t)pt[j] - (int32_t)pt[i];
            if (dt <= 0) continue;
            if (dt > (int32_t)cfg->target_zone_t) break;
            int32_t df = (int32_t)pk[j] - (int32_t)pk[i];
            if (df < 0) df = -df;
            if (df > (int32_t)cfg->target_zone_f) continue;
            if (n < cap) {
                out[2 * n] = (pk[i] << 23) | (pk[j] << 14) | ((uint32_t)dt & 0x3fffu);
                out[2 * n + 1] = pt[i];
            }
            n++;
            taken++;
        }
    }
    return n;
}

/* Whole path: 8 kHz samples -> hashes. Returns the number of hashes (may exceed cap: truncated). */
size_t ucfp_oracle_wang(const float* x, size_t n, const wang_cfg* cfg, uint32_t* out, size_t cap) {
    size_t T = ucfp_oracle_stft_frames(n, WANG_N, WANG_HOP);
    if (T == 0) return 0;
    float* P = (float*)malloc(sizeof(float) * T * WANG_BINS);
    ucfp_oracle_stft_power(x, n, WANG_N, WANG_HOP, P);
    size_t np = ucfp_oracle_wang_peaks(P, T, cfg->peaks_per_sec, NULL, NULL, NULL);
    uint32_t* pt = (uint32_t*)malloc(4 * (np + 1));
    uint32_t* pk = (uint32_t*)malloc(4 * (np + 1));
    float* pp = (float*)malloc(4 * (np + 1));
    ucfp_oracle_wang_peaks(P, T, cfg->peaks_per_sec, pt, pk, pp);
    size_t nh = ucfp_oracle_wang_pairs(pt, pk, pp, np, cfg, out, cap);
    free(P);
    free(pt);
    free(pk);
    free(pp);
    return nh;
}

/* ---- Haitsma-Kalker ------------------------------------------------------------------ */
#define HK_N 2048
#define HK_HOP 64
#define HK_SR 5000
#define HK_BANDS 33

/* band edges in bins: band b covers [edge[b], edge[b+1]); log-spaced between fmin and fmax */
void ucfp_oracle_haitsma_edges(float fmin, float fmax, uint32_t edges[HK_BANDS + 1]) {
    for (int b = 0; b <= HK_BANDS; b++) {
        double f = (double)fmin * pow((double)fmax / (double)fmin, (double)b / HK_BANDS);
        double bin = f * HK_N / HK_SR;
        uint32_t e = (uint32_t)ceil(bin);
        if (e > HK_N / 2) e = HK_N / 2;
        edges[b] = e;
    }
}

/* x at 5 kHz. out: one u32 per frame. Returns the frame count. */
size_t ucfp_oracle_haitsma_5k(const float* x, size_t n, float fmin, float fmax, uint32_t* out) {
    size_t T = ucfp_oracle_stft_frames(n, HK_N, HK_HOP);
    if (T == 0) return 0;
    uint32_t edges[HK_BANDS + 1];
    ucfp_oracle_haitsma_edges(fmin, fmax, edges);
    float* E = (float*)malloc(sizeof(float) * T * HK_BANDS);
#pragma omp parallel
    {
        float* re = (float*)malloc(sizeof(float) * HK_N);
        float* im = (float*)malloc(sizeof(float) * HK_N);
        float* P = (float*)malloc(sizeof(float) * HK_N / 2);
#pragma omp for
        for (size_t t = 0; t < T; t++) {
            frame_power(x + t * HK_HOP, HK_N, P, re, im);
            for (int b = 0; b < HK_BANDS; b++) {
                float e = 0.0f;
                for (uint32_t k = edges[b]; k < edges[b + 1]; k++) e = e + P[k];
                E[t * HK_BANDS + b] = e;
            }
        }
        free(re);
        free(im);
        free(P);
    }
    for (size_t t = 0; t < T; t++) {
        uint32_t h = 0;
        for (int b = 0; b < 32; b++) {
            const float cur = E[t * HK_BANDS + b] - E[t * HK_BANDS + b + 1];
            const float prev = t ? E[(t - 1) * HK_BANDS + b] - E[(t - 1) * HK_BANDS + b + 1] : 0.0f;
            const float dd = cur - prev;
            if (dd > 0.0f) h |= 1u << b;
        }
        out[t] = h;
    }
    free(E);
    return T;
}
