// audio.hip -- Wang landmark hashes and Haitsma-Kalker sub-fingerprints for gfx950.
//
// Replaces the arithmetic behind audio::fingerprint_wang_with (src/modality/audio.rs:64-98) and
// audio::fingerprint_haitsma_with (audio.rs:181-224), i.e. audiofp's Wang::extract /
// Haitsma::extract / dsp::resample::linear.  Spec: DESIGN.md "Audio spec" A1..A8; CPU statement:
// oracle/ (audio).  Every float step is one IEEE f32 operation in the order the oracle uses (the
// library is built with -ffp-contract=off), so the integer outputs are bit-identical.
//
//   resample_linear   A1   one thread per output sample
//   wave_fft_power_core<N> A2-3 ONE WAVE PER FRAME: Hann window + 1024/2048-point radix-2 FFT, 16/32 points per
//                          lane in registers as (re, im) pairs on packed-f32 instructions; the first transpose goes
//                          through LDS, the second runs on v_permlane32/16_swap; stage-major twiddle table
//   wang_stream       A3+5 Wang: frames streamed through an LDS ring of row maxima, peaks judged in the
//                          kernel one round behind (separable neighbourhood maximum + exact tie rule); nothing
//                          is spilled
//   stft_power<2048>  A7   Haitsma: 33 band energies per frame (chunked so the band buffer stays bounded)
//   wang_select       A5   one wave per second of audio: rank by strength, keep peaks_per_sec,
//                          order by (t, k)
//   wang_pair_*       A6   one thread per anchor walks the time-sorted peaks (audio.rs:965-1003)
//   haitsma_bits      A8   sign of the time/frequency double difference

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ucfp_fft_tw.h"
#include "common.h"

namespace ucfp {

namespace {

__constant__ float c_tw[1024][2] = UCFP_FFT_TW_INIT;

constexpr int kWangN = 1024, kWangHop = 128, kWangBins = 512, kRT = 7, kRK = 15, kWangSr = 8000;
constexpr int kHkN = 2048, kHkHop = 64, kHkBands = 33;
constexpr int kCandCap = 320;   // > (63/8 + 1) * (512/16) possible peaks per second

// ---- A1 ----------------------------------------------------------------------------------
__global__ void resample_linear_kernel(const float* __restrict__ in, size_t n, uint32_t sr_in, uint32_t sr_out,
                                       float* __restrict__ out, size_t m) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint64_t num = (uint64_t)i * sr_in;  // < 2^64 for any buffer that fits in memory
    const size_t idx = (size_t)(num / sr_out);
    const uint32_t rem = (uint32_t)(num % sr_out);
    const float frac = (float)((double)rem / (double)sr_out);
    const float x0 = in[idx], x1 = in[idx + 1 < n ? idx + 1 : n - 1];
    const float d = x1 - x0;
    const float mm = d * frac;
    out[i] = x0 + mm;
}

// ---- A2/A3: one wave per frame, FFT register-tiled ------------------------------------------------
// N = 64 * E points, E = 16 (Wang, N = 1024) or 32 (Haitsma, N = 2048) complex values per lane.
// The radix-2 DIT butterflies are EXACTLY those of the oracle (same operands, same f32 ops), only
// their placement changes: a lane keeps E points in registers and runs every stage whose partner
// distance stays inside its E points, then the wave transposes through LDS:
//   phase 1  lane L holds p = E*L + i           -> stages 1..B        (B = log2 E), twiddles constant
//   phase 2  lane (hi, lo) holds p = hi*E*E + m*E + lo  -> stages B+1..2B
//   phase 3  lane l holds p = e*64 + l          -> stages 2B+1..log2 N
// Two LDS transposes (padded: p + p/E) replace the ten LDS round trips of a stage-by-stage FFT.
// Twiddles of the stages that run after a transpose live in LDS STAGE-MAJOR: stage st (2^(st-1) distinct
// twiddles W^(jj * 2048 >> st)) occupies entries [2^(st-1) - E, 2^st - E), so a read at jj = const | lane-low-bits
// touches consecutive 8-byte entries.  (Read from the natural 1024-entry table at stride 2048 >> st, the 16 distinct
// addresses of a stage-5 read all fall on one bank: 59 % of the LDS cycles of the first version were conflicts.)
constexpr int kFftWaves = 12;  // one workgroup per CU, 3 waves per SIMD: 16 KiB of twiddles + 12 x 8.25 KiB (N = 2048)
template <int N>
struct FftLds {
    float2 stw[N - N / 64];
    float buf[kFftWaves][N + 64];    // per wave (split-component transpose); reused as the power spectrum float[N/2]
    float win[N / 64][64];           // Hann window at a lane's sample positions
};
template <int N>
__device__ __forceinline__ void fill_stage_twiddles(float2* __restrict__ stw, int tid, int nthreads) {
    constexpr int E = N / 64;
    for (int idx = tid; idx < N - E; idx += nthreads) {
        const int g = idx + E;                       // 2^(st-1) + jj
        const int st = 32 - __clz(g);
        const int jj = g - (1 << (st - 1));
        stw[idx] = make_float2(c_tw[jj * (2048 >> st)][0], c_tw[jj * (2048 >> st)][1]);
    }
}

// One radix-2 butterfly on (re, im) pairs in packed f32 (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations
// per lane per instruction, so 6 instructions instead of 10).  Same operations in the same order as the
// oracle: t1 = xr c, t2 = xi s, t3 = xr s, t4 = xi c, v = (t1 - t2, t3 + t4), u' = u + v, x' = u - v.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bfly(f32x2& u, f32x2& x, f32x2 tw) {   // tw = (c, s)
    const f32x2 a = x * f32x2{tw.x, tw.x};              // (t1, t4): broadcast by op_sel, no move
    // (-t2, t3) = (xi * -s, xr * s) in ONE instruction: swapped x halves by op_sel, the low product's sign by the
    // source modifier (xi * (-s) == -(xi * s) exactly); the compiler spends a negate + a move on this otherwise
    f32x2 b;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(b) : "v"(x), "v"(tw));
    const f32x2 v = a + b;                              // (t1 - t2, t4 + t3)
    const f32x2 w = u;
    u = w + v;
    x = w - v;
}

// element i of lane L is p = E*L + i = bit-reversed sample index n = (rev(i) << 6) | rev(L); the sample
// positions of a lane are the same for every frame, so a caller may keep the window in registers
// gfx950 half / row exchange between two registers (see transpose 2 in wave_fft_power_core)
template <int W>
__device__ __forceinline__ void swap_lanes(f32x2& a, f32x2& b) {   // both components
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const uint32_t ua = __float_as_uint(a[c]), ub = __float_as_uint(b[c]);
        if constexpr (W == 32) {
            const auto r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
            a[c] = __uint_as_float(r[0]);
            b[c] = __uint_as_float(r[1]);
        } else {
            const auto r = __builtin_amdgcn_permlane16_swap(ua, ub, false, false);
            a[c] = __uint_as_float(r[0]);
            b[c] = __uint_as_float(r[1]);
        }
    }
}

// first-stage forms for real input (see wave_fft_power_core)
__device__ __forceinline__ void bfly_real(f32x2& u, f32x2& x) {          // twiddle (1, -0), imaginary parts zero
    const float a = u.x, b = x.x;
    u.x = a + b;
    x.x = a - b;
}
__device__ __forceinline__ void bfly_real_in(f32x2& u, f32x2& x, f32x2 tw) {   // imaginary INPUTS zero
    f32x2 m;                                                                  // (xr c, xr s)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(m) : "v"(x), "v"(tw));
    const f32x2 w = u;
    u = w + m;
    x = w - m;
}

template <int N>
__device__ __forceinline__ void frame_load(const float* __restrict__ src, int lane, float (&s)[N / 64]) {
    constexpr int E = N / 64;
    constexpr int B = E == 16 ? 4 : 5;
    const uint32_t rl = __brev((uint32_t)lane) >> 26;  // 6-bit reversal of the lane
#pragma unroll
    for (int i = 0; i < E; i++) {
        const uint32_t ri = __brev((uint32_t)i) >> (32 - B);   // compile-time after unrolling
        s[i] = src[(ri << 6) | rl];
    }
}
template <int N>
__device__ __forceinline__ void frame_window(int lane, float (&w)[N / 64]) {
    constexpr int E = N / 64;
    constexpr int B = E == 16 ? 4 : 5;
    constexpr int TWS = 2048 / N;
    const uint32_t rl = __brev((uint32_t)lane) >> 26;
#pragma unroll
    for (int i = 0; i < E; i++) {
        const uint32_t ri = __brev((uint32_t)i) >> (32 - B);
        const uint32_t n = (ri << 6) | rl;
        float c = c_tw[(n * TWS) & 1023][0];
        if (n * TWS >= 1024) c = -c;
        w[i] = 0.5f - 0.5f * c;
    }
}

// SPLIT: the transposes move the real and the imaginary halves one after the other through a buffer of N + 64
// FLOATS (4.25 KiB at N = 1024) instead of N + 64 float2 -- twice the LDS instructions for half the LDS, which is
// what lets 16 frames be in flight per CU in wang_stream_kernel.
template <int N, bool SPLIT = false>
__device__ __forceinline__ void wave_fft_power_core(const float (&smp)[N / 64], const float (&win)[N / 64], int lane,
                                                    const float2* __restrict__ stw, float2* __restrict__ buf) {
    constexpr int E = N / 64;
    constexpr int B = E == 16 ? 4 : 5;
    constexpr int BITS = N == 1024 ? 10 : 11;
    f32x2 x[E];
#pragma unroll
    for (int i = 0; i < E; i++) x[i] = f32x2{smp[i] * win[i], 0.0f};
    // ---- phase 1: stages 1..B on bits 0..B-1 (register index), twiddles are table constants ----
    // The input is real, so the first two stages are cheaper than general butterflies without changing a bit of
    // the power spectrum: with twiddle (1, -0) and zero imaginary parts a butterfly is one add and one subtract of
    // the real parts (x*1 and 0*s are exact, the imaginary outputs stay zero); with zero imaginary INPUTS and any
    // twiddle, v = (xr c - 0 s, xr s + 0 c) = (xr c, xr s) exactly.  (Only the sign of exact zeros can differ from
    // the five-instruction form, and (+-0)^2 is +0.)
#pragma unroll
    for (int i0 = 0; i0 < E; i0 += 2) bfly_real(x[i0], x[i0 + 1]);
#pragma unroll
    for (int i0 = 0; i0 < E; i0 += 4) {
        bfly_real(x[i0], x[i0 + 2]);
        bfly_real_in(x[i0 + 1], x[i0 + 3], f32x2{c_tw[512][0], c_tw[512][1]});
    }
#pragma unroll
    for (int st = 3; st <= B; st++) {
        const int half = 1 << (st - 1), tstep = 2048 >> st;
#pragma unroll
        for (int i0 = 0; i0 < E; i0++) {
            if (i0 & half) continue;
            const int jj = i0 & (half - 1);
            bfly(x[i0], x[i0 + half], f32x2{c_tw[jj * tstep][0], c_tw[jj * tstep][1]});
        }
    }
    // ---- transpose 1 ----
    const int lo = lane & (E - 1), hi = lane >> B;
    float* fb = reinterpret_cast<float*>(buf);
    if constexpr (SPLIT) {
        float nr[E];
#pragma unroll
        for (int i = 0; i < E; i++) fb[(E + 1) * lane + i] = x[i].x;
        wave_lds_fence();
#pragma unroll
        for (int m = 0; m < E; m++) {
            const int pp = hi * E * E + m * E + lo;
            nr[m] = fb[pp + (pp >> B)];
        }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < E; i++) fb[(E + 1) * lane + i] = x[i].y;
        wave_lds_fence();
#pragma unroll
        for (int m = 0; m < E; m++) {
            const int pp = hi * E * E + m * E + lo;
            x[m] = f32x2{nr[m], fb[pp + (pp >> B)]};
        }
    } else {
#pragma unroll
        for (int i = 0; i < E; i++) buf[(E + 1) * lane + i] = make_float2(x[i].x, x[i].y);   // p + p/E, p = E*lane + i
        wave_lds_fence();
#pragma unroll
        for (int m = 0; m < E; m++) {
            const int pp = hi * E * E + m * E + lo;
            const float2 v = buf[pp + (pp >> B)];
            x[m] = f32x2{v.x, v.y};
        }
    }
    // ---- phase 2: stages B+1..2B on bits B..2B-1 (register index m) ----
#pragma unroll
    for (int st = B + 1; st <= 2 * B; st++) {
        const int halfm = 1 << (st - B - 1);
#pragma unroll
        for (int m0 = 0; m0 < E; m0++) {
            if (m0 & halfm) continue;
            const int jj = ((m0 & (halfm - 1)) << B) | lo;
            const float2 t2 = stw[(1 << (st - 1)) - E + jj];
            bfly(x[m0], x[m0 + halfm], f32x2{t2.x, t2.y});
        }
    }
    // ---- transpose 2: p = hi*E*E + m*E + lo  ->  p = e*64 + lane.  For E = 16 this only exchanges the two lane-row
    // bits (hi) with the two low bits of the register index: two rounds of gfx950's row/half swaps, no LDS at all
    // (E = 32: one lane bit and one register bit, one round).
    // v_permlane32_swap(A, B): lanes 32-63 of A <-> lanes 0-31 of B, i.e. (register bit, lane bit 5) transposed;
    // v_permlane16_swap: odd rows of A <-> even rows of B, i.e. (register bit, lane bit 4).  Afterwards register
    // r = (m_hi, hi) holds e = (hi, m_hi) of this lane: a renaming. ----
    if constexpr (E == 16) {
#pragma unroll
        for (int r = 0; r < E; r++) {
            if (r & 2) continue;
            swap_lanes<32>(x[r], x[r + 2]);
        }
#pragma unroll
        for (int r = 0; r < E; r++) {
            if (r & 1) continue;
            swap_lanes<16>(x[r], x[r + 1]);
        }
        f32x2 y[E];
#pragma unroll
        for (int r = 0; r < E; r++) y[((r & 3) << 2) | (r >> 2)] = x[r];
#pragma unroll
        for (int e = 0; e < E; e++) x[e] = y[e];
    } else {
        // E = 32: hi is lane bit 5 alone and trades places with register bit 0; e = (hi, m >> 1)
#pragma unroll
        for (int r = 0; r < E; r += 2) swap_lanes<32>(x[r], x[r + 1]);
        f32x2 y[E];
#pragma unroll
        for (int r = 0; r < E; r++) y[((r & 1) << 4) | (r >> 1)] = x[r];
#pragma unroll
        for (int e = 0; e < E; e++) x[e] = y[e];
    }
    // ---- phase 3: stages 2B+1..BITS on bits 2B.. (register index e, bit st-1-6) ----
#pragma unroll
    for (int st = 2 * B + 1; st <= BITS; st++) {
        const int halfe = 1 << (st - 1 - 6);
#pragma unroll
        for (int e0 = 0; e0 < E; e0++) {
            if (e0 & halfe) continue;
            const int jj = ((e0 & (halfe - 1)) << 6) | lane;
            const float2 t2 = stw[(1 << (st - 1)) - E + jj];
            bfly(x[e0], x[e0 + halfe], f32x2{t2.x, t2.y});
        }
    }
    wave_lds_fence();
    // ---- power spectrum, bins k = e*64 + lane < N/2, into the (now free) buffer as float[N/2] ----
    float* pw = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int e = 0; e < E / 2; e++) {
        const f32x2 sq = x[e] * x[e];
        pw[e * 64 + lane] = sq.x + sq.y;
    }
    wave_lds_fence();
}

template <int N, bool HAITSMA>
__global__ __launch_bounds__(kFftWaves * 64) void stft_power_kernel(const float* __restrict__ x, size_t first_frame,
                                                         size_t n_frames, int hop, float* __restrict__ out,
                                                         const uint32_t* __restrict__ edges,
                                                         float* __restrict__ rowmax_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    FftLds<N>& L = *reinterpret_cast<FftLds<N>*>(lds_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    fill_stage_twiddles<N>(L.stw, threadIdx.x, kFftWaves * 64);
    if (wave == 0) {   // a lane windows the same sample positions in every frame
        float w0[N / 64];
        frame_window<N>(lane, w0);
#pragma unroll
        for (int i = 0; i < N / 64; i++) L.win[i][lane] = w0[i];
    }
    __syncthreads();
    float2* buf = reinterpret_cast<float2*>(L.buf[wave]);
    const float* pw = L.buf[wave];
    // frames are dealt to waves round-robin over the whole grid (three waves per SIMD cover the sample loads)
    const uint32_t k0 = lane < kHkBands ? edges[lane] : 0u, k1 = lane < kHkBands ? edges[lane + 1] : 0u;
    const size_t step = (size_t)gridDim.x * kFftWaves;
    for (size_t f = (size_t)blockIdx.x * kFftWaves + wave; f < n_frames; f += step) {
        float smp[N / 64], win[N / 64];
        frame_load<N>(x + (first_frame + f) * (size_t)hop, lane, smp);
#pragma unroll
        for (int i = 0; i < N / 64; i++) win[i] = L.win[i][lane];
        wave_fft_power_core<N, true>(smp, win, lane, L.stw, buf);
        {
            // lane b sums band b sequentially (same order as the oracle); eight spectrum reads are in flight at a time
            float e = 0.0f;
            uint32_t k = k0;
            for (; k + 8 <= k1; k += 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = pw[k + j];
#pragma unroll
                for (int j = 0; j < 8; j++) e = e + v[j];
            }
            for (; k < k1; k++) e = e + pw[k];
            if (lane < kHkBands) out[f * (size_t)kHkBands + lane] = e;
        }
        wave_lds_fence();
    }
}

// ---- A3 + A5 fused: STFT frames streamed through LDS, peaks picked without spilling the spectrogram ----
// A workgroup (kSW waves) owns a segment of `seg` consecutive frames and walks it in rounds of kSW frames, one
// FFT per wave (plus kRT halo frames on each side, recomputed: 14 / seg extra).  Of every frame only two
// things survive in LDS:
//   ring    its +-kRK-bin running maximum ("row maximum"), 2 KiB, in a ring of kRing frames
//   plist   its row-local peak candidates: bins with P == row maximum > 0 and no equal value among the kRK
//           bins below (the same-row half of the tie rule) -- two of them are >= 16 bins apart, so <= 32 per frame
// Once the rows t-kRT .. t+kRT are in the ring, a candidate (t, k, v) is a peak iff v equals the largest
// of those 15 row maxima at bin k and none of the EARLIER rows' maxima equals v (an equal cell earlier in
// (t, k) order wins the tie; rows outside [0, total) duplicate rows inside the window, so they are simply
// skipped).  HBM sees the samples once and the peaks -- not 2 x 4 B x 512 bins per frame of spilled spectrum.
constexpr int kSegMax = 512;    // frames per workgroup segment: long inputs (halo 14 / 512 = 2.7 %); short ones get
                                // shorter segments so that ~1000 workgroups exist (wang_segment)
constexpr int kSW = 12;      // waves per workgroup = frames in flight (LDS: 8.5 KiB FFT buffer each + the ring)
constexpr int kRing = 38;    // >= 2 kRT + 2 kSW: the rows being judged (one round behind) + the rows being produced
constexpr int kPl = 32;      // row-local candidates per frame: two of them are always >= 16 bins apart

struct WangStreamLds {
    float2 stw[kWangN - kWangN / 64];
    float buf[kSW][kWangN + 64];
    float win[kWangN / 64][64];   // Hann window at this lane's 16 sample positions (registers are for the FFT)
    float ring[kRing][kWangBins];
    uint32_t pl_cnt[kRing];
    uint32_t pl_k[kRing][kPl];
    float pl_v[kRing][kPl];
};

inline uint32_t wang_segment(size_t frames) {
    size_t seg = (frames + 1023) / 1024;
    if (seg < 48) seg = 48;
    if (seg > (size_t)kSegMax) seg = kSegMax;
    return (uint32_t)seg;
}

__global__ __launch_bounds__(kSW * 64) void wang_stream_kernel(const float* __restrict__ x, size_t total_frames,
                                                          uint32_t seg, uint32_t* __restrict__ cand_cnt,
                                                          uint32_t* __restrict__ cand_t,
                                                          uint32_t* __restrict__ cand_k, float* __restrict__ cand_p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    WangStreamLds& L = *reinterpret_cast<WangStreamLds*>(lds_raw);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // frame and ring arithmetic stays scalar
    fill_stage_twiddles<kWangN>(L.stw, threadIdx.x, kSW * 64);
    if (wave == 0) {
        float w0[kWangN / 64];
        frame_window<kWangN>(lane, w0);
#pragma unroll
        for (int i = 0; i < kWangN / 64; i++) L.win[i][lane] = w0[i];
    }
    __syncthreads();
    float2* buf = reinterpret_cast<float2*>(L.buf[wave]);
    const float* pw = L.buf[wave];
    const long total = (long)total_frames;
    const long s0 = (long)blockIdx.x * seg;                        // frames [s0, s1) are this segment's to judge
    const long s1 = s0 + seg < total ? s0 + seg : total;
    const long f_lo = s0 - kRT < 0 ? 0 : s0 - kRT;                 // frames [f_lo, f_hi) are computed
    const long f_hi = s1 + kRT < total ? s1 + kRT : total;
    // the Hann window of this lane's 16 sample positions, and the NEXT frame's samples: loaded one round ahead, so
    // the HBM/L2 latency hides behind the current FFT instead of stalling every wave at the top of a round
    float nxt[kWangN / 64];
#pragma unroll
    for (int i = 0; i < kWangN / 64; i++) nxt[i] = 0.0f;
    if (f_lo + wave < f_hi) frame_load<kWangN>(x + (size_t)(f_lo + wave) * kWangHop, lane, nxt);
#pragma unroll
    for (int i = 0; i < kWangN / 64; i++) asm volatile("" : "+v"(nxt[i]));   // taken in before the loop (see below)
    int slot = (int)((f_lo + wave) % kRing);                       // ring row of frame base + wave
    // a found peak waits one round for its slot: the atomic's round trip overlaps the next FFT
    bool pend = false, pend_wave = false;      // pend_wave: wave-uniform "an atomic is in flight"
    uint32_t pend_base = 0, pend_rank = 0, pend_sec = 0, pend_t = 0, pend_k = 0;
    int pend_leader = 0;
    float pend_v = 0.0f;
    auto flush = [&]() {
        if (pend_wave) {
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)pend_base, pend_leader);
            const uint32_t pos = base + pend_rank;
            if (pend && pos < (uint32_t)kCandCap) {
                cand_t[(size_t)pend_sec * kCandCap + pos] = pend_t;
                cand_k[(size_t)pend_sec * kCandCap + pos] = pend_k;
                cand_p[(size_t)pend_sec * kCandCap + pos] = pend_v;
            }
        }
        pend = false;
        pend_wave = false;
    };
    for (long base = f_lo; base < s1 + kRT + kSW; base += kSW) {
        const long f = base + wave;
        // ---- judge frame base + wave - kSW - kRT: its window [t - kRT, t + kRT] was complete at the last barrier, so
        // this overlaps the other waves' FFTs instead of standing between two barriers of its own ----
        flush();
        // The prefetched samples are taken in HERE (they landed during the last FFT): returns come back in order, so
        // a wait for them placed after the judge's atomic would wait for the atomic as well.
#pragma unroll
        for (int i = 0; i < kWangN / 64; i++) asm volatile("" : "+v"(nxt[i]));
        const long t = f - kSW - kRT;
        if (t >= s0 && t < s1) {
            int st = slot - kSW - kRT;
            if (st < 0) st += kRing;
            // The whole wave pays for every instruction here, so the test is kept short: v is the row maximum of its
            // own row at k, hence "v equals the window maximum and no earlier row reaches it" is
            //   max(7 earlier rows at k) < v  and  max(7 later rows at k) <= v.
            // All reads are unconditional (lanes past the list read a stale entry and are masked at the end).
            const uint32_t n = L.pl_cnt[st];
            const uint32_t pk = L.pl_k[st][lane & (kPl - 1)] & (uint32_t)(kWangBins - 1);
            const float pv = L.pl_v[st][lane & (kPl - 1)];
            int rs = st - kRT;
            if (rs < 0) rs += kRing;
            float rr[2 * kRT + 1];
            const float* cell = &L.ring[0][pk];
            if (rs + 2 * kRT < kRing) {                    // the window does not wrap: one address, 15 offsets
                const float* c0 = cell + rs * kWangBins;
#pragma unroll
                for (int d = 0; d <= 2 * kRT; d++) rr[d] = c0[d * kWangBins];
            } else {
#pragma unroll
                for (int d = 0; d <= 2 * kRT; d++) {
                    const int r = rs + d >= kRing ? rs + d - kRing : rs + d;
                    rr[d] = cell[r * kWangBins];
                }
            }
            if (t < kRT || t + kRT >= total) {             // rows outside [0, total) duplicate rows inside: drop them
#pragma unroll
                for (int d = 0; d <= 2 * kRT; d++) {
                    const long tt = t + d - kRT;
                    if (tt < 0 || tt >= total) rr[d] = -1.0f;
                }
            }
            float mb = rr[0], ma = rr[kRT + 1];
#pragma unroll
            for (int d = 1; d < kRT; d++) {
                mb = fmaxf(mb, rr[d]);
                ma = fmaxf(ma, rr[kRT + 1 + d]);
            }
            const bool is_peak = (uint32_t)lane < n && mb < pv && ma <= pv;
            // one counter bump per wave (all its peaks share the frame, hence the second); the returned base is
            // consumed a round later by flush(), so the L2 round trip overlaps the next FFT.  The address is hidden
            // from the compiler: for a uniform address it aggregates by itself and reads the result back at once
            // (s_waitcnt vmcnt(0) + v_readfirstlane right behind the atomic), which parks every wave for the trip.
            const uint64_t pm = __ballot(is_peak);
            if (pm) {
                const uint32_t sec = (uint32_t)(((size_t)t * kWangHop) / kWangSr);
                pend = is_peak;
                pend_sec = sec;
                pend_t = (uint32_t)t;
                pend_k = pk;
                pend_v = pv;
                pend_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                pend_leader = __builtin_ctzll(pm);
                pend_wave = true;
                if (lane == pend_leader) {
                    const uint32_t cnt = (uint32_t)__popcll(pm);
                    typedef __attribute__((address_space(1))) uint32_t* global_u32;
                    global_u32 addr = (global_u32)(cand_cnt + sec);
                    asm volatile("" : "+v"(addr));
                    pend_base = __hip_atomic_fetch_add(addr, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // ---- produce frame base + wave ----
        if (f < f_hi) {
            float smp[kWangN / 64], win[kWangN / 64];
#pragma unroll
            for (int i = 0; i < kWangN / 64; i++) {
                smp[i] = nxt[i];
                win[i] = L.win[i][lane];
            }
            if (f + kSW < f_hi) frame_load<kWangN>(x + (size_t)(f + kSW) * kWangHop, lane, nxt);
            wave_fft_power_core<kWangN, true>(smp, win, lane, L.stw, buf);
            // Row maximum over +-kRK bins and the same-row tie test, blocked: lane L owns bins 8L .. 8L+7.  The
            // window [k-15, k+15] of bin k = 8L + j is  suffix_{L-2}[j+1] u block_{L-1} u block_L u block_{L+1} u
            // prefix_{L+2}[j-1], and the 15 bins below k are  suffix_{L-2}[j+1] u block_{L-1} u prefix_L[j-1]:
            // 14 maxima per lane for the prefix / suffix tables, two LDS exchanges, instead of 31 taps per bin.
            // P >= 0, so -1 stands for "no bin there" (a clamped duplicate never changes a maximum either).
            float* row = L.ring[slot];
            float* sx = L.buf[wave];      // scratch OVER the spectrum (b8 is read first; the LDS keeps a wave's order):
            float* px = sx + 64 * 7;      // [lane][7] suffix 1..7, [lane][8] prefix (row strides 7 and 9: conflict-free)
            float b8[8], pre[8], suf[8];
            {
                const float4 lo4 = *reinterpret_cast<const float4*>(pw + 8 * lane);
                const float4 hi4 = *reinterpret_cast<const float4*>(pw + 8 * lane + 4);
                b8[0] = lo4.x; b8[1] = lo4.y; b8[2] = lo4.z; b8[3] = lo4.w;
                b8[4] = hi4.x; b8[5] = hi4.y; b8[6] = hi4.z; b8[7] = hi4.w;
            }
            pre[0] = b8[0];
#pragma unroll
            for (int j = 1; j < 8; j++) pre[j] = fmaxf(pre[j - 1], b8[j]);
            suf[7] = b8[7];
#pragma unroll
            for (int j = 6; j >= 0; j--) suf[j] = fmaxf(suf[j + 1], b8[j]);
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (j > 0) sx[7 * lane + j - 1] = suf[j];
                if (j < 7) px[9 * lane + j] = pre[j];
            }
            px[9 * lane + 7] = pre[7];
            wave_lds_fence();
            // every neighbour read is unconditional at a clamped lane (one burst of 16 LDS reads, one wait) and the
            // out-of-range ones are replaced afterwards: a read under a lane condition becomes a branch of its own
            const int lm1 = lane >= 1 ? lane - 1 : 0, lp1 = lane <= 62 ? lane + 1 : 63;
            const int lm2 = lane >= 2 ? lane - 2 : 0, lp2 = lane <= 61 ? lane + 2 : 63;
            float nb_m1 = px[9 * lm1 + 7], nb_p1 = px[9 * lp1 + 7];     // block maxima of the neighbours
            float s2v[7], p2v[7];
#pragma unroll
            for (int j = 0; j < 7; j++) {
                s2v[j] = sx[7 * lm2 + j];          // suffix_{L-2}[j+1]
                p2v[j] = px[9 * lp2 + j];          // prefix_{L+2}[j]
            }
            const float blk_m1 = lane >= 1 ? nb_m1 : -1.0f;
            const float blk_p1 = lane <= 62 ? nb_p1 : -1.0f;
            float rm[8];
            uint32_t cbits = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float s2 = (lane >= 2 && j < 7) ? s2v[j < 7 ? j : 0] : -1.0f;       // suffix_{L-2}[j+1]
                const float p2 = (lane <= 61 && j > 0) ? p2v[j > 0 ? j - 1 : 0] : -1.0f;  // prefix_{L+2}[j-1]
                const float below = fmaxf(fmaxf(s2, blk_m1), j > 0 ? pre[j - 1] : -1.0f);
                const float m = fmaxf(fmaxf(below, pre[7]), fmaxf(blk_p1, p2));
                rm[j] = m;
                if (b8[j] > 0.0f && b8[j] == m && below != b8[j]) cbits |= 1u << j;
            }
            *reinterpret_cast<float4*>(row + 8 * lane) = make_float4(rm[0], rm[1], rm[2], rm[3]);
            *reinterpret_cast<float4*>(row + 8 * lane + 4) = make_float4(rm[4], rm[5], rm[6], rm[7]);
            // two row-local candidates are >= 16 bins apart, a lane owns 8 bins: at most ONE bit of cbits is set,
            // and one ballot compacts the row (the list's order is irrelevant: wang_select ranks the peaks)
            const bool c = cbits != 0;
            const uint64_t mask = __ballot(c);
            if (c) {
                const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                const uint32_t kk = 8u * (uint32_t)lane + (uint32_t)(__ffs((int)cbits) - 1);
                if (pos < (uint32_t)kPl) {
                    L.pl_k[slot][pos] = kk;
                    L.pl_v[slot][pos] = row[kk];   // a candidate equals its row maximum, just stored
                }
            }
            if (lane == 0) {
                const uint32_t npl = (uint32_t)__popcll(mask);
                L.pl_cnt[slot] = npl < (uint32_t)kPl ? npl : (uint32_t)kPl;
            }
        }
        __syncthreads();   // the only one per round: rows base .. base + kSW - 1 are complete
        slot = slot + kSW >= kRing ? slot + kSW - kRing : slot + kSW;
    }
    flush();
}

// one wave per second: keep the `pps` strongest, ordered by (t, k)
__global__ __launch_bounds__(64) void wang_select_kernel(const uint32_t* __restrict__ cand_cnt,
                                                         const uint32_t* __restrict__ cand_t,
                                                         const uint32_t* __restrict__ cand_k,
                                                         const float* __restrict__ cand_p, uint32_t pps,
                                                         uint32_t* __restrict__ sel_cnt, uint32_t* __restrict__ sel_t,
                                                         uint32_t* __restrict__ sel_k, float* __restrict__ sel_p) {
    __shared__ uint32_t st[kCandCap], sk[kCandCap];
    __shared__ float sp[kCandCap];
    __shared__ uint8_t keep[kCandCap];
    const uint32_t sec = blockIdx.x;
    const int lane = threadIdx.x;
    uint32_t n = cand_cnt[sec];
    n = n < (uint32_t)kCandCap ? n : (uint32_t)kCandCap;
    for (uint32_t i = lane; i < n; i += 64) {
        st[i] = cand_t[(size_t)sec * kCandCap + i];
        sk[i] = cand_k[(size_t)sec * kCandCap + i];
        sp[i] = cand_p[(size_t)sec * kCandCap + i];
    }
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64) {
        uint32_t rank = 0;
        const float p = sp[i];
        const uint32_t t = st[i], k = sk[i];
        for (uint32_t j = 0; j < n; j++) {
            const float q = sp[j];
            const bool before = q > p || (q == p && (st[j] < t || (st[j] == t && sk[j] < k)));
            rank += before ? 1u : 0u;
        }
        keep[i] = rank < pps ? 1 : 0;
    }
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64) {
        if (!keep[i]) continue;
        uint32_t pos = 0;
        const uint32_t t = st[i], k = sk[i];
        for (uint32_t j = 0; j < n; j++)
            if (keep[j] && (st[j] < t || (st[j] == t && sk[j] < k))) pos++;
        sel_t[(size_t)sec * pps + pos] = t;
        sel_k[(size_t)sec * pps + pos] = k;
        sel_p[(size_t)sec * pps + pos] = sp[i];
    }
    if (lane == 0) sel_cnt[sec] = n < pps ? n : pps;
}

// single-block exclusive scan: out[i] = sum(in[0..i)), out[n] = total
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const uint32_t* __restrict__ in, size_t n,
                                                              uint32_t* __restrict__ out) {
    // one block; a thread scans kE consecutive elements serially, the block scans the thread totals
    constexpr int kE = 16;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (size_t base = 0; base < n; base += (size_t)1024 * kE) {
        const size_t i0 = base + (size_t)tid * kE;
        uint32_t v[kE];
        uint32_t tot = 0;
#pragma unroll
        for (int e = 0; e < kE; e++) {
            v[e] = i0 + e < n ? in[i0 + e] : 0u;
            tot += v[e];
        }
        uint32_t inc = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const uint32_t c = carry;
        uint32_t run = c + woff + inc - tot;
#pragma unroll
        for (int e = 0; e < kE; e++) {
            if (i0 + e < n) out[i0 + e] = run;
            run += v[e];
        }
        __syncthreads();
        if (tid == 1023) carry = c + woff + inc;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}

// Multi-block exclusive scan: blocks of 4096 elements scan locally and publish their totals, one block scans
// the totals, a third pass adds the block offsets.  (The single-block kernel above walks 10^6 pair counts in
// 66 serial trips: 0.9 ms of an 12 ms job.)
constexpr int kScanBlock = 4096;
__global__ __launch_bounds__(256) void scan_blocks_kernel(const uint32_t* __restrict__ in, size_t n,
                                                          uint32_t* __restrict__ out, uint32_t* __restrict__ totals) {
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t i0 = (size_t)blockIdx.x * kScanBlock + (size_t)tid * 16;
    uint32_t v[16], tot = 0;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        v[e] = i0 + e < n ? in[i0 + e] : 0u;
        tot += v[e];
    }
    uint32_t inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    uint32_t run = woff + inc - tot;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        if (i0 + e < n) out[i0 + e] = run;
        run += v[e];
    }
    if (tid == 255) totals[blockIdx.x] = run;
}
__global__ void scan_add_kernel(uint32_t* __restrict__ out, size_t n, const uint32_t* __restrict__ block_off) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += block_off[i / kScanBlock];
    if (i == 0) out[n] = block_off[(n + kScanBlock - 1) / kScanBlock];   // grand total behind the last element
}
// out[0..n] (n + 1 entries); tmp holds 2 x (blocks + 1) words
void launch_exclusive_scan(const uint32_t* in, size_t n, uint32_t* out, uint32_t* tmp, hipStream_t stream) {
    const size_t blocks = (n + kScanBlock - 1) / kScanBlock;
    if (blocks <= 1) {
        hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, stream, in, n, out);
        return;
    }
    uint32_t* totals = tmp;
    uint32_t* offs = tmp + blocks + 1;
    hipLaunchKernelGGL(scan_blocks_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, n, out, totals);
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t*)totals, blocks, offs);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, n,
                       (const uint32_t*)offs);
}

__global__ void wang_compact_kernel(const uint32_t* __restrict__ sel_cnt, const uint32_t* __restrict__ sel_off,
                                    const uint32_t* __restrict__ sel_t, const uint32_t* __restrict__ sel_k,
                                    const float* __restrict__ sel_p, uint32_t n_sec, uint32_t pps,
                                    uint32_t* __restrict__ pt, uint32_t* __restrict__ pk, float* __restrict__ pp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_sec * pps) return;
    const uint32_t sec = (uint32_t)(i / pps), j = (uint32_t)(i - (size_t)sec * pps);
    if (j >= sel_cnt[sec]) return;
    const uint32_t o = sel_off[sec] + j;
    pt[o] = sel_t[i];
    pk[o] = sel_k[i];
    pp[o] = sel_p[i];
}

// ---- A6: pairing (src/modality/audio.rs:965-1003) ------------------------------------------------
template <bool EMIT>
__global__ void wang_pair_kernel(const uint32_t* __restrict__ pt, const uint32_t* __restrict__ pk,
                                 const float* __restrict__ pp, const uint32_t* __restrict__ np_ptr,
                                 uint32_t fan_out, uint32_t zone_t, uint32_t zone_f, float floor_p,
                                 uint32_t* __restrict__ counts, const uint32_t* __restrict__ offs,
                                 uint32_t* __restrict__ out, size_t cap) {
    const uint32_t np = *np_ptr;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    uint32_t taken = 0;
    if (pp[i] >= floor_p) {
        const uint32_t ta = pt[i], ka = pk[i];
        size_t o = EMIT ? offs[i] : 0;
        for (uint32_t j = i + 1; j < np && taken < fan_out; j++) {
            const int32_t dt = (int32_t)pt[j] - (int32_t)ta;
            if (dt <= 0) continue;
            if (dt > (int32_t)zone_t) break;
            int32_t df = (int32_t)pk[j] - (int32_t)ka;
            df = df < 0 ? -df : df;
            if (df > (int32_t)zone_f) continue;
            if (EMIT && o < cap) {
                out[2 * o] = (ka << 23) | (pk[j] << 14) | ((uint32_t)dt & 0x3fffu);
                out[2 * o + 1] = ta;
            }
            o++;
            taken++;
        }
    }
    if (!EMIT) counts[i] = taken;
}

__global__ void copy_u32_kernel(const uint32_t* __restrict__ src, uint64_t* __restrict__ dst) { *dst = *src; }

// ---- A8 ------------------------------------------------------------------------------------
__global__ void haitsma_bits_kernel(const float* __restrict__ E, size_t first, size_t n, uint32_t* __restrict__ out) {
    // E holds frames [first - 1, first + n) when first > 0 (row 0 = previous frame), else [0, n)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t row = first > 0 ? i + 1 : i;
    const float* cur = E + row * kHkBands;
    const bool has_prev = first > 0 || i > 0;
    const float* prv = cur - kHkBands;
    uint32_t h = 0;
#pragma unroll
    for (int b = 0; b < 32; b++) {
        const float c = cur[b] - cur[b + 1];
        const float p = has_prev ? prv[b] - prv[b + 1] : 0.0f;
        const float dd = c - p;
        if (dd > 0.0f) h |= 1u << b;
    }
    out[first + i] = h;
}

inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

size_t audio_resample_len(size_t n, uint32_t sr_in, uint32_t sr_out) {
    return (size_t)(((unsigned __int128)n * sr_out) / sr_in);
}

int launch_resample_linear(const float* in, size_t n, uint32_t sr_in, uint32_t sr_out, float* out,
                           hipStream_t stream) {
    const size_t m = audio_resample_len(n, sr_in, sr_out);
    if (m == 0) return 0;
    hipLaunchKernelGGL(resample_linear_kernel, dim3(blocks_for(m, 256)), dim3(256), 0, stream, in, n, sr_in, sr_out,
                       out, m);
    return 0;
}

size_t audio_stft_frames(size_t n, int N, int hop) { return n >= (size_t)N ? 1 + (n - N) / hop : 0; }

// ---- Wang orchestration ------------------------------------------------------------------------
constexpr size_t kChunkFrames = 32768;  // Haitsma band energies are produced in chunks of 4 x this many frames

WangWs wang_ws_layout(size_t n_samples, uint32_t pps) {
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    WangWs w;
    w.frames = audio_stft_frames(n_samples, kWangN, kWangHop);
    w.n_sec = w.frames ? (uint32_t)(((w.frames - 1) * kWangHop) / kWangSr + 1) : 0;
    size_t off = 0;
    w.P = off;        // (the spectrogram is no longer spilled: wang_stream_kernel)
    w.rowmax = off;
    w.cand_cnt = off; off = align(off + (size_t)w.n_sec * 4);
    w.cand_t = off;   off = align(off + (size_t)w.n_sec * kCandCap * 4);
    w.cand_k = off;   off = align(off + (size_t)w.n_sec * kCandCap * 4);
    w.cand_p = off;   off = align(off + (size_t)w.n_sec * kCandCap * 4);
    w.sel_cnt = off;  off = align(off + ((size_t)w.n_sec + 1) * 4);
    w.sel_off = off;  off = align(off + ((size_t)w.n_sec + 1) * 4);
    w.sel_t = off;    off = align(off + (size_t)w.n_sec * pps * 4);
    w.sel_k = off;    off = align(off + (size_t)w.n_sec * pps * 4);
    w.sel_p = off;    off = align(off + (size_t)w.n_sec * pps * 4);
    const size_t maxp = (size_t)w.n_sec * pps;
    w.pt = off;       off = align(off + maxp * 4);
    w.pk = off;       off = align(off + maxp * 4);
    w.pp = off;       off = align(off + maxp * 4);
    w.pair_cnt = off; off = align(off + (maxp + 1) * 4);
    w.pair_off = off; off = align(off + (maxp + 1) * 4);
    w.scan_tmp = off; off = align(off + 2 * (maxp / 4096 + 4) * 4);
    w.total = off + 256;
    return w;
}

int launch_wang(const float* pcm8k, size_t n, uint32_t fan_out, uint32_t zone_t, uint32_t zone_f, uint32_t pps,
                float floor_power, uint8_t* ws, const WangWs& w, uint32_t* out, size_t cap, uint64_t* out_count,
                hipStream_t stream) {
    if (w.frames == 0 || pps == 0) {
        (void)hipMemsetAsync(out_count, 0, 8, stream);
        return 0;
    }
    auto f32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    auto u32 = [&](size_t off) { return reinterpret_cast<uint32_t*>(ws + off); };
    (void)hipMemsetAsync(u32(w.cand_cnt), 0, (size_t)w.n_sec * 4, stream);
    const size_t lds = sizeof(WangStreamLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wang_stream_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const uint32_t seg = wang_segment(w.frames);
    hipLaunchKernelGGL(wang_stream_kernel, dim3((unsigned)((w.frames + seg - 1) / seg)), dim3(kSW * 64), lds, stream, pcm8k,
                       w.frames, seg, u32(w.cand_cnt), u32(w.cand_t), u32(w.cand_k), f32(w.cand_p));
    hipLaunchKernelGGL(wang_select_kernel, dim3(w.n_sec), dim3(64), 0, stream, u32(w.cand_cnt), u32(w.cand_t),
                       u32(w.cand_k), f32(w.cand_p), pps, u32(w.sel_cnt), u32(w.sel_t), u32(w.sel_k), f32(w.sel_p));
    launch_exclusive_scan(u32(w.sel_cnt), (size_t)w.n_sec, u32(w.sel_off), u32(w.scan_tmp), stream);
    hipLaunchKernelGGL(wang_compact_kernel, dim3(blocks_for((size_t)w.n_sec * pps, 256)), dim3(256), 0, stream,
                       u32(w.sel_cnt), u32(w.sel_off), u32(w.sel_t), u32(w.sel_k), f32(w.sel_p), w.n_sec, pps,
                       u32(w.pt), u32(w.pk), f32(w.pp));
    const size_t maxp = (size_t)w.n_sec * pps;
    const uint32_t* np_ptr = u32(w.sel_off) + w.n_sec;  // total peaks
    (void)hipMemsetAsync(u32(w.pair_cnt), 0, (maxp + 1) * 4, stream);
    hipLaunchKernelGGL(wang_pair_kernel<false>, dim3(blocks_for(maxp, 256)), dim3(256), 0, stream, u32(w.pt),
                       u32(w.pk), f32(w.pp), np_ptr, fan_out, zone_t, zone_f, floor_power, u32(w.pair_cnt),
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)0);
    launch_exclusive_scan(u32(w.pair_cnt), maxp, u32(w.pair_off), u32(w.scan_tmp), stream);
    hipLaunchKernelGGL(wang_pair_kernel<true>, dim3(blocks_for(maxp, 256)), dim3(256), 0, stream, u32(w.pt),
                       u32(w.pk), f32(w.pp), np_ptr, fan_out, zone_t, zone_f, floor_power, (uint32_t*)nullptr,
                       (const uint32_t*)u32(w.pair_off), out, cap);
    hipLaunchKernelGGL(copy_u32_kernel, dim3(1), dim3(1), 0, stream, (const uint32_t*)(u32(w.pair_off) + maxp),
                       out_count);
    return 0;
}

// ---- Haitsma orchestration -----------------------------------------------------------------------
size_t haitsma_ws_bytes(size_t n5k) {
    const size_t frames = audio_stft_frames(n5k, kHkN, kHkHop);
    const size_t chunk = frames < kChunkFrames * 4 ? frames : kChunkFrames * 4;
    return (chunk + 1) * kHkBands * 4 + 64 * 4 + 1024;
}

int launch_haitsma(const float* pcm5k, size_t n, const uint32_t* h_edges, uint8_t* ws, uint32_t* out,
                   hipStream_t stream) {
    const size_t frames = audio_stft_frames(n, kHkN, kHkHop);
    if (frames == 0) return 0;
    uint32_t* d_edges = reinterpret_cast<uint32_t*>(ws);
    float* E = reinterpret_cast<float*>(ws + 64 * 4);
    (void)hipMemcpyAsync(d_edges, h_edges, (kHkBands + 1) * 4, hipMemcpyHostToDevice, stream);
    const size_t lds = sizeof(FftLds<kHkN>);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_power_kernel<kHkN, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const size_t chunk = kChunkFrames * 4;
    for (size_t e0 = 0; e0 < frames; e0 += chunk) {
        const size_t e1 = e0 + chunk < frames ? e0 + chunk : frames;
        const size_t w0 = e0 > 0 ? e0 - 1 : 0;  // one frame of history for the time difference
        const size_t wn = e1 - w0;
        unsigned grid = blocks_for(wn, kFftWaves);
        if (grid > 256) grid = 256;
        hipLaunchKernelGGL((stft_power_kernel<kHkN, true>), dim3(grid), dim3(kFftWaves * 64), lds, stream, pcm5k, w0, wn, kHkHop,
                           E, (const uint32_t*)d_edges, (float*)nullptr);
        hipLaunchKernelGGL(haitsma_bits_kernel, dim3(blocks_for(e1 - e0, 256)), dim3(256), 0, stream, E, e0, e1 - e0,
                           out);
    }
    return 0;
}

}  // namespace ucfp
