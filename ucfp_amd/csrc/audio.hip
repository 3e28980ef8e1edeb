// audio.hip -- Wang landmark hashes and Haitsma-Kalker sub-fingerprints for gfx950.
//
// Replaces the arithmetic behind audio::fingerprint_wang_with (src/modality/audio.rs:64-98) and
// audio::fingerprint_haitsma_with (audio.rs:181-224), i.e. audiofp's Wang::extract /
// Haitsma::extract / dsp::resample::linear.  Spec: DESIGN.md "Audio spec" A1..A8; CPU statement:
// oracle/ (audio).  Every float step is one IEEE f32 operation in the order the oracle uses (the
// library is built with -ffp-contract=off), so the integer outputs are bit-identical.
//
//   resample_linear   A1   one thread per output sample (stand-alone entry; the Wang path resamples inside its
//                          stream kernel)
//   wave_rfft_power<N> A2-3 ONE WAVE PER FRAME, real-input route: the N real samples are packed into N/2 complex points
//                          (8 / 16 per lane in registers, packed-f32 butterflies), transformed by an N/2-point radix-2
//                          FFT whose first transpose runs on v_permlane32/16_swap + bank-masked DPP and whose second
//                          goes through LDS, then untangled into the power spectrum; window, twiddles and untangling
//                          factors are loop-invariant per lane and live in registers (Wang)
//   wang_stream       A1+3+5 Wang over a RAGGED BATCH of clips: a workgroup owns a segment of one clip, stages (and, if
//                          the input is not at 8 kHz, linearly resamples) the samples of its frames in an LDS ring,
//                          streams the frames through a ring of row maxima and judges peaks one round behind
//                          (separable neighbourhood maximum + exact tie rule); nothing is spilled
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
constexpr int kHkN = 2048, kHkHop = 64, kHkBands = 33, kHkSr = 5000;
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

// ---- A2/A3: one wave per frame, real-input FFT register-tiled -------------------------------------
// N real samples -> M = N/2 complex points z[n] = (x[2n] wh[2n], x[2n+1] wh[2n+1]) (wh = the HALVED Hann window)
// -> M-point radix-2 DIT FFT -> untangling -> P[k] = |X[k]|^2, k < M.  Exactly the operations of the oracle
// (oracle/ucfp_oracle_audio.c frame_power), only their placement changes.  M = 64 E, E = 8 (Wang, N = 1024) or
// 16 (Haitsma, N = 2048) complex values per lane, B = log2 E; point p (a bit-reversed sample-pair index) sits at
//   phase 1  register i = p[B-1:0];   lane bits [5:6-B] = p[2B-1:B],  lane bits [5-B:0] = p[.. :2B]    stages 1..B
//   phase 2  register m = p[2B-1:B];  lane bits [5:6-B] = p[B-1:0] ("lo"), low lane bits = "hi"        stages B+1..2B
//   phase 3  register e = p[..:6];    lane = p[5:0]                                                    stages 2B+1..
// Transpose 1 swaps register bit j with lane bit 6-B+j IN REGISTERS: lane bits 5 / 4 with gfx950's
// v_permlane32_swap / v_permlane16_swap (one instruction moves both directions), lane bits 3 / 2 with two
// bank-masked DPP moves (row_ror:8; row_shl:4 + row_shr:4).  Transpose 2 goes through LDS (real and imaginary halves
// one after the other through one padded buffer).  Everything that depends only on the lane -- window, the twiddles
// of phases 2 and 3, the untangling factors -- is loop-invariant and handed in by the caller (RfftConst).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N>
struct Rfft {
    static constexpr int M = N / 2, E = M / 64, B = E == 8 ? 3 : 4, BITS = N == 1024 ? 9 : 10, L6 = 6 - B;
    static constexpr int PADK = E == 8 ? 2 : 4;                 // transpose-2 padding per 2^(2B) points (see t2 below)
    static constexpr int T3 = (1 << (BITS - 6)) - (1 << (2 * B - 6));   // phase-3 twiddles per lane: 7 / 12
    static constexpr int T3_FIRST = 1 << (2 * B - 6);
    static constexpr int BUF_FLOATS = E == 8 ? 1056 : 2080;     // transpose / exchange / spectrum scratch per wave
};
template <int N>
struct RfftConst {
    f32x2 win[Rfft<N>::E];        // halved Hann at this lane's E sample pairs
    f32x2 tw2[Rfft<N>::E - 1];    // stage st in B+1..2B: entry (halfm - 1) + (m0 & (halfm - 1)), halfm = 2^(st-B-1)
    f32x2 tw3[Rfft<N>::T3];       // stage st in 2B+1..BITS: entry (halfe - T3_FIRST) + (e0 & (halfe - 1)), halfe = 2^(st-7)
    f32x2 utw[Rfft<N>::E / 2];    // W_N^k, k = 64 e + lane
};

// sample-pair index n (z[n]) of element i of a lane = bit reversal of p(lane, i); n = (rev_B(i) << (BITS-B)) | pair_base(lane)
template <int N>
__device__ __forceinline__ uint32_t rfft_pair_base(int lane) {
    using R = Rfft<N>;
    const uint32_t p = (((uint32_t)lane >> R::L6) << R::B) | (((uint32_t)lane & ((1u << R::L6) - 1u)) << (2 * R::B));
    return __brev(p) >> (32 - R::BITS);
}
__device__ __forceinline__ float half_hann_at(uint32_t m, int N) {
    const uint32_t tws = 2048u / (uint32_t)N;
    float c = c_tw[(m * tws) & 1023u][0];
    if (m * tws >= 1024u) c = -c;
    const float w = 0.5f - 0.5f * c;
    return 0.5f * w;
}
template <int N>
__device__ __forceinline__ void rfft_consts(int lane, RfftConst<N>& K) {
    using R = Rfft<N>;
    const uint32_t nb = rfft_pair_base<N>(lane);
    const int lo = lane >> R::L6;
#pragma unroll
    for (int i = 0; i < R::E; i++) {
        const uint32_t n = ((__brev((uint32_t)i) >> (32 - R::B)) << (R::BITS - R::B)) | nb;
        K.win[i] = f32x2{half_hann_at(2 * n, N), half_hann_at(2 * n + 1, N)};
    }
#pragma unroll
    for (int st = R::B + 1; st <= 2 * R::B; st++) {
        const int halfm = 1 << (st - R::B - 1);
#pragma unroll
        for (int j = 0; j < halfm; j++) {
            const int jj = (j << R::B) | lo;
            K.tw2[halfm - 1 + j] = f32x2{c_tw[jj * (2048 >> st)][0], c_tw[jj * (2048 >> st)][1]};
        }
    }
#pragma unroll
    for (int st = 2 * R::B + 1; st <= R::BITS; st++) {
        const int halfe = 1 << (st - 7);
#pragma unroll
        for (int j = 0; j < halfe; j++) {
            const int jj = (j << 6) | lane;
            K.tw3[halfe - R::T3_FIRST + j] = f32x2{c_tw[jj * (2048 >> st)][0], c_tw[jj * (2048 >> st)][1]};
        }
    }
#pragma unroll
    for (int e = 0; e < R::E / 2; e++) {
        const int k = e * 64 + lane;
        K.utw[e] = f32x2{c_tw[k * (2048 / N)][0], c_tw[k * (2048 / N)][1]};
    }
}

// One radix-2 butterfly on (re, im) pairs in packed f32 (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations
// per lane per instruction).  Same operations in the same order as the oracle:
// t1 = xr c, t2 = xi s, t3 = xr s, t4 = xi c, v = (t1 - t2, t3 + t4), u' = u + v, x' = u - v.
__device__ __forceinline__ f32x2 cmul_tw(f32x2 x, f32x2 tw) {      // (t1 - t2, t4 + t3), tw = (c, s)
    const f32x2 a = x * f32x2{tw.x, tw.x};              // (t1, t4): broadcast by op_sel, no move
    // (-t2, t3) = (xi * -s, xr * s) in ONE instruction: swapped x halves by op_sel, the low product's sign by the
    // source modifier (xi * (-s) == -(xi * s) exactly); the compiler spends a negate + a move on this otherwise
    f32x2 b;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(b) : "v"(x), "v"(tw));
    return a + b;
}
// the two packed products of W x, kept apart so that a caller can issue a whole stage's products before the sums
__device__ __forceinline__ void cmul_parts(f32x2 x, f32x2 tw, f32x2& a, f32x2& b) {
    a = x * f32x2{tw.x, tw.x};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(b) : "v"(x), "v"(tw));
}
__device__ __forceinline__ void bfly(f32x2& u, f32x2& x, f32x2 tw) {
    const f32x2 v = cmul_tw(x, tw);
    const f32x2 w = u;
    u = w + v;
    x = w - v;
}
__device__ __forceinline__ void bfly_one(f32x2& u, f32x2& x) {          // twiddle 1: v = x
    const f32x2 w = u, v = x;
    u = w + v;
    x = w - v;
}
// (a.x + b.y, a.y - b.x) and (a.x - b.y, a.y + b.x): b's halves swapped by op_sel, the sign by the source modifier
// (a + (-b) is a - b exactly) -- one packed instruction each, no moves
__device__ __forceinline__ f32x2 pk_add_swz_pm(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_add_swz_mp(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a.x + b.x, a.y - b.y) and (a.x - b.x, a.y + b.y)
__device__ __forceinline__ f32x2 pk_add_pm(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_add_mp(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void bfly_minus_i(f32x2& u, f32x2& x) {      // twiddle -i: v = (xi, -xr)
    const f32x2 w = u, xx = x;
    u = pk_add_swz_pm(w, xx);        // (ur + xi, ui - xr)      = u + v
    x = pk_add_swz_mp(w, xx);        // (ur - xi, ui + xr)      = u - v
}

// (register bit, lane bit LB) transposed: a is the register-bit-0 side, b the register-bit-1 side
template <int LB>
__device__ __forceinline__ void xchg_lane_bit(f32x2& a, f32x2& b) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const uint32_t ua = __float_as_uint(a[c]), ub = __float_as_uint(b[c]);
        uint32_t na, nb;
        if constexpr (LB == 5) {
            const auto r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);   // lanes 32-63 of a <-> lanes 0-31 of b
            na = r[0];
            nb = r[1];
        } else if constexpr (LB == 4) {
            const auto r = __builtin_amdgcn_permlane16_swap(ua, ub, false, false);   // odd rows of a <-> even rows of b
            na = r[0];
            nb = r[1];
        } else if constexpr (LB == 3) {
            // lanes with bit 3 clear (banks 0, 1 of every row) take a of lane ^ 8 into b; the others b of lane ^ 8 into a
            nb = (uint32_t)__builtin_amdgcn_update_dpp((int)ub, (int)ua, 0x128, 0xf, 0x3, false);   // row_ror:8
            na = (uint32_t)__builtin_amdgcn_update_dpp((int)ua, (int)ub, 0x128, 0xf, 0xc, false);
        } else {
            static_assert(LB == 2, "lane bits 2..5 only");
            nb = (uint32_t)__builtin_amdgcn_update_dpp((int)ub, (int)ua, 0x104, 0xf, 0x5, false);   // row_shl:4 -> from lane + 4
            na = (uint32_t)__builtin_amdgcn_update_dpp((int)ua, (int)ub, 0x114, 0xf, 0xa, false);   // row_shr:4 -> from lane - 4
        }
        a[c] = __uint_as_float(na);
        b[c] = __uint_as_float(nb);
    }
}

// The two non-trivial twiddles of stage 3, W8 = (r, -r) and W8^3 = (-r, -r), r = cos(pi/4): the products with -r are
// the negated products with r, so one broadcast pair (r, r) and source modifiers serve both
// rr lives in a scalar register pair (it is the same for every lane; vector registers are what this kernel is short of)
__device__ __forceinline__ void cmul_w8_parts(f32x2 x, f32x2 rr, bool third, f32x2& a, f32x2& b) {
    if (third) asm("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a) : "v"(x), "s"(rr));   // (xr * -r, xi * -r) = (t1, t4)
    else asm("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(x), "s"(rr));                                    // (xr * r, xi * r) = (t1, t4)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(b) : "v"(x), "s"(rr));   // (xi * r, xr * -r) = (-t2, t3)
}

// x: the windowed packed input z at this lane's E points (phase-1 layout).  buf: >= Rfft<N>::BUF_FLOATS floats of this
// wave's LDS, 8-byte aligned; on return buf[0 .. N/2) holds the power spectrum.
// The butterflies of a stage are written as "all products, then all sums": a product and the sum that consumes it
// are then several instructions apart, which is what the packed pipeline wants (back to back the assembler has to pad
// every butterfly with two s_nop, and an instruction slot is an instruction slot in this issue-bound kernel).
template <int N>
__device__ __forceinline__ void wave_rfft_power(f32x2 (&x)[Rfft<N>::E], const RfftConst<N>& K, int lane, float* __restrict__ buf) {
    using R = Rfft<N>;
    constexpr int E = R::E, B = R::B, M = R::M;
    // ---- phase 1: stages 1..B on the register index; stages 1..3 shortcut the twiddles 1 and -i ----
#pragma unroll
    for (int st = 1; st <= B; st++) {
        const int half = 1 << (st - 1), tstep = 2048 >> st;
        f32x2 v[E / 2], pa[E / 2], pb[E / 2];
#pragma unroll
        for (int i0 = 0, c = 0; i0 < E; i0++) {
            if (i0 & half) continue;
            const int ti = (i0 & (half - 1)) * tstep;
            if (st <= 3 && (ti == 0 || ti == 512)) v[c] = x[i0 + half];                  // taken as is below
            else if (st == 3) cmul_w8_parts(x[i0 + half], f32x2{c_tw[256][0], c_tw[256][0]}, ti == 768, pa[c], pb[c]);
            else cmul_parts(x[i0 + half], f32x2{c_tw[ti][0], c_tw[ti][1]}, pa[c], pb[c]);
            c++;
        }
#pragma unroll
        for (int i0 = 0, c = 0; i0 < E; i0++) {
            if (i0 & half) continue;
            const int ti = (i0 & (half - 1)) * tstep;
            if (!(st <= 3 && (ti == 0 || ti == 512))) v[c] = pa[c] + pb[c];
            c++;
        }
#pragma unroll
        for (int i0 = 0, c = 0; i0 < E; i0++) {
            if (i0 & half) continue;
            const int ti = (i0 & (half - 1)) * tstep;
            const f32x2 w = x[i0];
            if (st <= 3 && ti == 512) {                 // twiddle -i: v = (xi, -xr)
                x[i0] = pk_add_swz_pm(w, v[c]);
                x[i0 + half] = pk_add_swz_mp(w, v[c]);
            } else {
                x[i0] = w + v[c];
                x[i0 + half] = w - v[c];
            }
            c++;
        }
    }
    // ---- transpose 1, in registers: register bit j <-> lane bit 6 - B + j ----
#pragma unroll
    for (int j = 0; j < B; j++) {
#pragma unroll
        for (int r = 0; r < E; r++) {
            if (r & (1 << j)) continue;
            if (6 - B + j == 5) xchg_lane_bit<5>(x[r], x[r | (1 << j)]);
            else if (6 - B + j == 4) xchg_lane_bit<4>(x[r], x[r | (1 << j)]);
            else if (6 - B + j == 3) xchg_lane_bit<3>(x[r], x[r | (1 << j)]);
            else xchg_lane_bit<2>(x[r], x[r | (1 << j)]);
        }
    }
    // ---- phase 2: stages B+1..2B on the register index m ----
#pragma unroll
    for (int st = B + 1; st <= 2 * B; st++) {
        const int halfm = 1 << (st - B - 1);
        // two butterflies at a time: products, then the sums that consume them (a product and its sum must not be
        // adjacent, and more than two in flight would cost registers this kernel does not have)
        int ml[E / 2];
#pragma unroll
        for (int m0 = 0, c = 0; m0 < E; m0++)
            if (!(m0 & halfm)) ml[c++] = m0;
#pragma unroll
        for (int c = 0; c < E / 2; c += 2) {
            f32x2 pa0, pb0, pa1, pb1;
            cmul_parts(x[ml[c] + halfm], K.tw2[halfm - 1 + (ml[c] & (halfm - 1))], pa0, pb0);
            cmul_parts(x[ml[c + 1] + halfm], K.tw2[halfm - 1 + (ml[c + 1] & (halfm - 1))], pa1, pb1);
            const f32x2 v0 = pa0 + pb0, v1 = pa1 + pb1;
            const f32x2 w0 = x[ml[c]], w1 = x[ml[c + 1]];
            x[ml[c]] = w0 + v0;
            x[ml[c] + halfm] = w0 - v0;
            x[ml[c + 1]] = w1 + v1;
            x[ml[c + 1] + halfm] = w1 - v1;
        }
    }
    // ---- transpose 2 through LDS as (re, im) pairs: p = hi 2^2B + m E + lo  ->  p = 64 e + lane; a point p is stored
    // at pair index p + PADK (p >> 2B), which spreads the 16 lanes of a ds_write_b64 group over the 32 banks ----
    const int lo = lane >> R::L6, hi = lane & ((1 << R::L6) - 1);
    f32x2* tb = reinterpret_cast<f32x2*>(buf);
    const int wbase = hi * (E * E + R::PADK) + lo;
#pragma unroll
    for (int m = 0; m < E; m++) tb[wbase + m * E] = x[m];
    wave_lds_fence();
#pragma unroll
    for (int e = 0; e < E; e++) x[e] = tb[e * 64 + lane + R::PADK * ((e * 64) >> (2 * B))];
    // ---- phase 3: stages 2B+1..BITS on the register index e ----
#pragma unroll
    for (int st = 2 * B + 1; st <= R::BITS; st++) {
        const int halfe = 1 << (st - 7);
        int el[E / 2];
#pragma unroll
        for (int e0 = 0, c = 0; e0 < E; e0++)
            if (!(e0 & halfe)) el[c++] = e0;
#pragma unroll
        for (int c = 0; c < E / 2; c += 2) {
            f32x2 pa0, pb0, pa1, pb1;
            cmul_parts(x[el[c] + halfe], K.tw3[halfe - R::T3_FIRST + (el[c] & (halfe - 1))], pa0, pb0);
            cmul_parts(x[el[c + 1] + halfe], K.tw3[halfe - R::T3_FIRST + (el[c + 1] & (halfe - 1))], pa1, pb1);
            const f32x2 v0 = pa0 + pb0, v1 = pa1 + pb1;
            const f32x2 w0 = x[el[c]], w1 = x[el[c + 1]];
            x[el[c]] = w0 + v0;
            x[el[c] + halfe] = w0 - v0;
            x[el[c + 1]] = w1 + v1;
            x[el[c + 1] + halfe] = w1 - v1;
        }
    }
    // ---- untangle.  Z[k] sits in register e of lane l for k = 64 e + l; its partner Z[M - k] in register E-1-e of
    // lane 64 - l (lane 0: register E - e, and Z[0] pairs with itself).  The upper half of the spectrum goes through
    // LDS as float2 slots (bin b -> slot b - M/2; slot M/2 holds Z[0]) and every lane reads its E/2 partners.
    // Writes that only lane 0 owes are made by every lane, the others into a spare slot of their own: a store under a
    // lane condition costs a save / branch / restore of the exec mask, the spare store nothing. ----
    wave_lds_fence();
    f32x2* ex = reinterpret_cast<f32x2*>(buf);
#pragma unroll
    for (int e = E / 2; e < E; e++) ex[(e - E / 2) * 64 + lane] = x[e];
    ex[lane == 0 ? M / 2 : M / 2 + 1 + lane] = x[0];                     // spare slots M/2 + 2 .. M/2 + 64
    wave_lds_fence();
    f32x2 y[E / 2];
#pragma unroll
    for (int e = 0; e < E / 2; e++) y[e] = ex[M / 2 - 64 * e - lane];
    wave_lds_fence();
    // P[k] and P[M-k] from one (A, T)
    f32x2 ta[E / 2], tt[E / 2];
#pragma unroll
    for (int e = 0; e < E / 2; e++) {
        ta[e] = pk_add_pm(x[e], y[e]);                       // A = (Zr + Yr, Zi - Yi)
        tt[e] = pk_add_mp(x[e], y[e]);                       // B = (Zr - Yr, Zi + Yi)
    }
    {
        f32x2 pa[E / 2], pb[E / 2];
#pragma unroll
        for (int e = 0; e < E / 2; e++) cmul_parts(tt[e], K.utw[e], pa[e], pb[e]);
#pragma unroll
        for (int e = 0; e < E / 2; e++) tt[e] = pa[e] + pb[e];               // T = W B = (Tr, Ti)
    }
    float pk[E / 2], pmk[E / 2];
#pragma unroll
    for (int e = 0; e < E / 2; e++) {
        const f32x2 xk = pk_add_swz_pm(ta[e], tt[e]);        // X[k]   = (Ar + Ti, Ai - Tr)
        const f32x2 xm = pk_add_swz_mp(ta[e], tt[e]);        // X[M-k] = (Ar - Ti, Ai + Tr) (conjugate)
        const f32x2 s1 = xk * xk, s2 = xm * xm;
        pk[e] = s1.x + s1.y;
        pmk[e] = s2.x + s2.y;
    }
    // the self-paired bin M/2 (register E/2 of lane 0): A = (2 Zr, 0), B = (0, 2 Zi), W = -i  =>  |X|^2 = (2 Zr)^2 + (2 Zi)^2
    const f32x2 zm = x[E / 2] + x[E / 2];
    const f32x2 sm = zm * zm;
    const float pmid = sm.x + sm.y;
#pragma unroll
    for (int e = 0; e < E / 2; e++) {
        buf[e * 64 + lane] = pk[e];
        buf[M - e * 64 - lane] = pmk[e];                     // e = 0, lane 0: index M, a spare word
    }
    buf[lane == 0 ? M / 2 : M + 1 + lane] = pmid;            // spare words M + 2 .. M + 64
    wave_lds_fence();
}

// ---- Haitsma STFT: band energies per frame --------------------------------------------------------
constexpr int kFftWaves = 8;   // 2 waves per SIMD: the 2048-sample transform keeps ~100 loop-invariant registers per lane
template <int N>
struct FftLds {
    float buf[kFftWaves][Rfft<N>::BUF_FLOATS];
};

// src_map (optional, ragged batches): sample offset of every frame in x, top bit = first frame of its clip, all ones =
// past the batch's last frame; without it frame f starts at x + f * hop.
constexpr uint64_t kFrameFirst = 1ull << 63, kFrameNone = ~0ull;
template <int N, bool HAITSMA>
__global__ __launch_bounds__(kFftWaves * 64) void stft_power_kernel(const float* __restrict__ x, size_t first_frame,
                                                         size_t n_frames, int hop, float* __restrict__ out,
                                                         const uint32_t* __restrict__ edges,
                                                         float* __restrict__ rowmax_out,
                                                         const uint64_t* __restrict__ src_map = nullptr) {
    using R = Rfft<N>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    FftLds<N>& L = *reinterpret_cast<FftLds<N>*>(lds_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    RfftConst<N> K;
    rfft_consts<N>(lane, K);
    float* buf = L.buf[wave];
    const uint32_t nb = rfft_pair_base<N>(lane);
    // frames are dealt to waves round-robin over the whole grid
    const uint32_t k0 = lane < kHkBands ? edges[lane] : 0u, k1 = lane < kHkBands ? edges[lane + 1] : 0u;
    const size_t step = (size_t)gridDim.x * kFftWaves;
    for (size_t f = (size_t)blockIdx.x * kFftWaves + wave; f < n_frames; f += step) {
        const float* src = x + (first_frame + f) * (size_t)hop;
        if (src_map) {
            const uint64_t m = src_map[first_frame + f];
            if (m == kFrameNone) continue;
            src = x + (m & ~kFrameFirst);
        }
        f32x2 z[R::E];
#pragma unroll
        for (int i = 0; i < R::E; i++) {
            const uint32_t n = ((__brev((uint32_t)i) >> (32 - R::B)) << (R::BITS - R::B)) | nb;
            z[i] = f32x2{src[2 * n], src[2 * n + 1]};
        }
#pragma unroll
        for (int i = 0; i < R::E; i++) z[i] = z[i] * K.win[i];
        wave_rfft_power<N>(z, K, lane, buf);
        {
            // lane b sums band b sequentially (same order as the oracle); eight spectrum reads are in flight at a time
            const float* pw = buf;
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

// ---- A1 + A3 + A5 fused: STFT frames streamed through LDS, peaks picked without spilling the spectrogram ----
// The input is a RAGGED BATCH of clips (one clip = the single-stream entry points).  A workgroup (kSW waves) owns a
// segment of `seg` consecutive frames of ONE clip and walks it in rounds of kSW frames, one FFT per wave (plus kRT
// halo frames on each side, recomputed: 14 / seg extra).  Per round the workgroup also brings the next kSW * hop
// samples at 8 kHz into an LDS ring -- two per thread: copied, or linearly interpolated from a clip at another rate
// (A1: position exact in integers, x0 + (x1 - x0) * frac) -- so HBM sees every source sample once and the frames
// (each sample belongs to 8 of them) are cut from LDS.  Of every frame only two things survive in LDS:
//   ring    its +-kRK-bin running maximum ("row maximum"), 2 KiB, in a ring of kRing frames
//   plist   its row-local peak candidates: bins with P == row maximum > 0 and no equal value among the kRK
//           bins below (the same-row half of the tie rule) -- two of them are >= 16 bins apart, so <= 32 per frame
// Once the rows t-kRT .. t+kRT are in the ring, a candidate (t, k, v) is a peak iff v equals the largest
// of those 15 row maxima at bin k and none of the EARLIER rows' maxima equals v (an equal cell earlier in
// (t, k) order wins the tie; rows outside [0, total) duplicate rows inside the window, so they are simply
// skipped).  HBM sees the samples once and the peaks -- not 2 x 4 B x 512 bins per frame of spilled spectrum.
constexpr int kSegMax = 2048;   // frames per workgroup segment: a segment recomputes 14 halo frames and runs ~2 extra rounds to drain the
                                // judge, so long inputs want long segments (wang_segment); short ones get shorter segments so that ~1000 workgroups exist
constexpr int kSW = 12;      // waves per workgroup = frames in flight
constexpr int kRing = 38;    // >= 2 kRT + 2 kSW: the rows being judged (one round behind) + the rows being produced
constexpr int kPl = 32;      // row-local candidates per frame: two of them are always >= 16 bins apart
constexpr int kSmpRing = 4096;                // 8 kHz samples staged per workgroup
constexpr int kBatch = kSW * kWangHop;        // new samples per round = 2 per thread
constexpr int kPrologue = 2560;               // samples staged before round 0: >= (kSW - 1) hop + N = 2432 (round 0 can
                                              // be cut), <= kSmpRing - kBatch (a round's batch never lands on samples
                                              // another wave may still be cutting its frame from)
static_assert(kBatch == 2 * kSW * 64 && kPrologue >= (kSW - 1) * kWangHop + kWangN && kPrologue + kBatch <= kSmpRing, "");

struct WangClip {
    uint64_t src_off;   // first source sample of the clip in the batch buffer
    uint64_t src_n;     // source samples
    uint64_t n8k;       // samples at 8 kHz (= src_n when the input is at 8 kHz)
    uint32_t frames;    // STFT frames
    uint32_t n_sec;     // seconds holding at least one frame
    uint32_t n_seg;     // workgroup segments
    uint32_t pad;
};

struct WangStreamLds {
    float buf[kSW][Rfft<kWangN>::BUF_FLOATS];   // per wave: transpose 2, untangling exchange, power spectrum, row-maximum scratch
    float smp[kSmpRing + 64];     // the workgroup's 8 kHz samples, index swizzled (smp_swz); + a spare tail for stores that own nothing
    float ring[kRing][kWangBins];
    uint32_t pl_cnt[kRing];
    uint32_t pl_k[kRing][kPl];
    float pl_v[kRing][kPl];
    float2 win[kWangN / 128][64]; // halved Hann at a lane's 8 sample pairs (registers go to the FFT's twiddles)
    uint32_t spare_k[kSW][kPl];   // where the lanes WITHOUT a candidate store (an unconditional store is cheaper than a branch)
    float spare_v[kSW][kPl];
};


// A frame's pairs (x[2n], x[2n+1]) are read by ds_read_b64 with n = (register part) | (6 lane-dependent bits): the 32
// lanes of a group then touch dwords {0-15, 64-79, 32-47, 96-111} + const, i.e. two bank quarters twice.  XOR-ing
// dword-index bit 4 with bit 6 gives every 16-dword piece its own quarter of the 64 banks.
__device__ __forceinline__ uint32_t smp_swz(uint32_t s) { return s ^ ((s >> 2) & 16u); }

// Frames per workgroup segment.  Long inputs: as close to kSegMax as gives a whole number of rounds of 256 workgroups
// (one per CU) -- 4400 segments of 512 frames would run 17 full rounds and an 18th with 48 workgroups; 4608 of 489 run
// 18 full ones.  Short inputs get shorter segments so that ~1000 workgroups exist.
inline uint32_t wang_segment(size_t frames) {
    size_t seg = (frames + 1023) / 1024;
    if (seg < 48) seg = 48;
    if (seg > (size_t)kSegMax) {
        const size_t rounds = (frames + (size_t)256 * kSegMax - 1) / ((size_t)256 * kSegMax);
        seg = (frames + 256 * rounds - 1) / (256 * rounds);
    }
    return (uint32_t)seg;
}

template <bool RESAMPLE>
__global__ __launch_bounds__(kSW * 64) void wang_stream_kernel(const float* __restrict__ pcm,
                                                          const WangClip* __restrict__ clips,
                                                          const uint32_t* __restrict__ seg_clip,
                                                          const uint32_t* __restrict__ seg_base,
                                                          const uint32_t* __restrict__ sec_base,
                                                          const uint32_t* __restrict__ n_segs_total, uint32_t seg,
                                                          uint32_t sr_in, uint32_t* __restrict__ cand_cnt,
                                                          uint32_t* __restrict__ cand_t,
                                                          uint32_t* __restrict__ cand_k, float* __restrict__ cand_p) {
    // This kernel is bound by instruction issue (every instruction of any kind costs about the same: measured by
    // padding the loop with scalar or vector adds), so the loop body is written for instruction count: 32-bit indices,
    // loads and stores without lane conditions (clamped addresses, spare slots), immediate LDS offsets.
    if (blockIdx.x >= *n_segs_total) return;       // the grid is a host-side upper bound (whole workgroup leaves)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    WangStreamLds& L = *reinterpret_cast<WangStreamLds*>(lds_raw);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // frame and ring arithmetic stays scalar
    const uint32_t clip = seg_clip[blockIdx.x];
    const WangClip cl = clips[clip];
    const float* __restrict__ x = pcm + cl.src_off;
    const uint32_t sec0 = sec_base[clip];
    RfftConst<kWangN> K;
    rfft_consts<kWangN>(lane, K);
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < kWangN / 128; i++) L.win[i][lane] = make_float2(K.win[i].x, K.win[i].y);
    }
    float* buf = L.buf[wave];
    const float* pw = buf;
    const int total = (int)cl.frames;
    const int s0 = (int)((blockIdx.x - seg_base[clip]) * seg);     // frames [s0, s1) are this segment's to judge
    const int s1 = s0 + (int)seg < total ? s0 + (int)seg : total;
    const int f_lo = s0 - kRT < 0 ? 0 : s0 - kRT;                  // frames [f_lo, f_hi) are computed
    const int f_hi = s1 + kRT < total ? s1 + kRT : total;

    // ---- sample staging.  Everything is kept RELATIVE to the segment's first sample so that the per-thread index
    // arithmetic is 32-bit: b_i = 8 kHz index of the next batch's first sample minus i_base; for a resampled clip that
    // sample sits at source position q_base + b_q + b_r / 8000.  A thread fetches samples 2 tid, 2 tid + 1 of the batch
    // into registers (the loads stay in flight across the FFT) and stores them at the end of the round.  Loads are
    // unconditional at clamped positions: what lies past the clip's last frame is never cut into a frame. ----
    const uint64_t i_base = (uint64_t)f_lo * kWangHop;
    uint64_t q_base = i_base;
    uint32_t b_i = 0, b_q = 0, b_r = 0;
    if (RESAMPLE) {
        const uint64_t num = i_base * sr_in;
        q_base = num / (uint32_t)kWangSr;
        b_r = (uint32_t)(num - q_base * (uint32_t)kWangSr);
    }
    const char* __restrict__ xs = reinterpret_cast<const char*>(x + q_base);       // source sample q_base
    const uint64_t src_left = (RESAMPLE ? cl.src_n : cl.n8k) - q_base;             // >= 1: the segment has a frame
    const uint32_t last_off = (uint32_t)((src_left < (1ull << 29) ? src_left : (1ull << 29)) - 1) * 4u;   // byte offset of the last sample
    float sx0[2], sx1[2], sfr[2];
    const uint32_t j0 = 2u * threadIdx.x;
    auto ld = [&](uint32_t byte_off) { return *reinterpret_cast<const float*>(xs + byte_off); };   // uniform base + 32-bit offset
    auto fetch = [&]() {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t j = j0 + u;
            if (RESAMPLE) {
                const uint32_t num = b_r + j * sr_in;                 // < 2^32: sr_in <= 384 000, j < 1536
                const uint32_t qd = num / (uint32_t)kWangSr;
                const uint32_t rem = num - qd * (uint32_t)kWangSr;
                const uint32_t off = (b_q + qd) * 4u;
                sx0[u] = ld(off < last_off ? off : last_off);
                sx1[u] = ld(off + 4u < last_off ? off + 4u : last_off);
                // = (float)((double)rem / 8000.0), the oracle's expression: the reciprocal form gives the same
                // float for every rem in [0, 8000) (checked exhaustively, tests/test_oracle_spec.py)
                sfr[u] = (float)((double)rem * (1.0 / (double)kWangSr));
            } else {
                const uint32_t off = (b_i + j) * 4u;
                sx0[u] = ld(off < last_off ? off : last_off);
            }
        }
    };
    auto store = [&](uint32_t count) {
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (RESAMPLE) {
                const float d = sx1[u] - sx0[u];
                const float mm = d * sfr[u];
                v[u] = sx0[u] + mm;
            } else {
                v[u] = sx0[u];
            }
        }
        // ring position of 8 kHz sample i_base + b_i + j0 (i_base is a multiple of the hop; the ring's size too); a
        // thread past the batch (prologue only) stores into the spare tail of the ring array instead
        uint32_t sidx = smp_swz(((uint32_t)i_base + b_i + j0) & (uint32_t)(kSmpRing - 1));   // even: the pair stays together
        if (count < (uint32_t)kBatch) sidx = j0 < count ? sidx : (uint32_t)kSmpRing + (j0 & 62u);
        *reinterpret_cast<float2*>(&L.smp[sidx]) = make_float2(v[0], v[1]);
        b_i += count;
        if (RESAMPLE) {
            const uint32_t num = b_r + count * sr_in;
            const uint32_t qd = num / (uint32_t)kWangSr;
            b_q += qd;
            b_r = num - qd * (uint32_t)kWangSr;
        }
    };
    fetch();
    store(kPrologue - kBatch);
    fetch();
    store(kBatch);
    __syncthreads();
    // this lane's sample-pair offset inside a frame: dword 2 n, n = (rev(i) << 6) | nb; the swizzle only touches the
    // lane part (2 nb < 128), the register part + the frame start are wave-uniform
    const uint32_t nb2 = smp_swz(2u * rfft_pair_base<kWangN>(lane));

    int slot = (int)((f_lo + wave) % kRing);                       // ring row of frame base + wave
    // a found peak waits one round for its slot: the atomic's round trip overlaps the next FFT.
    // flush_take (the atomic's result, requested a round ago, becomes this lane's offset) + flush_put (the three
    // stores).  The round's sample loads are issued BETWEEN the two: a wait for the atomic must not include them, and
    // nothing may have to wait for the stores' acknowledgement (memory returns count in order).
    bool pend = false, pend_wave = false;      // pend_wave: wave-uniform "an atomic is in flight"
    uint32_t pend_base = 0, pend_rank = 0, pend_off = 0, pend_t = 0, pend_k = 0;
    int pend_leader = 0;
    float pend_v = 0.0f;
    auto flush_take = [&]() {
        if (pend_wave) {
            const uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)pend_base, pend_leader) + pend_rank;
            pend = pend && pos < (uint32_t)kCandCap;
            pend_off = (pend_off + pos) * 4u;                                   // byte offset into cand_*
        }
    };
    auto flush_put = [&]() {
        if (pend_wave && pend) {
            *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(cand_t) + pend_off) = pend_t;
            *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(cand_k) + pend_off) = pend_k;
            *reinterpret_cast<float*>(reinterpret_cast<char*>(cand_p) + pend_off) = pend_v;
        }
        pend = false;
        pend_wave = false;
    };
    for (int base = f_lo; base < s1 + kRT + kSW; base += kSW) {
        const int f = base + wave;
        flush_take();
        // next round's samples: requested now, stored behind the FFT
        fetch();
        flush_put();
        // this round's frame is cut from the sample ring first: the reads are in flight while the judge below runs
        f32x2 z[kWangN / 128];
        if (f < f_hi) {
            // frame start in the ring, in units of 128 dwords (the hop): wave-uniform.  A frame spans 8 units; unless it
            // wraps around the ring's end (7 of 32 start positions) the eight reads are one address + immediates
            const uint32_t fs = ((uint32_t)f * kWangHop & (uint32_t)(kSmpRing - 1)) >> 7;
            if (fs <= (uint32_t)(kSmpRing / 128 - 8)) {
                const float* p0 = &L.smp[(fs << 7) | nb2];
#pragma unroll
                for (int i = 0; i < kWangN / 128; i++) {
                    const uint32_t ri = __brev((uint32_t)i) >> 29;                       // 3-bit reversal: n = (ri << 6) | nb
                    const float2 v = *reinterpret_cast<const float2*>(__builtin_assume_aligned(p0 + (ri << 7), 8));
                    z[i] = f32x2{v.x, v.y};
                }
            } else {
#pragma unroll
                for (int i = 0; i < kWangN / 128; i++) {
                    const uint32_t ri = __brev((uint32_t)i) >> 29;
                    const uint32_t blk = (fs + ri) & (uint32_t)(kSmpRing / 128 - 1);     // scalar
                    const float2 v = *reinterpret_cast<const float2*>(
                        __builtin_assume_aligned(&L.smp[(blk << 7) | nb2], 8));
                    z[i] = f32x2{v.x, v.y};
                }
            }
        }
        // ---- judge frame base + wave - kSW - kRT: its window [t - kRT, t + kRT] was complete at the last barrier ----
        const int t = f - kSW - kRT;
        if (t >= s0 && t < s1) {
            int st = slot - kSW - kRT;
            if (st < 0) st += kRing;
            // The whole wave pays for every instruction here, so the test is kept short: v is the row maximum of its
            // own row at k, hence "v equals the window maximum and no earlier row reaches it" is
            //   max(7 earlier rows at k) < v  and  max(7 later rows at k) <= v.
            // All reads are unconditional (lanes past the list read a stale entry and are masked at the end).
            const uint32_t n = L.pl_cnt[st];
            const uint32_t pk_raw = L.pl_k[st][lane & (kPl - 1)] & (uint32_t)(kWangBins - 1);
            const float pv = L.pl_v[st][lane & (kPl - 1)];
            // lanes past the list follow lane 0's bin: one address, a broadcast -- their own stale bins would scatter
            // over the banks and multiply the cycles of each of the 15 row reads below
            const uint32_t pk = (uint32_t)lane < n ? pk_raw : (uint32_t)__builtin_amdgcn_readfirstlane((int)pk_raw);
            int rs = st - kRT;
            if (rs < 0) rs += kRing;
            float rr[2 * kRT + 1];
            const float* cell = &L.ring[0][pk];
            if (rs + 2 * kRT < kRing) {                    // the window does not wrap: one address, 15 immediates
                const float* c0 = cell + rs * kWangBins;
#pragma unroll
                for (int d = 0; d <= 2 * kRT; d++) rr[d] = c0[d * kWangBins];
            } else {                                       // rows rs .. kRing-1, then 0 ..: two addresses
                const float* c0 = cell + rs * kWangBins;
                const float* c1 = c0 - kRing * kWangBins;
                const int w = kRing - rs;                  // rows d < w sit before the wrap
#pragma unroll
                for (int d = 0; d <= 2 * kRT; d++) rr[d] = (d < w ? c0 : c1)[d * kWangBins];
            }
            if (t < kRT || t + kRT >= total) {             // rows outside [0, total) duplicate rows inside: drop them
#pragma unroll
                for (int d = 0; d <= 2 * kRT; d++) {
                    const int tt = t + d - kRT;
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
            // consumed a round later by flush_take(), so the L2 round trip overlaps the next FFT.  The address is hidden
            // from the compiler: for a uniform address it aggregates by itself and reads the result back at once
            // (s_waitcnt vmcnt(0) + v_readfirstlane right behind the atomic), which parks every wave for the trip.
            const uint64_t pm = __ballot(is_peak);
            if (pm) {
                const uint32_t sec = sec0 + ((uint32_t)t * kWangHop) / kWangSr;
                pend = is_peak;
                pend_off = sec * (uint32_t)kCandCap;
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
#pragma unroll
            for (int i = 0; i < kWangN / 128; i++) {
                const float2 wv = L.win[i][lane];
                z[i] = z[i] * f32x2{wv.x, wv.y};
            }
            wave_rfft_power<kWangN>(z, K, lane, buf);
            // Row maximum over +-kRK bins and the same-row tie test, blocked: lane L owns bins 8L .. 8L+7.  The
            // window [k-15, k+15] of bin k = 8L + j is  suffix_{L-2}[j+1] u block_{L-1} u block_L u block_{L+1} u
            // prefix_{L+2}[j-1], so with C = max(block_{L-1}, block_L, block_{L+1}) the row maximum is
            // max3(C, suffix_{L-2}[j+1], prefix_{L+2}[j-1]): 14 maxima for the prefix / suffix tables, one LDS exchange,
            // 9 max3 -- instead of 31 taps per bin.  A lane whose neighbour does not exist reads ITS OWN tables
            // instead (<= its own block maximum, already in C: neutral), so no edge selects are needed here.
            // A row-local candidate (P == row maximum > 0, no equal value among the 15 bins below) can only be the
            // FIRST occurrence js of the block maximum v, and it is one iff
            //     v > max(suffix_{L-2}[js+1], block_{L-1}, 0)   and   v >= max(block_{L+1}, prefix_{L+2}[js-1])
            // (the bins of its own block before js are < v, those after are <= v, by the choice of js).
            float* row = L.ring[slot];
            float* sx = buf;              // scratch OVER the spectrum (b8 is read first; the LDS keeps a wave's order):
            float* px = sx + 64 * 7;      // [lane][7] suffix 1..7, [lane][9] prefix 0..7 (row strides 7 and 9: conflict-free)
            float b8[8], pre[8], suf[8];
            {
                const float4 lo4 = *reinterpret_cast<const float4*>(pw + 8 * lane);
                const float4 hi4 = *reinterpret_cast<const float4*>(pw + 8 * lane + 4);
                b8[0] = lo4.x; b8[1] = lo4.y; b8[2] = lo4.z; b8[3] = lo4.w;
                b8[4] = hi4.x; b8[5] = hi4.y; b8[6] = hi4.z; b8[7] = hi4.w;
            }
            pre[0] = b8[0];
            uint32_t js = 0;
#pragma unroll
            for (int j = 1; j < 8; j++) {
                js = b8[j] > pre[j - 1] ? (uint32_t)j : js;
                pre[j] = fmaxf(pre[j - 1], b8[j]);
            }
            suf[7] = b8[7];
#pragma unroll
            for (int j = 6; j >= 0; j--) suf[j] = fmaxf(suf[j + 1], b8[j]);
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (j > 0) sx[7 * lane + j - 1] = suf[j];
                px[9 * lane + j] = pre[j];
            }
            wave_lds_fence();
            const int lm1 = lane >= 1 ? lane - 1 : lane, lp1 = lane <= 62 ? lane + 1 : lane;
            const int lm2 = lane >= 2 ? lane - 2 : lane, lp2 = lane <= 61 ? lane + 2 : lane;
            const float nb_m1 = px[9 * lm1 + 7], nb_p1 = px[9 * lp1 + 7];     // block maxima of the neighbours
            float s2v[7], p2v[7];
#pragma unroll
            for (int j = 0; j < 7; j++) {
                s2v[j] = sx[7 * lm2 + j];          // suffix_{L-2}[j+1]
                p2v[j] = px[9 * lp2 + j];          // prefix_{L+2}[j]
            }
            const float s2c = sx[7 * lm2 + (int)js];          // suffix_{L-2}[js+1]  (js = 7: no such bin, dropped below)
            const float p2c = px[9 * lp2 + (int)js - 1];      // prefix_{L+2}[js-1]  (js = 0: dropped below)
            const float cmax = fmaxf(fmaxf(nb_m1, pre[7]), nb_p1);
            float rm[8];
            rm[0] = fmaxf(cmax, s2v[0]);
#pragma unroll
            for (int j = 1; j < 7; j++) rm[j] = fmaxf(fmaxf(cmax, s2v[j]), p2v[j - 1]);
            rm[7] = fmaxf(cmax, p2v[6]);
            const float v8 = pre[7];
            const float lo_side = fmaxf((lane >= 2 && js < 7u) ? s2c : 0.0f, lane >= 1 ? nb_m1 : 0.0f);
            const float hi_side = fmaxf(nb_p1, js > 0u ? p2c : 0.0f);
            *reinterpret_cast<float4*>(row + 8 * lane) = make_float4(rm[0], rm[1], rm[2], rm[3]);
            *reinterpret_cast<float4*>(row + 8 * lane + 4) = make_float4(rm[4], rm[5], rm[6], rm[7]);
            // two row-local candidates are >= 16 bins apart, a lane owns 8 bins: at most one per lane, at most kPl per
            // row, and one ballot compacts the row (the list's order is irrelevant: wang_select ranks the peaks).  Every
            // lane stores: a candidate at its rank, the others into this wave's spare slots.
            const bool c = v8 > lo_side && v8 >= hi_side;
            const uint64_t mask = __ballot(c);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            uint32_t* kp = c ? &L.pl_k[slot][pos & (uint32_t)(kPl - 1)] : &L.spare_k[wave][lane & (kPl - 1)];
            float* vp = c ? &L.pl_v[slot][pos & (uint32_t)(kPl - 1)] : &L.spare_v[wave][lane & (kPl - 1)];
            *kp = 8u * (uint32_t)lane + js;
            *vp = v8;
            L.pl_cnt[slot] = (uint32_t)__popcll(mask);       // every lane, the same word
        }
        store(kBatch);     // the samples of the next round's frames (requested at the top of this round)
        __syncthreads();   // the only one per round: rows base .. base + kSW - 1 and the next samples are complete
        slot = slot + kSW >= kRing ? slot + kSW - kRing : slot + kSW;
    }
    flush_take();
    flush_put();
}

// one wave per second: keep the `pps` strongest, ordered by (t, k).  A candidate is (p, tk = t << 9 | k) in one 8-byte
// LDS word, so the rank loop reads one broadcast word per comparison; the survivors (<= pps <= 256) are compacted
// with a ballot and ordered among themselves only.
__global__ __launch_bounds__(64) void wang_select_kernel(const uint32_t* __restrict__ cand_cnt,
                                                         const uint32_t* __restrict__ cand_t,
                                                         const uint32_t* __restrict__ cand_k,
                                                         const float* __restrict__ cand_p, uint32_t pps,
                                                         const uint32_t* __restrict__ n_sec_total,
                                                         uint32_t* __restrict__ sel_cnt, uint32_t* __restrict__ sel_t,
                                                         uint32_t* __restrict__ sel_k, float* __restrict__ sel_p) {
    __shared__ float2 sc[kCandCap];       // (p, bits of tk)
    __shared__ float2 kept[256];
    const uint32_t sec = blockIdx.x;
    const int lane = threadIdx.x;
    if (sec >= *n_sec_total) {            // the grid is a host-side upper bound
        if (lane == 0) sel_cnt[sec] = 0;
        return;
    }
    uint32_t n = cand_cnt[sec];
    n = n < (uint32_t)kCandCap ? n : (uint32_t)kCandCap;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t tk = (cand_t[(size_t)sec * kCandCap + i] << 9) | cand_k[(size_t)sec * kCandCap + i];
        sc[i] = make_float2(cand_p[(size_t)sec * kCandCap + i], __uint_as_float(tk));
    }
    __syncthreads();
    uint32_t nk = 0;                      // survivors so far (wave-uniform)
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + lane;
        const float2 me = sc[i < n ? i : 0];
        const float p = me.x;
        const uint32_t tk = __float_as_uint(me.y);
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++) {
            const float2 o = sc[j];
            const bool before = o.x > p || (o.x == p && __float_as_uint(o.y) < tk);
            rank += before ? 1u : 0u;
        }
        const bool keep = i < n && rank < pps;
        const uint64_t m = __ballot(keep);
        if (keep) kept[nk + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = me;
        nk += (uint32_t)__popcll(m);
    }
    __syncthreads();
    for (uint32_t i = lane; i < nk; i += 64) {
        const float2 me = kept[i];
        const uint32_t tk = __float_as_uint(me.y);
        uint32_t pos = 0;
        for (uint32_t j = 0; j < nk; j++) pos += __float_as_uint(kept[j].y) < tk ? 1u : 0u;
        sel_t[(size_t)sec * pps + pos] = tk >> 9;
        sel_k[(size_t)sec * pps + pos] = tk & 511u;
        sel_p[(size_t)sec * pps + pos] = me.x;
    }
    if (lane == 0) sel_cnt[sec] = nk;
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
                                    const float* __restrict__ sel_p, const uint32_t* __restrict__ sec_clip,
                                    uint32_t n_sec, uint32_t pps, uint32_t* __restrict__ ptk,
                                    float* __restrict__ pp, uint32_t* __restrict__ pc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_sec * pps) return;
    const uint32_t sec = (uint32_t)(i / pps), j = (uint32_t)(i - (size_t)sec * pps);
    if (j >= sel_cnt[sec]) return;
    const uint32_t o = sel_off[sec] + j;
    ptk[o] = (sel_t[i] << 9) | sel_k[i];
    pp[o] = sel_p[i];
    pc[o] = sec_clip[sec];
}

// ---- A6: pairing (src/modality/audio.rs:965-1003) ------------------------------------------------
// ptk = t << 9 | k (t < 2^23 frames = 37 h per clip); the peaks of a clip are contiguous, [.., pend[clip]) ends it
template <bool EMIT>
__global__ void wang_pair_kernel(const uint32_t* __restrict__ ptk, const float* __restrict__ pp,
                                 const uint32_t* __restrict__ pc, const uint32_t* __restrict__ sec_base,
                                 const uint32_t* __restrict__ sel_off, const uint32_t* __restrict__ np_ptr,
                                 uint32_t fan_out, uint32_t zone_t, uint32_t zone_f, float floor_p,
                                 uint32_t* __restrict__ counts, const uint32_t* __restrict__ offs,
                                 uint2* __restrict__ out, size_t cap) {
    const uint32_t np = *np_ptr;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    uint32_t taken = 0;
    if (pp[i] >= floor_p) {
        const uint32_t a = ptk[i];
        const int32_t ta = (int32_t)(a >> 9), ka = (int32_t)(a & 511u);
        const uint32_t end = sel_off[sec_base[pc[i] + 1]];         // first peak of the next clip
        size_t o = EMIT ? offs[i] : 0;
        // the walk is a chain of dependent decisions but not of dependent loads: eight following peaks are fetched at
        // once (a peak past the clip reads as "infinitely late" and ends the walk)
        bool done = false;
        for (uint32_t j0 = i + 1; j0 < end && !done; j0 += 8) {
            uint32_t bb[8];
#pragma unroll
            for (int u = 0; u < 8; u++) bb[u] = j0 + u < end ? ptk[j0 + u] : 0xffffffffu;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (done) continue;
                const uint32_t b = bb[u];
                const int32_t dt = (int32_t)(b >> 9) - ta;
                if (dt <= 0) continue;
                if (dt > (int32_t)zone_t) {
                    done = true;
                    continue;
                }
                const int32_t kb = (int32_t)(b & 511u);
                int32_t df = kb - ka;
                df = df < 0 ? -df : df;
                if (df > (int32_t)zone_f) continue;
                if (EMIT && o < cap) out[o] = make_uint2(((uint32_t)ka << 23) | ((uint32_t)kb << 14) | ((uint32_t)dt & 0x3fffu), (uint32_t)ta);
                o++;
                taken++;
                if (taken >= fan_out) done = true;
            }
        }
    }
    if (!EMIT) counts[i] = taken;
}

__global__ void copy_u32_kernel(const uint32_t* __restrict__ src, uint64_t* __restrict__ dst) { *dst = *src; }

// ---- the ragged batch: clip table, segment -> clip and second -> clip maps, per-clip hash offsets ----
// one thread per clip.  offsets == nullptr: a single clip [0, n_single)
__global__ void wang_clip_prep_kernel(const uint64_t* __restrict__ offsets, uint64_t n_single, uint32_t n_clips,
                                      uint32_t sr_in, uint32_t seg, WangClip* __restrict__ clips,
                                      uint32_t* __restrict__ seg_cnt, uint32_t* __restrict__ sec_cnt) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clips) return;
    WangClip cl;
    cl.src_off = offsets ? offsets[c] : 0;
    cl.src_n = offsets ? offsets[c + 1] - offsets[c] : n_single;
    cl.n8k = sr_in == (uint32_t)kWangSr ? cl.src_n : (uint64_t)(((unsigned __int128)cl.src_n * kWangSr) / sr_in);
    const uint64_t fr = cl.n8k >= (uint64_t)kWangN ? 1 + (cl.n8k - kWangN) / kWangHop : 0;
    cl.frames = (uint32_t)(fr < (1u << 23) ? fr : (1u << 23) - 1);     // (t << 9 | k) packing: 37 h per clip (the C ABI rejects longer single clips)
    cl.n_sec = fr ? (uint32_t)(((fr - 1) * kWangHop) / kWangSr + 1) : 0;
    cl.n_seg = (uint32_t)((fr + seg - 1) / seg);
    cl.pad = 0;
    clips[c] = cl;
    seg_cnt[c] = cl.n_seg;
    sec_cnt[c] = cl.n_sec;
}
// base[0..n_clips] ascending (exclusive scan of counts); entry g of the map = the clip c with base[c] <= g < base[c+1]
__device__ __forceinline__ uint32_t clip_of(const uint32_t* __restrict__ base, uint32_t n_clips, uint32_t g) {
    uint32_t lo = 0, hi = n_clips;                 // first index in (0, n_clips] with base[idx] > g, minus one
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (base[mid + 1] > g) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}
__global__ void wang_clip_map_kernel(const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ sec_base,
                                     uint32_t n_clips, uint32_t* __restrict__ seg_clip, uint32_t* __restrict__ sec_clip) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < seg_base[n_clips]) seg_clip[g] = clip_of(seg_base, n_clips, g);
    if (g < sec_base[n_clips]) sec_clip[g] = clip_of(sec_base, n_clips, g);
}
// hashes of clip c = out[out_off[c] .. out_off[c+1]): the pair offset of the clip's first peak
__global__ void wang_clip_offsets_kernel(const uint32_t* __restrict__ sec_base, const uint32_t* __restrict__ sel_off,
                                         const uint32_t* __restrict__ pair_off, uint32_t n_clips,
                                         uint64_t* __restrict__ out_off) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_clips) return;
    out_off[c] = pair_off[sel_off[sec_base[c]]];
}

// ---- A8 ------------------------------------------------------------------------------------
__global__ void haitsma_bits_kernel(const float* __restrict__ E, size_t first, size_t n, uint32_t* __restrict__ out,
                                    const uint64_t* __restrict__ src_map = nullptr, size_t cap = ~(size_t)0) {
    // E holds frames [first - 1, first + n) when first > 0 (row 0 = previous frame), else [0, n)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || first + i >= cap) return;
    const size_t row = first > 0 ? i + 1 : i;
    const float* cur = E + row * kHkBands;
    bool has_prev = first > 0 || i > 0;
    if (src_map) {
        const uint64_t m = src_map[first + i];
        if (m == kFrameNone) return;
        has_prev = !(m & kFrameFirst);        // a clip's first frame has a zero history (A8), whatever precedes it in the batch
    }
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

// ---- Haitsma over a ragged batch of clips ------------------------------------------------------------------------
// One workgroup: per clip its length at 5 kHz and its frame count, exclusive-scanned into s5_off / fr_off (n_clips + 1
// entries each); fr_off is also the caller's output offset table.
__global__ __launch_bounds__(1024) void haitsma_clip_prep_kernel(const uint64_t* __restrict__ offsets, size_t n_clips, uint32_t sr,
                                                                 uint64_t* __restrict__ s5_off, uint64_t* __restrict__ fr_off,
                                                                 uint64_t* __restrict__ out_off) {
    __shared__ uint64_t part[2][16];
    __shared__ uint64_t carry[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry[0] = carry[1] = 0;
    __syncthreads();
    for (size_t base = 0; base < n_clips; base += 1024) {
        const size_t c = base + tid;
        uint64_t n5 = 0, fr = 0;
        if (c < n_clips) {
            const uint64_t len = offsets[c + 1] - offsets[c];
            n5 = sr == (uint32_t)kHkSr ? len : (uint64_t)(((unsigned __int128)len * kHkSr) / sr);
            fr = n5 >= (uint64_t)kHkN ? 1 + (n5 - kHkN) / kHkHop : 0;
        }
        uint64_t a = n5, b = fr;          // inclusive scans inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t oa = __shfl_up(a, d, 64), ob = __shfl_up(b, d, 64);
            if (lane >= d) a += oa, b += ob;
        }
        if (lane == 63) part[0][wave] = a, part[1][wave] = b;
        __syncthreads();
        uint64_t wa = 0, wb = 0;
        for (int w = 0; w < wave; w++) wa += part[0][w], wb += part[1][w];
        const uint64_t ca = carry[0], cb = carry[1];
        if (c < n_clips) {
            s5_off[c] = ca + wa + a - n5;
            fr_off[c] = cb + wb + b - fr;
            if (out_off) out_off[c] = cb + wb + b - fr;
        }
        __syncthreads();
        if (tid == 1023) carry[0] = ca + wa + a, carry[1] = cb + wb + b;
        __syncthreads();
    }
    if (tid == 0) {
        s5_off[n_clips] = carry[0];
        fr_off[n_clips] = carry[1];
        if (out_off) out_off[n_clips] = carry[1];
    }
}

__device__ __forceinline__ size_t upper_clip(const uint64_t* __restrict__ off, size_t n_clips, uint64_t g) {
    size_t lo = 0, hi = n_clips;          // largest c with off[c] <= g (off[0] = 0)
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (off[mid] <= g) lo = mid;
        else hi = mid;
    }
    return lo;
}

// A1 for every clip of the batch at once: output sample g of the concatenated 5 kHz streams.
__global__ void haitsma_resample_batch_kernel(const float* __restrict__ in, const uint64_t* __restrict__ offsets,
                                              const uint64_t* __restrict__ s5_off, size_t n_clips, uint32_t sr,
                                              float* __restrict__ out, size_t m_ub) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= m_ub || g >= s5_off[n_clips]) return;
    const size_t c = upper_clip(s5_off, n_clips, g);
    const uint64_t i = g - s5_off[c], n = offsets[c + 1] - offsets[c];
    const float* src = in + offsets[c];
    const uint64_t num = i * sr;
    const size_t idx = (size_t)(num / kHkSr);
    const uint32_t rem = (uint32_t)(num % kHkSr);
    const float frac = (float)((double)rem / (double)kHkSr);
    const float x0 = src[idx], x1 = src[idx + 1 < n ? idx + 1 : n - 1];
    const float d = x1 - x0;
    const float mm = d * frac;
    out[g] = x0 + mm;
}

// Frame g of the batch -> where its 2048 samples start (in the 5 kHz stream, or in the caller's own when sr = 5000).
__global__ void haitsma_frame_map_kernel(const uint64_t* __restrict__ offsets, const uint64_t* __restrict__ s5_off,
                                         const uint64_t* __restrict__ fr_off, size_t n_clips, bool own_stream,
                                         uint64_t* __restrict__ src_map, size_t frames_ub) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= frames_ub) return;
    if (g >= fr_off[n_clips]) {
        src_map[g] = kFrameNone;
        return;
    }
    const size_t c = upper_clip(fr_off, n_clips, g);
    const uint64_t f = g - fr_off[c];
    src_map[g] = ((own_stream ? offsets[c] : s5_off[c]) + f * kHkHop) | (f == 0 ? kFrameFirst : 0);
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

WangWs wang_ws_layout(size_t n_src_total, size_t n_clips, uint32_t sr_in, uint32_t pps) {
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    WangWs w;
    // upper bounds from what the host knows (the per-clip lengths live on the device): sum of floors <= floor of sum
    const size_t n8k = sr_in == (uint32_t)kWangSr ? n_src_total
                                                   : (size_t)(((unsigned __int128)n_src_total * kWangSr) / sr_in);
    w.frames = n8k / kWangHop + 1;                                   // frames_c < n8k_c / hop
    w.seg = wang_segment(w.frames);
    w.n_seg = (uint32_t)(w.frames / w.seg + n_clips + 1);
    w.n_sec = (uint32_t)((w.frames * kWangHop) / kWangSr + n_clips + 1);
    w.n_clips = (uint32_t)n_clips;
    size_t off = 0;
    w.clips = off;    off = align(off + n_clips * sizeof(WangClip));
    w.seg_cnt = off;  off = align(off + (n_clips + 1) * 4);
    w.seg_base = off; off = align(off + (n_clips + 1) * 4);
    w.sec_cnt = off;  off = align(off + (n_clips + 1) * 4);
    w.sec_base = off; off = align(off + (n_clips + 1) * 4);
    w.seg_clip = off; off = align(off + (size_t)w.n_seg * 4);
    w.sec_clip = off; off = align(off + (size_t)w.n_sec * 4);
    w.out_off = off;  off = align(off + (n_clips + 1) * 8);
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
    w.pc = off;       off = align(off + maxp * 4);
    w.pair_cnt = off; off = align(off + (maxp + 1) * 4);
    w.pair_off = off; off = align(off + (maxp + 1) * 4);
    const size_t scan_n = maxp > n_clips ? maxp : n_clips;
    w.scan_tmp = off; off = align(off + 2 * (scan_n / 4096 + 4) * 4);
    w.total = off + 256;
    return w;
}

// pcm: the batch buffer; d_offsets: n_clips + 1 device offsets into it (nullptr: one clip [0, n_src_total)); sr_in: the
// clips' sample rate (8000: taken as is; anything else: resampled to 8 kHz inside the stream kernel, A1);
// d_out_off: n_clips + 1 hash offsets (may be nullptr); out_count: total hashes produced (may be nullptr)
int launch_wang_batch(const float* pcm, const uint64_t* d_offsets, size_t n_src_total, size_t n_clips, uint32_t sr_in,
                      uint32_t fan_out, uint32_t zone_t, uint32_t zone_f, uint32_t pps,
                      float floor_power, uint8_t* ws, const WangWs& w, uint32_t* out, size_t cap, uint64_t* d_out_off,
                      uint64_t* out_count, hipStream_t stream) {
    if (n_clips == 0 || pps == 0) {
        if (out_count) (void)hipMemsetAsync(out_count, 0, 8, stream);
        if (d_out_off) (void)hipMemsetAsync(d_out_off, 0, (n_clips + 1) * 8, stream);
        return 0;
    }
    auto f32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    auto u32 = [&](size_t off) { return reinterpret_cast<uint32_t*>(ws + off); };
    WangClip* clips = reinterpret_cast<WangClip*>(ws + w.clips);
    const uint32_t nc = (uint32_t)n_clips;
    hipLaunchKernelGGL(wang_clip_prep_kernel, dim3(blocks_for(n_clips, 256)), dim3(256), 0, stream, d_offsets,
                       (uint64_t)n_src_total, nc, sr_in, w.seg, clips, u32(w.seg_cnt), u32(w.sec_cnt));
    launch_exclusive_scan(u32(w.seg_cnt), n_clips, u32(w.seg_base), u32(w.scan_tmp), stream);
    launch_exclusive_scan(u32(w.sec_cnt), n_clips, u32(w.sec_base), u32(w.scan_tmp), stream);
    const size_t map_n = w.n_seg > w.n_sec ? w.n_seg : w.n_sec;
    hipLaunchKernelGGL(wang_clip_map_kernel, dim3(blocks_for(map_n, 256)), dim3(256), 0, stream,
                       (const uint32_t*)u32(w.seg_base), (const uint32_t*)u32(w.sec_base), nc, u32(w.seg_clip),
                       u32(w.sec_clip));
    const uint32_t* n_segs_total = u32(w.seg_base) + n_clips;
    const uint32_t* n_sec_total = u32(w.sec_base) + n_clips;
    (void)hipMemsetAsync(u32(w.cand_cnt), 0, (size_t)w.n_sec * 4, stream);
    const size_t lds = sizeof(WangStreamLds);
    if (sr_in == (uint32_t)kWangSr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wang_stream_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(wang_stream_kernel<false>, dim3(w.n_seg), dim3(kSW * 64), lds, stream, pcm,
                           (const WangClip*)clips, (const uint32_t*)u32(w.seg_clip), (const uint32_t*)u32(w.seg_base),
                           (const uint32_t*)u32(w.sec_base), n_segs_total, w.seg, sr_in, u32(w.cand_cnt),
                           u32(w.cand_t), u32(w.cand_k), f32(w.cand_p));
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wang_stream_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(wang_stream_kernel<true>, dim3(w.n_seg), dim3(kSW * 64), lds, stream, pcm,
                           (const WangClip*)clips, (const uint32_t*)u32(w.seg_clip), (const uint32_t*)u32(w.seg_base),
                           (const uint32_t*)u32(w.sec_base), n_segs_total, w.seg, sr_in, u32(w.cand_cnt),
                           u32(w.cand_t), u32(w.cand_k), f32(w.cand_p));
    }
    hipLaunchKernelGGL(wang_select_kernel, dim3(w.n_sec), dim3(64), 0, stream, u32(w.cand_cnt), u32(w.cand_t),
                       u32(w.cand_k), f32(w.cand_p), pps, n_sec_total, u32(w.sel_cnt), u32(w.sel_t), u32(w.sel_k),
                       f32(w.sel_p));
    launch_exclusive_scan(u32(w.sel_cnt), (size_t)w.n_sec, u32(w.sel_off), u32(w.scan_tmp), stream);
    hipLaunchKernelGGL(wang_compact_kernel, dim3(blocks_for((size_t)w.n_sec * pps, 256)), dim3(256), 0, stream,
                       u32(w.sel_cnt), u32(w.sel_off), u32(w.sel_t), u32(w.sel_k), f32(w.sel_p),
                       (const uint32_t*)u32(w.sec_clip), w.n_sec, pps, u32(w.pt), f32(w.pp), u32(w.pc));
    const size_t maxp = (size_t)w.n_sec * pps;
    const uint32_t* np_ptr = u32(w.sel_off) + w.n_sec;  // total peaks
    (void)hipMemsetAsync(u32(w.pair_cnt), 0, (maxp + 1) * 4, stream);
    hipLaunchKernelGGL(wang_pair_kernel<false>, dim3(blocks_for(maxp, 256)), dim3(256), 0, stream,
                       (const uint32_t*)u32(w.pt), (const float*)f32(w.pp), (const uint32_t*)u32(w.pc),
                       (const uint32_t*)u32(w.sec_base), (const uint32_t*)u32(w.sel_off), np_ptr, fan_out, zone_t, zone_f,
                       floor_power, u32(w.pair_cnt), (const uint32_t*)nullptr, (uint2*)nullptr, (size_t)0);
    launch_exclusive_scan(u32(w.pair_cnt), maxp, u32(w.pair_off), u32(w.scan_tmp), stream);
    hipLaunchKernelGGL(wang_pair_kernel<true>, dim3(blocks_for(maxp, 256)), dim3(256), 0, stream,
                       (const uint32_t*)u32(w.pt), (const float*)f32(w.pp), (const uint32_t*)u32(w.pc),
                       (const uint32_t*)u32(w.sec_base), (const uint32_t*)u32(w.sel_off), np_ptr, fan_out, zone_t, zone_f,
                       floor_power, (uint32_t*)nullptr, (const uint32_t*)u32(w.pair_off), reinterpret_cast<uint2*>(out), cap);
    if (out_count)
        hipLaunchKernelGGL(copy_u32_kernel, dim3(1), dim3(1), 0, stream, (const uint32_t*)(u32(w.pair_off) + maxp),
                           out_count);
    if (d_out_off)
        hipLaunchKernelGGL(wang_clip_offsets_kernel, dim3(blocks_for(n_clips + 1, 256)), dim3(256), 0, stream,
                           (const uint32_t*)u32(w.sec_base), (const uint32_t*)u32(w.sel_off),
                           (const uint32_t*)u32(w.pair_off), nc, d_out_off);
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

HaitsmaBatchWs haitsma_batch_ws(size_t n_total, size_t n_clips, uint32_t sr) {
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    HaitsmaBatchWs w;
    w.n5_ub = sr == (uint32_t)kHkSr ? n_total : (size_t)(((unsigned __int128)n_total * kHkSr) / sr) + 1;
    w.frames_ub = w.n5_ub / kHkHop + 1;       // a clip of n5 >= 2048 samples has n5 / 64 - 31 frames
    const size_t chunk = w.frames_ub < kChunkFrames * 4 ? w.frames_ub : kChunkFrames * 4;
    size_t off = 0;
    w.edges = off;   off = align(off + 64 * 4);
    w.s5_off = off;  off = align(off + (n_clips + 1) * 8);
    w.fr_off = off;  off = align(off + (n_clips + 1) * 8);
    w.src_map = off; off = align(off + w.frames_ub * 8);
    w.pcm5k = off;   off = align(off + (sr == (uint32_t)kHkSr ? 0 : (w.n5_ub + 64) * 4));
    w.E = off;       off = align(off + (chunk + 1) * kHkBands * 4);
    w.total = off;
    return w;
}

int launch_haitsma_batch(const float* pcm, const uint64_t* d_offsets, size_t n_total, size_t n_clips, uint32_t sr,
                         const uint32_t* h_edges, uint8_t* ws, const HaitsmaBatchWs& w, uint32_t* out, size_t cap_frames,
                         uint64_t* d_out_offsets, hipStream_t stream) {
    uint32_t* d_edges = reinterpret_cast<uint32_t*>(ws + w.edges);
    uint64_t* s5_off = reinterpret_cast<uint64_t*>(ws + w.s5_off);
    uint64_t* fr_off = reinterpret_cast<uint64_t*>(ws + w.fr_off);
    uint64_t* src_map = reinterpret_cast<uint64_t*>(ws + w.src_map);
    float* E = reinterpret_cast<float*>(ws + w.E);
    (void)hipMemcpyAsync(d_edges, h_edges, (kHkBands + 1) * 4, hipMemcpyHostToDevice, stream);
    hipLaunchKernelGGL(haitsma_clip_prep_kernel, dim3(1), dim3(1024), 0, stream, d_offsets, n_clips, sr, s5_off, fr_off,
                       d_out_offsets);
    if (n_clips == 0 || n_total == 0) return 0;
    const bool own = sr == (uint32_t)kHkSr;
    const float* x = pcm;
    if (!own) {
        float* p5 = reinterpret_cast<float*>(ws + w.pcm5k);
        hipLaunchKernelGGL(haitsma_resample_batch_kernel, dim3(blocks_for(w.n5_ub, 256)), dim3(256), 0, stream, pcm, d_offsets,
                           s5_off, n_clips, sr, p5, w.n5_ub);
        x = p5;
    }
    hipLaunchKernelGGL(haitsma_frame_map_kernel, dim3(blocks_for(w.frames_ub, 256)), dim3(256), 0, stream, d_offsets, s5_off,
                       fr_off, n_clips, own, src_map, w.frames_ub);
    const size_t lds = sizeof(FftLds<kHkN>);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_power_kernel<kHkN, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const size_t chunk = kChunkFrames * 4;
    for (size_t e0 = 0; e0 < w.frames_ub; e0 += chunk) {
        const size_t e1 = e0 + chunk < w.frames_ub ? e0 + chunk : w.frames_ub;
        const size_t w0 = e0 > 0 ? e0 - 1 : 0;  // one frame of history for the time difference
        const size_t wn = e1 - w0;
        unsigned grid = blocks_for(wn, kFftWaves);
        if (grid > 256) grid = 256;
        hipLaunchKernelGGL((stft_power_kernel<kHkN, true>), dim3(grid), dim3(kFftWaves * 64), lds, stream, x, w0, wn, kHkHop,
                           E, (const uint32_t*)d_edges, (float*)nullptr, (const uint64_t*)src_map);
        hipLaunchKernelGGL(haitsma_bits_kernel, dim3(blocks_for(e1 - e0, 256)), dim3(256), 0, stream, E, e0, e1 - e0,
                           out, (const uint64_t*)src_map, cap_frames);
    }
    return 0;
}

}  // namespace ucfp
