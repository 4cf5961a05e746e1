// tip_corr.h -- exact scipy-order separable correlation kernels (device templates).
//
// Arithmetic contract (scipy/ndimage/src/ni_filters.c NI_Correlate1D, symmetric branch; called through
// gaussian_filter at bim.py:389):
//     tmp  = x[c] * w[r]
//     tmp += (x[c-d] + x[c+d]) * w[r-d]      for d = r, r-1, ..., 1      (all in double, no FMA contraction)
//     out  = (T) tmp
// with mode='nearest' (index clamp).  The whole library is compiled with -ffp-contract=off so that the
// multiply and the add round separately, as scipy's x86-64 builds do.
#pragma once
#include "tip_internal.h"

namespace tip {

template <typename T>
struct LoadPlain {
    const T *p;
    long sz, sy;  // strides of z and y in elements (x stride 1)
    __device__ __forceinline__ double operator()(int z, int y, int x) const { return (double)p[z * sz + y * sy + x]; }
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Generic kernel: one thread per output element, taps read through the cache hierarchy.
// AXIS: 0 = z, 1 = y, 2 = x of a (Z,Y,X) volume.  Good for short kernels (<= ~25 taps).
template <typename T, int AXIS, typename Load>
__global__ void __launch_bounds__(256) k_corr_generic(Load in, T *__restrict__ out, int Z, int Y, int X, Taps taps)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int z = blockIdx.z;
    if (x >= X) return;
    const int r = taps.n >> 1;
    const int len = AXIS == 0 ? Z : (AXIS == 1 ? Y : X);
    const int c = AXIS == 0 ? z : (AXIS == 1 ? y : x);
    auto at = [&](int i) -> double {
        i = clampi(i, 0, len - 1);
        return AXIS == 0 ? in(i, y, x) : (AXIS == 1 ? in(z, i, x) : in(z, y, i));
    };
    double tmp = at(c) * taps.w[r];
    for (int d = r; d >= 1; --d) tmp += (at(c - d) + at(c + d)) * taps.w[r - d];
    out[((long)z * Y + y) * X + x] = (T)tmp;
}

// Stage a (positions x 64 lines) float tile into LDS with 'nearest' clamping.  A block owns its CU's LDS alone, so
// nothing else hides the HBM latency of this phase: every wave keeps 8 (AXIS 1) or 16 (AXIS 2) loads in flight
// before the first LDS write.  first = plane coordinate of tile position 0 along the filter axis.
template <int AXIS, int NW>
__device__ __forceinline__ void stage_tile(float *tile, const float *__restrict__ src, int npos, int first, int l0, int Y, int X,
                                           int lane, int wave)
{
    constexpr int LS = AXIS == 1 ? 64 : 65;
    if (AXIS == 1) {
        constexpr int UL = 8;
        const int xx = min(l0 + lane, X - 1);
        for (int pb = wave; pb < npos; pb += NW * UL) {
            float v[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) v[u] = src[(long)clampi(first + pb + u * NW, 0, Y - 1) * X + xx];
#pragma unroll
            for (int u = 0; u < UL; ++u)
                if (pb + u * NW < npos) tile[(pb + u * NW) * LS + lane] = v[u];
        }
    } else {
        // npos <= 256 + 2*127 < 512: eight 64-wide chunks cover a line; two lines per trip
        for (int l = wave; l < 64; l += 2 * NW) {
            float v[2][8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int yy = min(l0 + l + h * NW, Y - 1);
#pragma unroll
                for (int u = 0; u < 8; ++u) v[h][u] = src[(long)yy * X + clampi(first + lane + 64 * u, 0, X - 1)];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (lane + 64 * u < npos && l + h * NW < 64) tile[(lane + 64 * u) * LS + l + h * NW] = v[h][u];
        }
    }
}

// Long-kernel variant (radius up to 127, float32 volumes): every lane owns one line and slides along
// the filter axis with register-resident left/right windows of R outputs, the line segment (tile +
// 2*radius halo) staged once in LDS as float32.
//   AXIS==1: lanes run along x (coalesced loads/stores), positions along y: LDS[pos][64]
//   AXIS==2: lanes run along y, positions along x: loaded coalesced along x and written transposed
//            into LDS[pos][65] (stride 65 keeps both the transposed write and the lane-major read
//            conflict-free)
// Block = 256 threads = 4 waves; the 4 waves split the tile's TO outputs per line.
template <int AXIS, int TO, int R>
__global__ void __launch_bounds__(256) k_corr_long_f32(const float *__restrict__ in, float *__restrict__ out,
                                                       int Z, int Y, int X, Taps taps)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];
    constexpr int LS = AXIS == 1 ? 64 : 65;
    const int r = taps.n >> 1;
    const int npos = TO + 2 * r;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int z = blockIdx.z;
    const int len = AXIS == 1 ? Y : X;      // filter-axis length
    const int nlines = AXIS == 1 ? X : Y;   // line-axis length
    const int p0 = blockIdx.y * TO;         // first output position of this tile
    const int l0 = blockIdx.x * 64;         // first line of this tile
    const float *src = in + (long)z * Y * X;
    float *dst = out + (long)z * Y * X;

    stage_tile<AXIS, 4>(tile, src, npos, p0 - r, l0, Y, X, lane, wave);
    __syncthreads();

    const int line = l0 + lane;
    constexpr int PER_WAVE = TO / 4;
    for (int g = 0; g < PER_WAVE / R; ++g) {
        const int o0 = wave * PER_WAVE + g * R;  // first output (tile-relative) of this group
        if (p0 + o0 >= len) break;               // wave-uniform
        const float *ctr = tile + (r + o0) * LS + lane;
        double acc[R], L[R], Rr[R];
        const double wc = taps.w[r];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            acc[i] = (double)ctr[i * LS] * wc;
            L[i] = (double)ctr[(i - r) * LS];
            Rr[i] = (double)ctr[(i + r) * LS];
        }
#pragma unroll 8
        for (int d = r; d >= 1; --d) {
            const double w = taps.w[r - d];
#pragma unroll
            for (int i = 0; i < R; ++i) acc[i] += (L[i] + Rr[i]) * w;
            // slide: left window moves right by one, right window moves left by one
#pragma unroll
            for (int i = 0; i < R - 1; ++i) L[i] = L[i + 1];
            L[R - 1] = (double)ctr[(R - 1 - (d - 1)) * LS];
#pragma unroll
            for (int i = R - 1; i > 0; --i) Rr[i] = Rr[i - 1];
            Rr[0] = (double)ctr[(d - 1) * LS];
        }
        if (line < nlines) {
            if (AXIS == 1) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int yy = p0 + o0 + i;
                    if (yy < Y) dst[(long)yy * X + line] = (float)acc[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int xx = p0 + o0 + i;
                    if (xx < X) dst[(long)line * X + xx] = (float)acc[i];
                }
            }
        }
    }
}

// ---- fast (NOT bit-exact) long pass for the certified argmax: float32 math, FMA allowed -----------------------------------
// Same tiling as k_corr_long_f32; every lane slides R = 2H outputs.
//   VAR 1 (default): outputs j and j+H share one register pair,
//       acc[j] = pk_fma(L[j] + Rr[j], {w, w}, acc[j]),   L[j] = {x[j-d], x[j+H-d]},  Rr[j] = {x[j+d], x[j+H+d]}
//     = one v_pk_add_f32 + one v_pk_fma_f32 per TWO outputs and tap.  Pairing outputs H apart (not neighbours) makes the
//     window slide a pure renaming of whole pairs (L[j] <- L[j+1]); the one new pair per side and tap is read from LDS
//     through a volatile pointer -- otherwise the compiler notices that half of it is already in a register and assembles
//     the pair with v_mov, which costs as much as the arithmetic it saves.  Needs radius % H == 0.
//   VAR 0: scalar v_add_f32 + v_fmac_f32 per output and tap (any radius).  VOP2 float32 on VGPR operands issues about
//     twice as fast as the packed instructions (tools/ubench/valu_rate.hip: 109 vs 104 TFLOP/s for this add+fma mix), so
//     the two variants run within a few percent of each other.
// Error vs the exact pass: every term is non-negative.  The taps are summed in partial sums of at most FAST_SEG taps
// that are flushed into a running total, so a term sees at most FAST_SEG + 1 FMA roundings (its partial sum, the centre
// term included), one tap rounding, one pair-sum rounding and at most ceil(r / 8) + 1 roundings of the running total:
//     |fast - exact| <= ((1+u)^(FAST_SEG + 3 + ceil(r/8) + 1) - 1) * exact,   u = 2^-24
// i.e. 35 u for r = 120 and 36 u at the largest radius (a single running sum would be 123 u); see k_argmax_certify for
// how the bound is used.
constexpr int FAST_SEG = 16;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int AXIS, int TO, int NW, int R, int VAR, int REM = 0>
__global__ void __launch_bounds__(NW * 64) k_corr_long_fast(const float *__restrict__ in, float *__restrict__ out, int Z, int Y, int X,
                                                        TapsF taps)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];
    constexpr int LS = AXIS == 1 ? 64 : 65;
    const int r = taps.n >> 1;
    const int npos = TO + 2 * r;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int z = blockIdx.z;
    const int len = AXIS == 1 ? Y : X;
    const int nlines = AXIS == 1 ? X : Y;
    const int p0 = blockIdx.y * TO;
    const int l0 = blockIdx.x * 64;
    const float *src = in + (long)z * Y * X;
    float *dst = out + (long)z * Y * X;
    // taps live in LDS: scalar loads in the tap loop would share lgkmcnt with the window reads and force a full
    // drain (s_waitcnt lgkmcnt(0)) at every use; LDS broadcasts return in order and pipeline with them
    __shared__ __attribute__((aligned(16))) float wl[256];
    for (int i = threadIdx.x; i < 256; i += NW * 64) wl[i] = taps.w[i];
    stage_tile<AXIS, NW>(tile, src, npos, p0 - r, l0, Y, X, lane, wave);
    __syncthreads();
    const int line = l0 + lane;
    constexpr int PER_WAVE = TO / NW;
    static_assert(PER_WAVE % R == 0, "outputs per wave must be a multiple of the group size");
    constexpr bool STAGED = AXIS == 2 && PER_WAVE == R;  // x pass: results go back through LDS for coalesced row stores
    for (int g = 0; g < PER_WAVE / R; ++g) {
        const int o0 = wave * PER_WAVE + g * R;
        const bool active = p0 + o0 < len;  // wave-uniform
        if (!STAGED && !active) break;
        const float *ctr = tile + (r + o0) * LS + lane;
        typedef __attribute__((address_space(3))) const volatile float *lds_cvf32;   // (a plain volatile pointer decays to flat loads)
        constexpr int H = R / 2;
        float res[R];
        if (active) {
            const float wc = wl[r];
            if constexpr (VAR == 0) {   // scalar: one v_add_f32 + one v_fmac_f32 per output and tap
                float acc[R], L[R], Rr[R];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    acc[i] = ctr[i * LS] * wc;
                    L[i] = ctr[(i - r) * LS];
                    Rr[i] = ctr[(i + r) * LS];
                }
                float tot[R];
#pragma unroll
                for (int i = 0; i < R; ++i) tot[i] = 0.f;
                int seg = 0;
#pragma unroll R
                for (int d = r; d >= 1; --d) {
                    const float w = wl[r - d];
#pragma unroll
                    for (int i = 0; i < R; ++i) acc[i] = __builtin_fmaf(L[i] + Rr[i], w, acc[i]);
#pragma unroll
                    for (int i = 0; i < R - 1; ++i) L[i] = L[i + 1];
                    L[R - 1] = ctr[(R - d) * LS];
#pragma unroll
                    for (int i = R - 1; i > 0; --i) Rr[i] = Rr[i - 1];
                    Rr[0] = ctr[(d - 1) * LS];
                    if (++seg == FAST_SEG) {   // short partial sums: see the error bound above k_argmax_certify
                        seg = 0;
#pragma unroll
                        for (int i = 0; i < R; ++i) { tot[i] += acc[i]; acc[i] = 0.f; }
                    }
                }
#pragma unroll
                for (int i = 0; i < R; ++i) res[i] = tot[i] + acc[i];
            } else {
                static_assert(H <= FAST_SEG, "partial sums must stay within the certified bound");
                f32x2 acc[H], L[H], Rr[H], tot[H];
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    tot[j] = f32x2{0.f, 0.f};
                    acc[j] = f32x2{ctr[j * LS], ctr[(j + H) * LS]} * wc;
                    L[j] = f32x2{ctr[(j - r) * LS], ctr[(j + H - r) * LS]};
                    Rr[j] = f32x2{ctr[(j + r) * LS], ctr[(j + H + r) * LS]};
                }
                // H taps per trip, written as a constant-trip inner loop (the radius must be a multiple of H: checked by
                // the launcher) so that it is fully unrolled and the window rotation is pure renaming
                for (int db = r; db >= H; db -= H) {
#pragma unroll
                    for (int kk = 0; kk < H; ++kk) {
                        const int d = db - kk;
                        const float w = wl[r - d];
#pragma unroll
                        for (int j = 0; j < H; ++j) acc[j] = __builtin_elementwise_fma(L[j] + Rr[j], f32x2{w, w}, acc[j]);
#pragma unroll
                        for (int j = 0; j < H - 1; ++j) L[j] = L[j + 1];
#pragma unroll
                        for (int j = H - 1; j > 0; --j) Rr[j] = Rr[j - 1];
                        // volatile LDS reads: never merged with, or replaced by copies of, words already in registers
                        // (same-run A/B on MI355X: this 0.80 ms per pass, "laundered" non-volatile pointers 0.865 ms,
                        // the scalar variant 0.845 / 0.885 ms)
                        lds_cvf32 q = (lds_cvf32)ctr;
                        L[H - 1] = f32x2{q[(H - d) * LS], q[(2 * H - d) * LS]};
                        Rr[0] = f32x2{q[(d - 1) * LS], q[(H + d - 1) * LS]};
                    }
                    // short partial sums (H <= FAST_SEG taps each): see the error bound above k_argmax_certify
#pragma unroll
                    for (int j = 0; j < H; ++j) { tot[j] += acc[j]; acc[j] = f32x2{0.f, 0.f}; }
                }
                if constexpr (REM > 0) {   // radius % H == REM: the last REM taps (d = REM .. 1), one more short partial sum
#pragma unroll
                    for (int kk = 0; kk < REM; ++kk) {
                        const int d = REM - kk;
                        const float w = wl[r - d];
#pragma unroll
                        for (int j = 0; j < H; ++j) acc[j] = __builtin_elementwise_fma(L[j] + Rr[j], f32x2{w, w}, acc[j]);
#pragma unroll
                        for (int j = 0; j < H - 1; ++j) L[j] = L[j + 1];
#pragma unroll
                        for (int j = H - 1; j > 0; --j) Rr[j] = Rr[j - 1];
                        lds_cvf32 q = (lds_cvf32)ctr;
                        L[H - 1] = f32x2{q[(H - d) * LS], q[(2 * H - d) * LS]};
                        Rr[0] = f32x2{q[(d - 1) * LS], q[(H + d - 1) * LS]};
                    }
#pragma unroll
                    for (int j = 0; j < H; ++j) tot[j] += acc[j];
                }
#pragma unroll
                for (int j = 0; j < H; ++j) { res[j] = tot[j].x; res[j + H] = tot[j].y; }
            }
        }
        if (STAGED) {
            constexpr int OS = TO + 1;  // odd row stride: lane-major writes and row-major reads are both conflict-free
            __syncthreads();            // every wave has finished reading the input tile
            if (active) {
#pragma unroll
                for (int i = 0; i < R; ++i) tile[lane * OS + o0 + i] = res[i];
            }
            __syncthreads();
            for (int l = wave; l < 64; l += NW) {
                const int yy = l0 + l;
                if (yy >= Y) break;
                for (int p = lane; p < TO; p += 64)
                    if (p0 + p < X) dst[(long)yy * X + p0 + p] = tile[l * OS + p];
            }
        } else if (line < nlines) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int pp = p0 + o0 + i;
                if (AXIS == 1) { if (pp < Y) dst[(long)pp * X + line] = res[i]; }
                else { if (pp < X) dst[(long)line * X + pp] = res[i]; }
            }
        }
    }
}

}  // namespace tip
