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

    if (AXIS == 1) {
        const int xx = min(l0 + lane, X - 1);
        for (int p = wave; p < npos; p += 4) {
            const int yy = clampi(p0 - r + p, 0, Y - 1);
            tile[p * LS + lane] = src[(long)yy * X + xx];
        }
    } else {
        for (int l = wave; l < 64; l += 4) {
            const int yy = min(l0 + l, Y - 1);
            for (int p = lane; p < npos; p += 64) {
                const int xx = clampi(p0 - r + p, 0, X - 1);
                tile[p * LS + l] = src[(long)yy * X + xx];
            }
        }
    }
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

// ---- fast (NOT bit-exact) long pass for the certified argmax: float32 packed math, FMA allowed ---------------------------
// Same tiling as k_corr_long_f32; every lane slides two adjacent 8-output groups packed as float2 so that the
// compiler emits v_pk_add_f32 / v_pk_fma_f32 (2 instructions per tap pair for 2 outputs instead of 6 double ones).
// Error vs the exact pass: every term is non-negative, so |fast - exact| <= ((1+u)^(r+3) - 1) * exact, u = 2^-24
// (tap rounding + pair-sum rounding + at most r+1 FMA roundings); see k_argmax_certify for how the bound is used.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int AXIS, int TO, int NW>
__global__ void __launch_bounds__(NW * 64) k_corr_long_fast(const float *__restrict__ in, float *__restrict__ out, int Z, int Y, int X,
                                                        TapsF taps)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];
    constexpr int LS = AXIS == 1 ? 64 : 65;
    constexpr int R = 8;
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
    if (AXIS == 1) {
        const int xx = min(l0 + lane, X - 1);
        for (int p = wave; p < npos; p += NW) {
            const int yy = clampi(p0 - r + p, 0, Y - 1);
            tile[p * LS + lane] = src[(long)yy * X + xx];
        }
    } else {
        for (int l = wave; l < 64; l += NW) {
            const int yy = min(l0 + l, Y - 1);
            for (int p = lane; p < npos; p += 64) {
                const int xx = clampi(p0 - r + p, 0, X - 1);
                tile[p * LS + l] = src[(long)yy * X + xx];
            }
        }
    }
    __syncthreads();
    const int line = l0 + lane;
    constexpr int PER_WAVE = TO / NW;
    for (int g = 0; g < PER_WAVE / R; ++g) {
        const int o0 = wave * PER_WAVE + g * R;
        if (p0 + o0 >= len) break;
        const float *ctr = tile + (r + o0) * LS + lane;
        // 8 outputs as 4 packed pairs (o, o+1).  Tap d uses "even" windows, tap d-1 "odd" windows (shifted by one
        // position); both shift by a whole pair every two taps, so after 8 taps every register has been renewed and the
        // unrolled loop needs no moves.  r must be even (120 for sigma 30).
        const float wc = taps.w[r];
        f32x2 acc[4], EL[4], OL[4], ER[4], OR[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = f32x2{ctr[(2 * i) * LS], ctr[(2 * i + 1) * LS]} * wc;
            EL[i] = f32x2{ctr[(2 * i - r) * LS], ctr[(2 * i + 1 - r) * LS]};
            OL[i] = f32x2{ctr[(2 * i + 1 - r) * LS], ctr[(2 * i + 2 - r) * LS]};
            ER[i] = f32x2{ctr[(2 * i + r) * LS], ctr[(2 * i + 1 + r) * LS]};
            OR[i] = f32x2{ctr[(2 * i + r - 1) * LS], ctr[(2 * i + r) * LS]};
        }
#pragma unroll 4
        for (int d = r; d >= 2; d -= 2) {
            const float w0 = taps.w[r - d], w1 = taps.w[r - d + 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_elementwise_fma(EL[i] + ER[i], f32x2{w0, w0}, acc[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_elementwise_fma(OL[i] + OR[i], f32x2{w1, w1}, acc[i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) { EL[i] = EL[i + 1]; OL[i] = OL[i + 1]; }
            EL[3] = f32x2{ctr[(8 - d) * LS], ctr[(9 - d) * LS]};
            OL[3] = f32x2{ctr[(9 - d) * LS], ctr[(10 - d) * LS]};
#pragma unroll
            for (int i = 3; i > 0; --i) { ER[i] = ER[i - 1]; OR[i] = OR[i - 1]; }
            ER[0] = f32x2{ctr[(d - 2) * LS], ctr[(d - 1) * LS]};
            OR[0] = f32x2{ctr[(d - 3) * LS], ctr[(d - 2) * LS]};
        }
        if (line < nlines) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pp = p0 + o0 + 2 * i + h;
                    const float val = h == 0 ? acc[i].x : acc[i].y;
                    if (AXIS == 1) { if (pp < Y) dst[(long)pp * X + line] = val; }
                    else { if (pp < X) dst[(long)line * X + pp] = val; }
                }
        }
    }
}

}  // namespace tip
