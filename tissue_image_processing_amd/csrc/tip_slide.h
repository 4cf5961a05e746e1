// tip_slide.h -- register-sliding short-kernel passes of the projection (same exact scipy arithmetic as tip_corr.h,
// every input loaded once per thread instead of once per tap; HBM-bound instead of instruction-bound).
#pragma once
#include "tip_corr.h"

namespace tip {

// scipy-order symmetric tap sum over a compile-time window: win[0..2R], centre at R
template <int R>
__device__ __forceinline__ double tap_sum(const double (&win)[2 * R + 1], const Taps &taps)
{
    double tmp = win[R] * taps.w[R];
#pragma unroll
    for (int d = R; d >= 1; --d) tmp += (win[R - d] + win[R + d]) * taps.w[R - d];
    return tmp;
}

// ---- z pass, radius 2 (sigma 0.5), four x columns per thread ------------------------------------------------------------
struct Src4U16Clip {       // uint16 stack reader with the airyscan offset and the percentile clip fused (sp.py:26-36)
    const uint16_t *p;
    int airy;
    const float *clip_p95;
    const int *clip_has;
    __device__ __forceinline__ void load(long idx, float (&v)[4]) const
    {
        const ushort4 u = *reinterpret_cast<const ushort4 *>(p + idx);
        v[0] = (float)u.x; v[1] = (float)u.y; v[2] = (float)u.z; v[3] = (float)u.w;
        const bool has = *clip_has != 0;
        const float c = *clip_p95;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (airy) { v[k] -= 10000.f; if (v[k] < 0.f) v[k] = 0.f; }
            if (has && v[k] > c) v[k] = c;
        }
    }
};
struct Src4F32 {
    const float *p;
    __device__ __forceinline__ void load(long idx, float (&v)[4]) const
    {
        const float4 f = *reinterpret_cast<const float4 *>(p + idx);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    }
};

// requires X % 4 == 0.  One thread: 4 adjacent x, all z (window of 5 planes in registers, 'nearest' at both ends).
template <typename Src>
__global__ void __launch_bounds__(256) k_zpass_r2_x4(Src src, float *__restrict__ out, int Z, long P, Taps taps)
{
    const long q = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (q >= P) return;
    double win[5][4];
    float v[4];
    src.load(q, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) win[0][k] = win[1][k] = win[2][k] = (double)v[k];
    src.load((long)min(1, Z - 1) * P + q, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) win[3][k] = (double)v[k];
    src.load((long)min(2, Z - 1) * P + q, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) win[4][k] = (double)v[k];
    for (int z = 0; z < Z; ++z) {
        float4 o;
        float *op = &o.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double tmp = win[2][k] * taps.w[2];
            tmp += (win[0][k] + win[4][k]) * taps.w[0];
            tmp += (win[1][k] + win[3][k]) * taps.w[1];
            op[k] = (float)tmp;
        }
        *reinterpret_cast<float4 *>(out + (long)z * P + q) = o;
        src.load((long)min(z + 3, Z - 1) * P + q, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            win[0][k] = win[1][k]; win[1][k] = win[2][k]; win[2][k] = win[3][k]; win[3][k] = win[4][k];
            win[4][k] = (double)v[k];
        }
    }
}

// ---- y pass: one thread per x, slides down a segment of SEG*(2R+1) outputs with a rotating register window ---------------
template <typename T, int R, int SEG>
__global__ void __launch_bounds__(256) k_ypass_slide(const T *__restrict__ in, T *__restrict__ out, int Y, int X, Taps taps)
{
    constexpr int W = 2 * R + 1;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= X) return;
    const int y0 = blockIdx.y * (SEG * W);
    const T *src = in + (long)blockIdx.z * Y * X + x;
    T *dst = out + (long)blockIdx.z * Y * X + x;
    double win[W];
#pragma unroll
    for (int i = 0; i < W; ++i) win[i] = (double)src[(long)clampi(y0 - R + i, 0, Y - 1) * X];
    for (int s = 0; s < SEG; ++s) {
#pragma unroll
        for (int o = 0; o < W; ++o) {
            const int y = y0 + s * W + o;
            // logical window element j lives in physical slot (o + j) % W
            double tmp = win[(o + R) % W] * taps.w[R];
#pragma unroll
            for (int d = R; d >= 1; --d) tmp += (win[(o + R - d) % W] + win[(o + R + d) % W]) * taps.w[R - d];
            if (y < Y) dst[(long)y * X] = (T)tmp;
            win[o % W] = (double)src[(long)clampi(y + R + 1, 0, Y - 1) * X];  // the slot that just left the window
        }
    }
}

// ---- x pass: 8 consecutive outputs per thread from 8 + 2R inputs (aligned float4 loads when possible) --------------------
template <typename T, int R>
__global__ void __launch_bounds__(256) k_xpass_slide(const T *__restrict__ in, T *__restrict__ out, int Y, int X, Taps taps)
{
    constexpr int N = 8 + 2 * R;
    constexpr bool F32 = sizeof(T) == 4;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long row = (long)blockIdx.z * Y + blockIdx.y;
    if (x0 >= X) return;
    const T *src = in + row * X;
    T v[N];
    if (F32 && x0 - R >= 0 && x0 + 8 + R <= X && (X & 3) == 0 && (R & 3) == 0) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const float4 f = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(src) + x0 - R + 4 * i);
            v[4 * i] = (T)f.x; v[4 * i + 1] = (T)f.y; v[4 * i + 2] = (T)f.z; v[4 * i + 3] = (T)f.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = src[clampi(x0 - R + i, 0, X - 1)];
    }
    T o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double tmp = (double)v[k + R] * taps.w[R];
#pragma unroll
        for (int d = R; d >= 1; --d) tmp += ((double)v[k + R - d] + (double)v[k + R + d]) * taps.w[R - d];
        o[k] = (T)tmp;
    }
    T *dst = out + row * X + x0;
    if (F32 && x0 + 8 <= X && (X & 3) == 0) {
        float *df = reinterpret_cast<float *>(dst);
        *reinterpret_cast<float4 *>(df) = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        *reinterpret_cast<float4 *>(df + 4) = make_float4((float)o[4], (float)o[5], (float)o[6], (float)o[7]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (x0 + k < X) dst[k] = o[k];
    }
}

}  // namespace tip
