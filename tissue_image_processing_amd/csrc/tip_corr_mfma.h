// tip_corr_mfma.h -- the sigma-30 score passes (certified-argmax path) on the matrix cores.
//
// A long 1-D correlation is a banded-Toeplitz matrix product: for 32 consecutive outputs along the filter axis and 32
// lines, Out(32 x 32) = A(32 x K) * B(K x 32) with K = 32 + 2r input positions, A[i][kk] = w(|kk - i - r|) (zero outside
// the band) and B[kk][j] = input position kk of line j.  gfx950's float32-input MFMA (v_mfma_f32_32x32x2_f32) computes
// exactly a k-ordered float32 FMA chain -- one rounding per product, no wider accumulation -- at the float32 vector peak
// rate, but without the VALU loop's operand shuffling: the same certified error class as k_corr_long_fast (partial sums
// of 16 products flushed into a running total: a term sees at most 16 + 1 + 17 + 1 = 35 roundings, the bound
// k_argmax_certify uses), at roughly twice the sustained rate of the VALU kernel.
//
//   tile: TO = 256 outputs x 32 lines (+ 2r halo positions) in LDS = 62 KB -> two blocks per CU, so one block's staging
//   overlaps the other's MFMA loop.  A wave owns 32 outputs x 32 lines = one accumulator tile; per K-step (2 positions)
//   it reads one weight (A: lane i = l & 31, k = l >> 5) and one sample (B: lane j = l & 31, k = l >> 5) from LDS.
//   y pass (AXIS 1): LDS image [position][line]; results go straight to global memory (a lane's column is a line:
//   128-byte row segments).  x pass (AXIS 2): LDS image [line][position] with an odd pitch; results are transposed
//   through LDS so that rows are stored contiguously.
#pragma once
#include "tip_corr.h"

namespace tip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MF_TO = 256, MF_LN = 32, MF_NW = 8, MF_SEG = 8;   // MF_SEG K-steps (16 products) per partial sum

constexpr int MF_PRE = 32;   // tile floats per thread: (256 + 2 * 127 max) * 32 lines / 512 threads, rounded up

template <int AXIS>
__global__ void __launch_bounds__(MF_NW * 64, 4) k_corr_long_mfma(const float *__restrict__ in, float *__restrict__ out, int Z, int Y, int X,
                                                            TapsF taps, int ntiles, int tiles_pos, int tiles_ln)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ float wfull[2 * 127 + 64];
    const int r = taps.n >> 1;
    const int npos = MF_TO + 2 * r;
    const int pitch = AXIS == 1 ? MF_LN : npos + 1 + (npos & 1);      // AXIS 2: odd row pitch (lanes walk down the lines)
    // (fp32 MFMA shares the SIMD's vector ALUs on gfx950 -- see tip_corr_mfma2.h: wave index in a scalar register, packed
    //  flush adds.  Scalar row bases for the 31 prefetch loads were tried: 119 spilled SGPRs, 0.72 ms instead of 0.67)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int len = AXIS == 1 ? Y : X;
    // zero-padded symmetric kernel: wfull[d + r + 31] = w(|d|) for |d| <= r, else 0
    for (int j = threadIdx.x; j < 2 * r + 63; j += MF_NW * 64) {
        const int d = j - (r + 31), ad = d < 0 ? -d : d;
        wfull[j] = ad <= r ? taps.w[r - ad] : 0.f;
    }
    // The tile of step t+1 is fetched into registers while the MFMA loop runs on the tile of step t (a persistent block
    // walks tiles blockIdx.x, + gridDim.x, ...): with two blocks per CU the matrix pipe only waits for the LDS fill.
    // Thread -> tile element map: AXIS 1 element e = u * 512 + tid is row e / 32, line e % 32 (128-byte row segments);
    // AXIS 2 element e is line e / npos', position e % npos' with npos' = 512 (positions >= npos unused).
    float pre[MF_PRE];
    auto fetch = [&](int t) {
        const int lx = t % tiles_ln, py = (t / tiles_ln) % tiles_pos, z = t / (tiles_ln * tiles_pos);
        const float *src = in + (long)z * Y * X;
        const int p0 = py * MF_TO, l0 = lx * MF_LN;
        if (AXIS == 1) {
            const int col = threadIdx.x & 31, rb = threadIdx.x >> 5;
            const int xx = min(l0 + col, X - 1);
#pragma unroll
            for (int u = 0; u < MF_PRE; ++u) {
                const int row = rb + 16 * u;
                pre[u] = row < npos ? src[(long)clampi(p0 - r + row, 0, Y - 1) * X + xx] : 0.f;
            }
        } else {
#pragma unroll
            for (int u = 0; u < MF_PRE; ++u) {
                const int e = u * (MF_NW * 64) + threadIdx.x, l = e >> 9, pos = e & 511;
                const int yy = min(l0 + l, Y - 1);
                pre[u] = pos < npos ? src[(long)yy * X + clampi(p0 - r + pos, 0, X - 1)] : 0.f;
            }
        }
    };
    auto fill = [&]() {
        if (AXIS == 1) {
            const int col = threadIdx.x & 31, rb = threadIdx.x >> 5;
#pragma unroll
            for (int u = 0; u < MF_PRE; ++u)
                if (rb + 16 * u < npos) tile[(rb + 16 * u) * MF_LN + col] = pre[u];
        } else {
#pragma unroll
            for (int u = 0; u < MF_PRE; ++u) {
                const int e = u * (MF_NW * 64) + threadIdx.x, l = e >> 9, pos = e & 511;
                if (pos < npos) tile[l * pitch + pos] = pre[u];
            }
        }
    };
    const int o0 = wave * 32;                      // this wave's 32 outputs along the filter axis
    const int i = lane & 31, k = lane >> 5;
    const int steps = 16 + r;                      // K = 32 + 2r positions, two per MFMA
    const float *wp = wfull + (k - i + 31);        // + 2s
    const float *bp = AXIS == 1 ? tile + (o0 + k) * MF_LN + i : tile + i * pitch + o0 + k;   // + 2s positions
    const int bstep = AXIS == 1 ? 2 * MF_LN : 2;
    int t = blockIdx.x;
    if (t < ntiles) fetch(t);
    for (; t < ntiles; t += gridDim.x) {
        const int lx = t % tiles_ln, py = (t / tiles_ln) % tiles_pos, z = t / (tiles_ln * tiles_pos);
        float *dst = out + (long)z * Y * X;
        const int p0 = py * MF_TO, l0 = lx * MF_LN;
        __syncthreads();                            // the previous step is done with the LDS image
        fill();
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
        f32x16 zero, tot;
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        f32x2_ tot2[8];
#pragma unroll
        for (int q = 0; q < 16; ++q) zero[q] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) tot2[q] = f32x2_{0.f, 0.f};
        const bool active = p0 + o0 < len;          // (wave-uniform: a whole 32-output group beyond the axis end does nothing)
        if (active) {
            // groups of MF_SEG K-steps (steps % MF_SEG == 0: checked by the launcher); the other waves of the SIMD cover
            // the LDS latency of a group's operand reads (prefetching the next group in registers spills at 128 VGPRs).
            // A group's chain starts from a zero C operand and is flushed into the running total with packed adds.
            for (int s0 = 0; s0 < steps; s0 += MF_SEG) {
                float a[MF_SEG], b[MF_SEG];
#pragma unroll
                for (int u = 0; u < MF_SEG; ++u) { a[u] = wp[2 * (s0 + u)]; b[u] = bp[(s0 + u) * bstep]; }
                f32x16 acc = zero;
#pragma unroll
                for (int u = 0; u < MF_SEG; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) tot2[q] += f32x2_{acc[2 * q], acc[2 * q + 1]};     // short partial sums: the certified bound
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) tot[q] = tot2[q >> 1][q & 1];
        // D layout: lane l holds column j = l & 31, rows (q & 3) + 8 * (q >> 2) + 4 * (l >> 5)
        if (AXIS == 1) {
            const int xx = l0 + i;
            if (xx < X && active) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int yy = p0 + o0 + (q & 3) + 8 * (q >> 2) + 4 * k;
                    if (yy < Y) dst[(long)yy * X + xx] = tot[q];
                }
            }
        } else {
            constexpr int OP = MF_TO + 1;          // odd pitch of the transposed result image
            __syncthreads();                        // every wave has finished reading the input tile
            if (active) {
#pragma unroll
                for (int q = 0; q < 16; ++q) tile[i * OP + o0 + (q & 3) + 8 * (q >> 2) + 4 * k] = tot[q];
            }
            __syncthreads();
            for (int l = wave; l < MF_LN; l += MF_NW) {
                const int yy = l0 + l;
                if (yy >= Y) break;
                for (int p = lane; p < MF_TO; p += 64)
                    if (p0 + p < X) dst[(long)yy * X + p0 + p] = tile[l * OP + p];
            }
        }
    }
}

}  // namespace tip
