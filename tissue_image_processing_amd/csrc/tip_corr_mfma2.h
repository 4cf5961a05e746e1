// tip_corr_mfma2.h -- the matrix-core score pass of tip_corr_mfma.h with the staging taken off the critical path.
//
// Same arithmetic (banded-Toeplitz float32 MFMA tiles, partial sums of 16 products), different pipeline: ONE persistent
// block per CU owns two LDS tile buffers; while the eight waves run the MFMA loop on tile t they issue the loads of tile
// t + gridDim.x as asynchronous global -> LDS copies (global_load_lds_dword: no register staging, no LDS-fill phase),
// two copies per group of eight MFMAs, so that VMEM issue and address arithmetic hide under the matrix pipe.  One barrier
// per tile.  (tip_corr_mfma.h's two register-staged blocks per CU run in lock-step -- both fill, then both compute -- and
// leave the matrix pipe idle a third of the time.)
//
//   y pass (AXIS 1): LDS image [position][32 lines]; a copy instruction moves 2 positions x 32 lines (two 128-byte row
//     segments); A = weights, B = samples, a lane's result column is a line -> 128-byte row segments to global memory.
//   x pass (AXIS 2): LDS image [line][pitch], pitch = 2 (mod 64) floats so that the 32 lines x 2 positions of an operand
//     read hit 64 different banks; a copy instruction moves 64 consecutive positions of one line; A = samples,
//     B = weights, so that a lane's result column is an output position -> rows are stored contiguously, no transposition.
#pragma once
#include "tip_corr_mfma.h"

namespace tip {

typedef __attribute__((address_space(3))) float lds_float;

template <int AXIS>
__global__ void __launch_bounds__(MF_NW * 64, 2) k_corr_long_mfma2(const float *__restrict__ in, float *__restrict__ out, int Z, int Y, int X,
                                                                   TapsF taps, int ntiles, int tiles_pos, int tiles_ln, int pitch,
                                                                   int bufsz)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [2][bufsz] tile buffers, then the padded kernel
    float *wfull = lds + 2 * bufsz;
    const int r = taps.n >> 1;
    const int npos = MF_TO + 2 * r;                                   // (r % 8 == 0: npos is a multiple of 16)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int len = AXIS == 1 ? Y : X;
    const long P = (long)Y * X;
    for (int j = threadIdx.x; j < 2 * r + 63; j += MF_NW * 64) {      // wfull[d + r + 31] = w(|d|), zero outside the band
        const int d = j - (r + 31), ad = d < 0 ? -d : d;
        wfull[j] = ad <= r ? taps.w[r - ad] : 0.f;
    }
    const int i = lane & 31, k = lane >> 5;
    float *sink = wfull + 2 * 127 + 64;                                // 64 floats nobody reads: target of surplus copies
    struct TileAt { const float *src; int p0, l0, fix; };
    auto locate = [&](int t) {
        const int lx = t % tiles_ln, py = (t / tiles_ln) % tiles_pos, z = t / (tiles_ln * tiles_pos);
        TileAt a;
        a.src = in + (long)z * P;
        a.p0 = py * MF_TO;
        a.l0 = lx * MF_LN;
        // the per-tile constant part of a lane's source address: AXIS 1 the line (column), AXIS 2 the clamped position
        a.fix = AXIS == 1 ? min(a.l0 + i, X - 1) : clampi(a.p0 - r + wave * 64 + lane, 0, X - 1);
        return a;
    };
    // Copy instruction u of this wave for the tile at `a` into buffer `buf` ('nearest' edges: clamped source addresses).
    // Branch-free on purpose (the MFMA loop is one basic block): AXIS 1 moves rows 2n, 2n + 1 with n = 8u + wave, a copy
    // beyond the tile lands in the sink; AXIS 2 moves positions 64 * wave + lane of line u -- all 512 positions of the
    // pitch, the ones beyond the tile are never read.
    auto copy = [&](const TileAt &a, float *buf, int u) {
        if (AXIS == 1) {
            const int n = u * MF_NW + wave;
            const int row = clampi(a.p0 - r + 2 * n + k, 0, Y - 1);
            __builtin_amdgcn_global_load_lds(a.src + (long)row * X + a.fix, (lds_float *)(2 * n < npos ? buf + n * 64 : sink), 4, 0, 0);
        } else {
            const int yy = min(a.l0 + u, Y - 1);
            __builtin_amdgcn_global_load_lds(a.src + (long)yy * X + a.fix, (lds_float *)(u < MF_LN ? buf + u * pitch + wave * 64 : sink), 4, 0, 0);
        }
    };
    const int o0 = wave * 32;                                         // this wave's 32 outputs along the filter axis
    const int steps = 16 + r, ngroups = steps / MF_SEG;               // K = 32 + 2r positions, two per MFMA
    const float *wp = wfull + (k - i + 31);                           // + 2s
    const int doff = AXIS == 1 ? (o0 + k) * MF_LN + i : i * pitch + o0 + k;
    const int dstep = AXIS == 1 ? 2 * MF_LN : 2;
    const int nu = AXIS == 1 ? (npos / 2 + MF_NW - 1) / MF_NW : MF_LN;   // copy instructions per wave and tile

    int t = blockIdx.x;
    if (t >= ntiles) return;
    TileAt cur = locate(t);
    for (int u = 0; u < nu; ++u) copy(cur, lds, u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int b = 0;
    for (; t < ntiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        const TileAt nxt = locate(tn < ntiles ? tn : t);            // (last step: the tile is copied once more, unused)
        float *nbuf = lds + (b ^ 1) * bufsz;
        const float *dp = lds + b * bufsz + doff;
        f32x16 zero, tot;
#pragma unroll
        for (int q = 0; q < 16; ++q) { zero[q] = 0.f; tot[q] = 0.f; }
        // One group: MF_SEG MFMAs (a partial sum of 16 products per output, started from zero) with, in their shadow, two
        // copies of the next tile and the LDS reads of the next group's operands.
        float a[MF_SEG], d[MF_SEG];
#pragma unroll
        for (int u = 0; u < MF_SEG; ++u) { a[u] = wp[2 * u]; d[u] = dp[u * dstep]; }
        auto group = [&](int g) -> f32x16 {
            copy(nxt, nbuf, 2 * g);
            copy(nxt, nbuf, 2 * g + 1);
            float na[MF_SEG], nd[MF_SEG];
            const int sn = (g + 1 < ngroups ? g + 1 : g) * MF_SEG;                  // (last group: re-reads itself, unused)
#pragma unroll
            for (int u = 0; u < MF_SEG; ++u) { na[u] = wp[2 * (sn + u)]; nd[u] = dp[(sn + u) * dstep]; }
            f32x16 acc = zero;
#pragma unroll
            for (int u = 0; u < MF_SEG; ++u)
                acc = AXIS == 1 ? __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], d[u], acc, 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_32x32x2f32(d[u], a[u], acc, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < MF_SEG; ++u) { a[u] = na[u]; d[u] = nd[u]; }
            return acc;
        };
        // two accumulators in flight: the flush of one group's partial sums runs under the next group's MFMAs
        f32x16 pa = group(0), pb;
        int g = 1;
        for (; g + 1 < ngroups; g += 2) {
            pb = group(g);
#pragma unroll
            for (int q = 0; q < 16; ++q) tot[q] += pa[q];
            pa = group(g + 1);
#pragma unroll
            for (int q = 0; q < 16; ++q) tot[q] += pb[q];
        }
        if (g < ngroups) {
            pb = group(g);
#pragma unroll
            for (int q = 0; q < 16; ++q) tot[q] += pa[q];
#pragma unroll
            for (int q = 0; q < 16; ++q) tot[q] += pb[q];
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) tot[q] += pa[q];
        }
        for (int u = 2 * ngroups; u < nu; ++u) copy(nxt, nbuf, u);                    // (none at the production radius)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's copies have landed (before its stores join the queue)
        // D layout: lane l holds column l & 31, rows (q & 3) + 8 * (q >> 2) + 4 * (l >> 5)
        float *dst = out + (cur.src - in);
        if (cur.p0 + o0 < len) {                             // (a whole 32-output group beyond the axis end stores nothing)
            if (AXIS == 1) {
                const int xx = cur.l0 + i;
                if (xx < X) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int yy = cur.p0 + o0 + (q & 3) + 8 * (q >> 2) + 4 * k;
                        if (yy < Y) dst[(long)yy * X + xx] = tot[q];
                    }
                }
            } else {
                const int xx = cur.p0 + o0 + i;
                if (xx < X) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int yy = cur.l0 + (q & 3) + 8 * (q >> 2) + 4 * k;
                        if (yy < Y) dst[(long)yy * X + xx] = tot[q];
                    }
                }
            }
        }
        __syncthreads();            // every wave is done with buffer b, every wave's copies into the other one have landed
        b ^= 1;
        cur = nxt;
    }
}

}  // namespace tip
