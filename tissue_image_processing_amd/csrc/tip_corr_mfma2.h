// tip_corr_mfma2.h -- the matrix-core score pass of tip_corr_mfma.h with the staging taken off the critical path.
//
// Same arithmetic (banded-Toeplitz float32 MFMA tiles, partial sums of 16 products), different pipeline: ONE persistent
// block per CU owns two LDS tile buffers; while the eight waves run the MFMA loop on tile t they issue the loads of tile
// t + gridDim.x as asynchronous global -> LDS copies (global_load_lds_dword: no register staging, no LDS-fill phase),
// four copies per group of eight MFMAs in the first half of the tile (so that the last of them has half a tile to land).
// One barrier per tile.  The float32 MFMA shares the SIMD's vector ALUs on gfx950 (see the kernel body), so the loop is
// written for as few vector instructions as possible.
//
//   y pass (AXIS 1): LDS image [position][32 lines]; a copy instruction moves 2 positions x 32 lines (two 128-byte row
//     segments); A = weights, B = samples, a lane's result column is a line -> 128-byte row segments to global memory.
//   x pass (AXIS 2): LDS image [line][pitch] with an odd pitch: an operand read is served per 32-lane half (the 32 lines of
//     one position) on 32 banks, so consecutive lines must land on consecutive banks; a copy instruction moves 64
//     consecutive positions of one line; A = samples, B = weights, so that a lane's result column is an output position ->
//     rows are stored contiguously, no transposition.
#pragma once
#include "tip_corr_mfma.h"

namespace tip {

typedef __attribute__((address_space(3))) float lds_float;
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int MF_CPG = 4;   // copy instructions per MFMA group (32 per wave and tile at the production radius: groups 0..7 of 17)

template <int AXIS>
__global__ void __launch_bounds__(MF_NW * 64, 2) k_corr_long_mfma2(const float *__restrict__ in, float *__restrict__ out, int Z, int Y, int X,
                                                                   TapsF taps, int ntiles, int tiles_pos, int tiles_ln, int pitch,
                                                                   int bufsz)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [2][bufsz] tile buffers, then the padded kernel
    float *wfull = lds + 2 * bufsz;
    const int r = taps.n >> 1;
    const int npos = MF_TO + 2 * r;                                   // (r % 8 == 0: npos is a multiple of 16)
    // On gfx950 the float32 MFMA runs on the SIMD's vector ALUs: a v_fma / v_add issued by ANY wave of the SIMD delays the
    // matrix work by its own 4 cycles (tools/ubench/mfma_rate.hip: 8.66 ms of MFMA + 3.10 ms of v_fma on partner waves =
    // 11.74 ms together).  So everything around the MFMAs is kept off the vector ALU: the wave index and the tile
    // coordinates live in scalar registers, the copies use a scalar base + a per-lane offset fixed per tile, the partial
    // sums are flushed with packed adds.
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int len = AXIS == 1 ? Y : X;
    const long P = (long)Y * X;
    for (int j = threadIdx.x; j < 2 * r + 63; j += MF_NW * 64) {      // wfull[d + r + 31] = w(|d|), zero outside the band
        const int d = j - (r + 31), ad = d < 0 ? -d : d;
        wfull[j] = ad <= r ? taps.w[r - ad] : 0.f;
    }
    const int i = lane & 31, k = lane >> 5;
    float *sink = wfull + 2 * 127 + 64;                                // 64 floats nobody reads: target of surplus copies
    // tile coordinates (line block lx, position block py, plane z), advanced by gridDim.x tiles per step with carries
    // instead of divisions (a division of a uniform value is ~30 vector instructions)
    struct TileAt { int lx, py, z; };
    const int G = gridDim.x;
    const int dlx = G % tiles_ln, dpy = (G / tiles_ln) % tiles_pos, dz = G / (tiles_ln * tiles_pos);
    auto advance = [&](TileAt a) {
        a.lx += dlx;
        int c = a.lx >= tiles_ln ? 1 : 0;
        a.lx -= c * tiles_ln;
        a.py += dpy + c;
        c = a.py >= tiles_pos ? 1 : 0;
        a.py -= c * tiles_pos;
        a.z += dz + c;
        return a;
    };
    // the per-lane part of a copy's source address, fixed for the tile: AXIS 1 the line (column) -- plus one row for the
    // upper half-wave --, AXIS 2 the clamped position
    auto lane_fix = [&](const TileAt &a) -> unsigned {
        return (unsigned)(AXIS == 1 ? min(a.lx * MF_LN + i, X - 1) : clampi(a.py * MF_TO - r + wave * 64 + lane, 0, X - 1));
    };
    // Copy instruction u of this wave for the tile `a` into buffer `buf` ('nearest' edges: clamped source addresses):
    // scalar row base + per-lane offset.  AXIS 1 moves rows 2n, 2n + 1 with n = 8u + wave (a copy beyond the tile lands in
    // the sink); AXIS 2 moves positions 64 * wave + lane of line u -- all 512 positions of the pitch, the ones beyond the
    // tile are never read.
    auto copy = [&](const TileAt &a, unsigned fix, unsigned fix_up, float *buf, int u) {
        const float *plane = in + (long)a.z * P;
        if (AXIS == 1) {
            const int n = u * MF_NW + wave;
            const int base = a.py * MF_TO - r + 2 * n;
            const int r0 = clampi(base, 0, Y - 1), r1 = clampi(base + 1, 0, Y - 1);      // (scalar)
            const float *rowp = plane + (long)r0 * X;
            // (the upper half-wave reads the next row unless both clamp to the same one)
            __builtin_amdgcn_global_load_lds(rowp + (r1 != r0 ? fix_up : fix), (lds_float *)(2 * n < npos ? buf + n * 64 : sink), 4, 0, 0);
        } else {
            const float *rowp = plane + (long)min(a.lx * MF_LN + u, Y - 1) * X;
            __builtin_amdgcn_global_load_lds(rowp + fix, (lds_float *)(u < MF_LN ? buf + u * pitch + wave * 64 : sink), 4, 0, 0);
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
    TileAt cur{t % tiles_ln, (t / tiles_ln) % tiles_pos, t / (tiles_ln * tiles_pos)};
    {
        const unsigned fix = lane_fix(cur), fix_up = fix + (AXIS == 1 ? (unsigned)(k * X) : 0u);
        for (int u = 0; u < nu; ++u) copy(cur, fix, fix_up, lds, u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int b = 0;
    for (; t < ntiles; t += G) {
        const TileAt nxt = t + G < ntiles ? advance(cur) : cur;       // (last step: the tile is copied once more, unused)
        const unsigned nfix = lane_fix(nxt), nfix_up = nfix + (AXIS == 1 ? (unsigned)(k * X) : 0u);
        float *nbuf = lds + (b ^ 1) * bufsz;
        const float *dp = lds + b * bufsz + doff;
        f32x16 zero;
        f32x2 tot[8];
#pragma unroll
        for (int q = 0; q < 16; ++q) zero[q] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) tot[q] = f32x2{0.f, 0.f};
        // One group: MF_SEG MFMAs (a partial sum of 16 products per output, started from zero); the copies of the next tile
        // go out in the FIRST groups (MF_CPG per group: the last of them has half a tile of MFMA time to land) and the LDS
        // reads of the next group's operands are issued a group ahead.
        float a[MF_SEG], d[MF_SEG];
#pragma unroll
        for (int u = 0; u < MF_SEG; ++u) { a[u] = wp[2 * u]; d[u] = dp[u * dstep]; }
        auto group = [&](int g) -> f32x16 {
            if (MF_CPG * g < nu) {                  // (wave-uniform; a copy index beyond the tile goes to the sink)
#pragma unroll
                for (int q = 0; q < MF_CPG; ++q) copy(nxt, nfix, nfix_up, nbuf, MF_CPG * g + q);
            }
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
        auto flush = [&](const f32x16 &p) {       // eight packed adds (v_pk_add_f32) instead of sixteen scalar ones
#pragma unroll
            for (int q = 0; q < 8; ++q) tot[q] += f32x2{p[2 * q], p[2 * q + 1]};
        };
        // two accumulators in flight: the flush of one group's partial sums runs under the next group's MFMAs
        f32x16 pa = group(0), pb;
        int g = 1;
        for (; g + 1 < ngroups; g += 2) {
            pb = group(g);
            flush(pa);
            pa = group(g + 1);
            flush(pb);
        }
        if (g < ngroups) {
            pb = group(g);
            flush(pa);
            flush(pb);
        } else {
            flush(pa);
        }
        for (int u = MF_CPG * ngroups; u < nu; ++u) copy(nxt, nfix, nfix_up, nbuf, u);   // (none at the production radius)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's copies have landed (before its stores join the queue)
        // D layout: lane l holds column l & 31, rows (q & 3) + 8 * (q >> 2) + 4 * (l >> 5)
        float *dst = out + (long)cur.z * P;
        const int p0 = cur.py * MF_TO, l0 = cur.lx * MF_LN;
        if (p0 + o0 < len) {                                 // (a whole 32-output group beyond the axis end stores nothing)
            if (AXIS == 1) {
                const int xx = l0 + i;
                if (xx < X) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int yy = p0 + o0 + (q & 3) + 8 * (q >> 2) + 4 * k;
                        if (yy < Y) dst[(long)yy * X + xx] = tot[q >> 1][q & 1];
                    }
                }
            } else {
                const int xx = p0 + o0 + i;
                if (xx < X) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int yy = l0 + (q & 3) + 8 * (q >> 2) + 4 * k;
                        if (yy < Y) dst[(long)yy * X + xx] = tot[q >> 1][q & 1];
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
