// tip_corr_f16.h -- the sigma-30 score passes (certified-argmax path) on the 16-bit matrix cores with SPLIT float32 operands.
//
// The passes are arithmetic-bound (241 taps per voxel), and gfx950's float32-input MFMA runs at the float32 VECTOR rate (the round-2/3
// kernels of tip_corr_mfma*.h: 0.67-0.74 ms per pass at 65 % of that 157 TFLOP/s peak).  The fp16 MFMA is sixteen times faster, and the
// score only feeds an argmax whose error bars are certified afterwards (k_argmax_certify), so the same banded-Toeplitz product
//     Out(32 x 32) = W(32 x K) In(K x 32),   K = 32 + 2 r positions,   W[o][kk] = w(|kk - o - r|)
// is formed from fp16 PIECES of power-of-two-scaled values with float32 accumulation (the U-Net's f16x3 arithmetic, tip_unet_conv.h):
//     sample  a s = hi + 2^-11 lo',  hi = fp16(a s),  lo' = fp16((a s - hi) 2^11)      (s = 2^k puts the clip value in [2^13, 2^14))
//     weight  w 2^17 = hi + 2^-11 lo'                                                   (every tap's lo' is a normal fp16 number)
//     a w  ~  hi hi + 2^-11 (hi lo' + lo' hi)              three v_mfma_f32_32x32x16_f16 per 16 positions; the dropped lo' lo' <= 2^-22 a w
// The low pieces are PRE-SCALED by 2^11 and their products collected in their own accumulator, so a sample keeps 22 significant bits
// down to 2^-36 of the clip value (fp16's subnormal floor would otherwise cost the dark fringes of a black-background image their
// relative accuracy, and with it their certification); a positive sample below even that is stored as the smallest positive lo', so
// that "the fast score is exactly zero" still means "every input is exactly zero".
//
// Error bound (what k_argmax_certify relies on), per pass, u = 2^-24, all samples and taps >= 0:
//   split: |a s - hi - 2^-11 lo'| <= 2^-22 a s (+ 2^-35 absolute), the same for the taps, the dropped product 2^-22: <= 12 u of a term;
//   accumulation: a group of two K-steps is six MFMAs started from zero accumulators; one MFMA adds its sixteen products as two
//   exactly-summed halves of eight, each rounded once into the accumulator (measured: tools/ubench/mfma_f16_rounding.hip; a GPU
//   test asserts the probe's outcomes on the device it runs on) -- taken as <= 2 u each --, and the two accumulators of a group (hi hi;
//   hi lo' + lo' hi) are flushed into their own running totals with one float32 add each, the totals joined by one fused multiply-add
//   at the end: a mixed term passes through <= 4 MFMAs x 2 halves x 2 u + 9 flushes + 1 = 26 u (17 K-steps: 9 groups), a hi hi term
//   through 18 u; total <= (1 + u)^26 (1 + 12.1 u) - 1 < 38.2 u relative plus 2^-35 / s absolute; scipy's own pass is within 1.0001 u
//   of the exact sum.  Two passes compose to < 79 u relative (CERT_EPS = 96 u) and < clip 2^-47 absolute (CERT_ABS = clip 2^-44).
//
// Tile: 256 outputs x 32 lines per 256-thread block (4 waves x 64 outputs x 32 lines), halo of r positions either side; the samples are
// split while they are staged ([line][position] fp16 images, row pitch an odd number of 16-byte chunks: conflict-free ds_read_b128),
// 74 KB of LDS -> two blocks per CU, one staging while the other multiplies.  A wave's two 32-output row tiles share the sample
// fragment of a K-step, and the second tile's weight fragment is the first tile's of two steps earlier (Toeplitz): 4 ds_read_b128
// per 6 MFMAs.  The weight fragments of all lanes come out of ONE compact table: fragment (u, i) = w~[8 u - i .. + 7] depends on
// 8 (u - i / 8) - i % 8 only, 37 rows x 8 entries of 16 bytes per piece.
#pragma once
#include "tip_corr.h"
#include "tip_unet_conv.h"      // f16x8, f32x16

namespace tip {

constexpr int HF_TO = 256, HF_LN = 32, HF_NW = 4;
typedef float hf_f32x4 __attribute__((ext_vector_type(4)));     // (a native vector: HIP's float4 struct is loaded member by member)

struct HfScale {                 // power-of-two scales of a pass, from the clip value (device memory)
    float s, inv;                // samples are multiplied by s; the total by inv = 1 / (s 2^17)
};
__device__ __forceinline__ HfScale hf_scale(const float *clip_p95, const int *clip_has)
{
    const float c = (clip_has && *clip_has && *clip_p95 > 0.f) ? *clip_p95 : 1.f;
    const int e = min(max((int)((__float_as_uint(c) >> 23) & 0xffu) - 127, -60), 100);      // floor(log2 c) (uint16-derived data: 0 .. 15)
    HfScale h;
    h.s = __uint_as_float((unsigned)(127 + 13 - e) << 23);                 // c s in [2^13, 2^14)
    h.inv = __uint_as_float((unsigned)(127 - 13 + e - 17) << 23);
    return h;
}

// R8 = radius / 8 (sigma 30: 15).  AXIS 1: along y (lines = 32 consecutive x), AXIS 2: along x (lines = 32 consecutive rows).
template <int AXIS, int R8>
__global__ void __launch_bounds__(HF_NW * 64, 2) k_corr_long_f16(const float *__restrict__ in, float *__restrict__ out, int Z, int Y, int X,
                                                                  TapsF taps, const float *__restrict__ clip_p95, const int *__restrict__ clip_has,
                                                                  int *__restrict__ range_flag, int tiles_ln, int tiles_pos, int xcd_bands)
{
    constexpr int r = 8 * R8, NPOS = HF_TO + 2 * r, S = R8 + 2;           // S K-steps per 32-output row tile, S + 2 per wave
    constexpr int PCH = (NPOS / 8) | 1, PITCH = PCH * 8;                   // row pitch: an odd number of 16-byte chunks (in fp16 elements)
    constexpr int WROWS = 2 * S + 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char hf_smem[];
    _Float16 *sB_hi = reinterpret_cast<_Float16 *>(hf_smem), *sB_lo = sB_hi + HF_LN * PITCH;
    uint4 *sW_hi = reinterpret_cast<uint4 *>(sB_lo + HF_LN * PITCH), *sW_lo = sW_hi + WROWS * 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> tile.  The grid is (tiles of a plane, planes).  y pass: line tiles (32-column strips) fastest -- the blocks in flight cover
    // whole rows, and the tiles that share a halo (py +- 1: tiles_ln blocks apart, a multiple of eight) sit on one XCD's L2.  x pass: a
    // line tile is a band of 32 rows and the position tiles of a band lie side by side in memory, so the blocks of ONE XCD (block b runs
    // on XCD b % 8) walk a band's position tiles one after the other: whole rows again, halo neighbours on one L2 (with the line tiles
    // fastest the blocks in flight read 2 KB out of every 8 KB row: 0.52 ms against the y pass's 0.40).
    const int z = blockIdx.y;
    int lx, py;
    if (AXIS == 2 && xcd_bands) {
        const int b = blockIdx.x, k = b & 7, j = b >> 3;
        py = j % tiles_pos;
        lx = k + 8 * (j / tiles_pos);
    } else {
        lx = blockIdx.x % tiles_ln;
        py = blockIdx.x / tiles_ln;
    }
    const int p0 = py * HF_TO, l0 = lx * HF_LN;
    const long P = (long)Y * X;
    const float *src = in + (long)z * P;
    const HfScale sc = hf_scale(clip_p95, clip_has);

    // ---- weight table: entry (u', c) = w~[8 u' - c + e], e = 0..7, w~[d] = w(|d - r|) for 0 <= d <= 2 r, times 2^17, split ---------------
    for (int idx = tid; idx < WROWS * 8; idx += HF_NW * 64) {
        const int up = (idx >> 3) - 3, c = idx & 7;
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = 8 * up - c + e, ad = d - r < 0 ? r - d : d - r;
            const float w = (d >= 0 && d <= 2 * r) ? taps.w[r - ad] * 131072.f : 0.f;
            hi[e] = (_Float16)w;
            lo[e] = (_Float16)((w - (float)hi[e]) * 2048.f);
        }
        sW_hi[idx] = __builtin_bit_cast(uint4, hi);
        sW_lo[idx] = __builtin_bit_cast(uint4, lo);
    }
    // ---- samples ('nearest' edges: clamped coordinates).  ALL of a thread's loads are issued before the first conversion (sixteen
    // 16-byte loads in flight: the block is bound by their latency otherwise -- eight dependent round trips to memory per tile cost ten
    // times the tile's MFMA time), then split and written as 16-byte pieces of [line][position] -----------------------------------------
    // (the staging is the kernel's vector-ALU load -- the matrix pipe is a quarter busy -- so the split is kept short: the range check is
    //  one running maximum, and "a positive sample keeps a positive low piece" is an addend min(vs 2^120, 2^-24) of the low piece -- 2^-24
    //  for every positive sample, 0 for zero: below fp16's resolution wherever the remainder is a normal number, the smallest subnormal
    //  where the sample is lost otherwise: a zero score still means zero inputs; at most 2^-35 absolute in scaled units, inside the bound)
    float vmax = 0.f;
    auto split8 = [&](const float (&v)[8], int line, int g) {
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float vs = v[e] * sc.s;
            vmax = fmax_raw(vmax, vs);
            hi[e] = (_Float16)vs;
            lo[e] = (_Float16)__builtin_fmaf(vs - (float)hi[e], 2048.f, fminf(vs * 1.329227995784916e36f, 5.9604644775390625e-8f));
        }
        *reinterpret_cast<uint4 *>(sB_hi + line * PITCH + 8 * g) = __builtin_bit_cast(uint4, hi);
        *reinterpret_cast<uint4 *>(sB_lo + line * PITCH + 8 * g) = __builtin_bit_cast(uint4, lo);
    };
    constexpr int NG = NPOS / 8;                                  // position groups of eight (62)
    if (AXIS == 2) {
        // unit = (row, position group): consecutive lanes take consecutive groups of one row (32 contiguous bytes each; a lane per
        // 16 bytes with 8-byte LDS writes measured slower)
        constexpr int NIT = (HF_LN * NG + HF_NW * 64 - 1) / (HF_NW * 64);
        hf_f32x4 ld[NIT][2];
        const bool vec = (X & 3) == 0;
        // (line, group) of unit it * 256 + tid without a division per unit: 256 = 4 NG + 8 for the production radius
        constexpr int DL = (HF_NW * 64) / NG, DG = (HF_NW * 64) - DL * NG;
        const int line_0 = tid / NG, g_0 = tid - line_0 * NG;
        int uline[NIT], ug[NIT];
        {
            int line = line_0, g = g_0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                uline[it] = line;
                ug[it] = g;
                line += DL;
                g += DG;
                if (g >= NG) { g -= NG; ++line; }
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int line = uline[it], g = ug[it];
            const float *row = src + (long)min(l0 + min(line, HF_LN - 1), Y - 1) * X;
            const int x0 = p0 - r + 8 * g;
            if (vec && x0 >= 0 && x0 + 8 <= X) {
                ld[it][0] = *reinterpret_cast<const hf_f32x4 *>(row + x0);
                ld[it][1] = *reinterpret_cast<const hf_f32x4 *>(row + x0 + 4);
            } else {
                ld[it][0] = hf_f32x4{row[clampi(x0, 0, X - 1)], row[clampi(x0 + 1, 0, X - 1)], row[clampi(x0 + 2, 0, X - 1)], row[clampi(x0 + 3, 0, X - 1)]};
                ld[it][1] = hf_f32x4{row[clampi(x0 + 4, 0, X - 1)], row[clampi(x0 + 5, 0, X - 1)], row[clampi(x0 + 6, 0, X - 1)], row[clampi(x0 + 7, 0, X - 1)]};
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int line = uline[it], g = ug[it];
            if (line < HF_LN) {
                const float v[8] = {ld[it][0][0], ld[it][0][1], ld[it][0][2], ld[it][0][3], ld[it][1][0], ld[it][1][1], ld[it][1][2], ld[it][1][3]};
                split8(v, line, g);
            }
        }
    } else {
        // unit = (group of eight rows, four adjacent lines): eight 16-byte loads; eight lanes cover a 128-byte row segment
        constexpr int NIT = (NG * (HF_LN / 4) + HF_NW * 64 - 1) / (HF_NW * 64);
        hf_f32x4 ld[NIT][8];
        const int lq = tid & 7;
        const int xq = l0 + 4 * lq;
        const bool vec = (X & 3) == 0 && xq + 4 <= X;
        if (vec) {                               // (two whole loops: merged, the compiler falls back to four 4-byte loads for both)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int g = min(it * (HF_NW * 8) + (tid >> 3), NG - 1), y0 = p0 - r + 8 * g;
#pragma unroll
                for (int e = 0; e < 8; ++e) ld[it][e] = *reinterpret_cast<const hf_f32x4 *>(src + (long)clampi(y0 + e, 0, Y - 1) * X + xq);
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int g = min(it * (HF_NW * 8) + (tid >> 3), NG - 1), y0 = p0 - r + 8 * g;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float *row = src + (long)clampi(y0 + e, 0, Y - 1) * X;
                    ld[it][e] = hf_f32x4{row[min(xq, X - 1)], row[min(xq + 1, X - 1)], row[min(xq + 2, X - 1)], row[min(xq + 3, X - 1)]};
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int g = it * (HF_NW * 8) + (tid >> 3);
            if (g < NG) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ld[it][e].x;
                split8(v, 4 * lq, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ld[it][e].y;
                split8(v, 4 * lq + 1, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ld[it][e].z;
                split8(v, 4 * lq + 2, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ld[it][e].w;
                split8(v, 4 * lq + 3, g);
            }
        }
    }
    if (!(vmax < 40000.f)) atomicOr(range_flag, 8);               // (never for data bounded by the clip value)
    __syncthreads();

    // ---- products ------------------------------------------------------------------------------------------------------------------------
    const int i = lane & 31, h = lane >> 5;
    const int len = AXIS == 1 ? Y : X;
    if (p0 + wave * 64 >= len) return;                                       // (wave-uniform: nothing of this wave's outputs is inside)
    const _Float16 *bh = sB_hi + i * PITCH + wave * 64 + 8 * h, *bl = sB_lo + i * PITCH + wave * 64 + 8 * h;      // + 16 s
    const int wrow0 = (h - (i >> 3) + 3) * 8 + (i & 7);                      // + 16 s: entry (u' = 2 s + h - i / 8, c = i % 8)
    f32x16 thh[2], tm[2], ahh[2], am[2];             // running totals and group accumulators of (hi hi) and (hi lo' + lo' hi)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) { thh[t][q] = 0.f; tm[t][q] = 0.f; ahh[t][q] = 0.f; am[t][q] = 0.f; }
    uint4 fh[3], fl[3];                                                      // weight fragments of steps s, s - 1, s - 2 (rotating)
    auto mm = [](const uint4 &wf, const uint4 &sf, const f32x16 &c) -> f32x16 {
        if (AXIS == 1) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, sf), c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, sf), __builtin_bit_cast(f16x8, wf), c, 0, 0, 0);
    };
    auto flush = [&](int t) {
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            // packed adds; the empty asm pins the sums HERE (left alone, the compiler sinks the whole chain of adds into the store
            // blocks at the kernel's end and keeps -- spills -- every group's accumulators until then)
            f32x2 x = f32x2{thh[t][q], thh[t][q + 1]} + f32x2{ahh[t][q], ahh[t][q + 1]};
            f32x2 y = f32x2{tm[t][q], tm[t][q + 1]} + f32x2{am[t][q], am[t][q + 1]};
            asm volatile("" : "+v"(x), "+v"(y));
            thh[t][q] = x[0]; thh[t][q + 1] = x[1];
            tm[t][q] = y[0]; tm[t][q + 1] = y[1];
            ahh[t][q] = 0.f; ahh[t][q + 1] = 0.f;
            am[t][q] = 0.f; am[t][q + 1] = 0.f;
        }
    };
#pragma unroll
    for (int s = 0; s < S + 2; ++s) {
        const uint4 sh = *reinterpret_cast<const uint4 *>(bh + 16 * s), sl = *reinterpret_cast<const uint4 *>(bl + 16 * s);
        if (s < S) {
            fh[s % 3] = sW_hi[wrow0 + 16 * s];
            fl[s % 3] = sW_lo[wrow0 + 16 * s];
            ahh[0] = mm(fh[s % 3], sh, ahh[0]);
            am[0] = mm(fh[s % 3], sl, am[0]);
            am[0] = mm(fl[s % 3], sh, am[0]);
            if ((s & 1) == 1 || s == S - 1) flush(0);
        }
        if (s >= 2) {
            const int k = (s - 2) % 3;
            ahh[1] = mm(fh[k], sh, ahh[1]);
            am[1] = mm(fh[k], sl, am[1]);
            am[1] = mm(fl[k], sh, am[1]);
            if (((s - 2) & 1) == 1 || s == S + 1) flush(1);
        }
        __builtin_amdgcn_sched_barrier(0);       // (a step's reads are not hoisted over the steps before it: the unrolled loop would otherwise hold every fragment at once)
    }
    // ---- store: lane l holds column l & 31, rows (q & 3) + 8 (q >> 2) + 4 (l >> 5) of each 32 x 32 tile -------------------------------
    float *dst = out + (long)z * P;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int o0 = p0 + wave * 64 + 32 * t;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int rr = (q & 3) + 8 * (q >> 2) + 4 * h;
            const float val = __builtin_fmaf(tm[t][q], 4.8828125e-4f, thh[t][q]) * sc.inv;      // hh + 2^-11 (hl + lh), unscaled (exact)
            if (AXIS == 1) {
                const int yy = o0 + rr, xx = l0 + i;
                if (yy < Y && xx < X) dst[(long)yy * X + xx] = val;
            } else {
                const int yy = l0 + rr, xx = o0 + i;
                if (yy < Y && xx < X) dst[(long)yy * X + xx] = val;
            }
        }
    }
}

}  // namespace tip
