// tip_project.hip -- time_point_surface_projection (sp.py:17-85) as a device pipeline.
//
//   P1  uint16 -> float32, airyscan offset (sp.py:26-29)            fused into every reader of the stack
//   P2  95th percentile of the non-zero reference voxels + clip     exact 65536-bin histogram (sp.py:32-36)
//   P3  Gaussian (0.5,1,1)  (sp.py:37)                              exact scipy-order correlate passes
//   P4  Gaussian (0.5,30,30) score (sp.py:55)                       long-kernel passes (tip_corr.h)
//   P5  argmax over z, first maximum (sp.py:61)
//   P6/P7 one-hot mask + Gaussian (1,2,2) (sp.py:62-71)             z pass = ZxZ table, y pass from the z-map,
//   P8  per-channel max_z(image * mask) -> float64 (sp.py:72-81)    x pass fused with the weighted z-max
#include "tip_slide.h"
#include "tip_preblur.h"
#include "tip_corr_mfma2.h"
#include "tip_corr_f16.h"
#include <atomic>
#include "tip_manifold.h"
#include <algorithm>
#include <cstdlib>

namespace tip {


int correlate1d_dev(const void *in, void *out, int dtype, int Z, int Y, int X, int axis, const Taps &t, int force);

struct ClipInfo {
    double p95d;
    float p95;
    int has;
};

// ---- P2: histogram of the (offset-corrected) reference channel -----------------------------------------
// One persistent 1024-thread block per CU with a private 65536-bin LDS histogram of 16-bit counters packed two per dword.
// A block walks chunks of 57344 voxels; a chunk adds at most 57344 to a bin, so between chunks every bin that has reached
// 8192 is moved to the global histogram (a handful per chunk) and no counter can overflow; the table is cleared once and
// flushed once per block.  (One block per chunk -- 2196 clears, scans and full flushes of the 128 KB table, 6.6 M global
// atomics on ~3000 addresses -- took 0.13 ms; the 252 MB read alone is 0.06 ms.)
constexpr int HIST_PER_BLOCK = 57344;  // 7 trips of 1024 threads x 8 voxels; 57344 + 8191 < 65536
__global__ void __launch_bounds__(1024) k_hist_u16(const uint16_t *__restrict__ in, long n, int airy,
                                                   unsigned long long *__restrict__ hist)
{
    __shared__ unsigned int h[32768];
    for (int i = threadIdx.x; i < 32768; i += 1024) h[i] = 0;
    __syncthreads();
    const long nchunks = (n + HIST_PER_BLOCK - 1) / HIST_PER_BLOCK;
    for (long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const long base = chunk * HIST_PER_BLOCK;
        constexpr int TRIPS = HIST_PER_BLOCK / (1024 * 8);
        if (base + HIST_PER_BLOCK <= n && ((((uintptr_t)(in + base)) & 15) == 0)) {
            // whole chunk: all seven 16-byte loads of the thread in flight before the first LDS atomic (one block per CU:
            // nothing else hides the memory latency)
            uint4 v[TRIPS];
#pragma unroll
            for (int k = 0; k < TRIPS; ++k) v[k] = *reinterpret_cast<const uint4 *>(in + base + ((long)k * 1024 + threadIdx.x) * 8);
#pragma unroll
            for (int k = 0; k < TRIPS; ++k) {
                const unsigned int w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        int val = (int)((w[j] >> (16 * s)) & 0xffffu);
                        if (airy) { val -= 10000; if (val < 0) val = 0; }
                        atomicAdd(&h[val >> 1], 1u << (16 * (val & 1)));
                    }
                }
            }
        } else {
            for (int k = 0; k < TRIPS; ++k) {
                const long i0 = base + ((long)k * 1024 + threadIdx.x) * 8;
                for (int j = 0; j < 8; ++j) {
                    if (i0 + j < n) {
                        int val = in[i0 + j];
                        if (airy) { val -= 10000; if (val < 0) val = 0; }
                        atomicAdd(&h[val >> 1], 1u << (16 * (val & 1)));
                    }
                }
            }
        }
        __syncthreads();
        if (chunk + gridDim.x < nchunks) {      // another chunk follows: make room in the bins that are filling up
            for (int i = threadIdx.x; i < 32768; i += 1024) {
                unsigned int c = h[i];
                if ((c & 0xffffu) >= 8192u) { atomicAdd(&hist[2 * i], (unsigned long long)(c & 0xffffu)); c &= 0xffff0000u; h[i] = c; }
                if ((c >> 16) >= 8192u) { atomicAdd(&hist[2 * i + 1], (unsigned long long)(c >> 16)); h[i] = c & 0xffffu; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < 32768; i += 1024) {
        const unsigned int c = h[i];
        if (c & 0xffffu) atomicAdd(&hist[2 * i], (unsigned long long)(c & 0xffffu));
        if (c >> 16) atomicAdd(&hist[2 * i + 1], (unsigned long long)(c >> 16));
    }
}

// The same histogram over a sub-box [z0,z1) x [y0,y1) x [x0,x1) of one channel plane stack (row length X, plane size
// Y*X): spatially tiled frames (config 5) count every voxel once -- tile interiors -- and add the tiles' histograms up
// before anyone clips (the percentile is a property of the whole frame).
__global__ void __launch_bounds__(1024) k_hist_u16_box(const uint16_t *__restrict__ in, int Y, int X, int z0, int y0, int x0, int bz,
                                                       int by, int bx, int airy, unsigned long long *__restrict__ hist)
{
    __shared__ unsigned int h[32768];
    for (int i = threadIdx.x; i < 32768; i += 1024) h[i] = 0;
    __syncthreads();
    const long n = (long)bz * by * bx;
    const long base = (long)blockIdx.x * HIST_PER_BLOCK;
    for (int k = 0; k < HIST_PER_BLOCK / 1024; ++k) {
        const long i = base + (long)k * 1024 + threadIdx.x;
        if (i < n) {
            const int x = (int)(i % bx), y = (int)((i / bx) % by), z = (int)(i / ((long)bx * by));
            int val = in[((long)(z0 + z) * Y + (y0 + y)) * X + (x0 + x)];
            if (airy) { val -= 10000; if (val < 0) val = 0; }
            atomicAdd(&h[val >> 1], 1u << (16 * (val & 1)));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32768; i += 1024) {
        const unsigned int c = h[i];
        if (c & 0xffffu) atomicAdd(&hist[2 * i], (unsigned long long)(c & 0xffffu));
        if (c >> 16) atomicAdd(&hist[2 * i + 1], (unsigned long long)(c >> 16));
    }
}

// np.percentile(nonzero, 95) with numpy 1.26 arithmetic (see oracle.percentile_linear): one block.
// (with_zero: over all voxels, zeros included -- the second channel of method 'multi_channel', sp.py:46)
__global__ void __launch_bounds__(1024) k_percentile95(const unsigned long long *__restrict__ hist, ClipInfo *out, int with_zero)
{
    // 1024 threads x 64 bins.  Chunk sums -> block-wide inclusive scan (wave shuffles + 16 wave totals) -> the chunk that
    // holds a rank is the one thread whose [exclusive, inclusive) range contains it -> one wave scans that chunk's 64 bins.
    // (A single thread walking the 1024 chunk sums and the bins took 35 us: the whole projection waits for this value.)
    __shared__ unsigned long long wave_tot[16];
    __shared__ unsigned long long s_n;
    __shared__ int s_chunk[2];
    __shared__ unsigned long long s_before[2];
    __shared__ int s_val[2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned long long s = 0;
    for (int b = t * 64; b < t * 64 + 64; ++b)
        if (b > 0 || with_zero) s += hist[b];
    unsigned long long inc = s;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    unsigned long long base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    inc += base;
    if (t == 1023) s_n = inc;
    __syncthreads();
    const unsigned long long n = s_n;
    if (n == 0) {
        if (t == 0) { out->has = 0; out->p95 = 0.f; out->p95d = 0.0; }
        return;
    }
    const double quant = 95.0 / 100.0;
    const double virt = (double)(n - 1) * quant;  // numpy 'linear': (n - 1) * quantiles
    const double fl = floor(virt);
    const double gamma = virt - fl;
    long long prev = (long long)fl;
    if (prev < 0) prev = 0;
    if (prev > (long long)n - 1) prev = (long long)n - 1;
    long long next = prev + 1;
    if (next > (long long)n - 1) next = (long long)n - 1;
    const unsigned long long exc = inc - s;
    if (s > 0) {          // the chunk whose cumulative range holds the rank (exactly one thread per rank)
        if (exc <= (unsigned long long)prev && (unsigned long long)prev < inc) { s_chunk[0] = t; s_before[0] = exc; }
        if (exc <= (unsigned long long)next && (unsigned long long)next < inc) { s_chunk[1] = t; s_before[1] = exc; }
    }
    __syncthreads();
    if (wave < 2) {       // wave 0: value at rank prev, wave 1: at rank next
        const unsigned long long k = (unsigned long long)(wave == 0 ? prev : next) - s_before[wave];
        const int b = s_chunk[wave] * 64 + lane;
        unsigned long long c = (b > 0 || with_zero) ? hist[b] : 0ULL, ci = c;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(ci, d, 64);
            if (lane >= d) ci += o;
        }
        // first bin whose inclusive count exceeds k
        const unsigned long long hit = __ballot(ci > k);
        if (lane == 0) s_val[wave] = hit ? s_chunk[wave] * 64 + (__ffsll((long long)hit) - 1) : 65535;
    }
    __syncthreads();
    if (t != 0) return;
    const float lo = (float)s_val[0], hi = (float)s_val[1];
    const double diff = (double)(hi - lo);
    double res = (double)lo + diff * gamma;
    if (gamma >= 0.5) res = (double)hi - diff * (1.0 - gamma);
    out->p95d = res;
    out->p95 = (float)res;  // numpy 1.x value-based casting: compared and assigned as float32 (sp.py:36)
    out->has = 1;
}

// ---- P1+P2 fused reader of the reference channel ----------------------------------------------------------
struct LoadU16Clip {
    const uint16_t *p;
    long sz, sy;
    int airy;
    const ClipInfo *clip;
    __device__ __forceinline__ double operator()(int z, int y, int x) const
    {
        float f = (float)p[z * sz + y * sy + x];
        if (airy) { f -= 10000.f; if (f < 0.f) f = 0.f; }
        if (clip->has && f > clip->p95) f = clip->p95;
        return (double)f;
    }
};

// ---- P5: first-maximum argmax over z ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_argmax_z(const float *__restrict__ score, int Z, long P, int min_z,
                                                  int atoh_shift, int32_t *__restrict__ zsel,
                                                  int32_t *__restrict__ zsel_atoh, int64_t *__restrict__ zmap,
                                                  int *__restrict__ err)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float best = score[p];
    int bi = 0;
    for (int z = 1; z < Z; ++z) {
        const float s = score[(long)z * P + p];
        if (s > best) { best = s; bi = z; }
    }
    const int cz = min_z + bi;               // sp.py:61
    if (zmap) zmap[p] = cz;
    int ca = cz;
    if (atoh_shift != 0) {                   // sp.py:62 np.clip(chosen_z + shift, 0, Z)  (upper bound Z, sic)
        ca = cz + atoh_shift;
        ca = ca < 0 ? 0 : (ca > Z ? Z : ca);
    }
    if (cz >= Z || ca >= Z) {                // the reference's fancy index would raise IndexError (sp.py:68-69)
        atomicOr(err, 1);
        zsel[p] = cz >= Z ? Z - 1 : cz;
        zsel_atoh[p] = ca >= Z ? Z - 1 : ca;
        return;
    }
    zsel[p] = cz;
    zsel_atoh[p] = ca;
}

__global__ void k_identity(float *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * n) out[i] = (i / n == i % n) ? 1.f : 0.f;
}

// ---- P6/P7: the z pass of a one-hot column is a table lookup -----------------------------------------------------
struct LoadMaskTable {
    const float *T;       // T[z*Zs + z0] = z-blurred one-hot(z0) at z
    const int32_t *zsel;  // (Y,X) chosen plane
    int Zs;
    long X;
    __device__ __forceinline__ double operator()(int z, int y, int x) const
    {
        return (double)T[z * Zs + zsel[(long)y * X + x]];
    }
};

// ---- P7 x pass + P8 weighted z-max, all channels in one sweep ---------------------------------------------------
// ymask: (Zs,Y,X) float32 after the z and y passes.  For every pixel: for each z, finish the x pass (exact
// scipy order), round to float32, multiply with the float32 image value and keep the per-channel maximum.
// chan_mask selects which channels this launch writes (reference vs. atoh-shifted mask, sp.py:75-79).
template <int MAXC>
__global__ void __launch_bounds__(256) k_xpass_wmax(const float *__restrict__ ymask, const uint16_t *__restrict__ img,
                                                    int C, int Zfull, int zlo, int Zs, int Y, int X, int airy,
                                                    unsigned chan_mask, Taps taps, double *__restrict__ proj)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= X) return;
    const int r = taps.n >> 1;
    const long P = (long)Y * X;
    float mx[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) mx[c] = 0.f;
    for (int z = 0; z < Zs; ++z) {
        const float *row = ymask + (long)z * P + (long)y * X;
        double tmp = (double)row[x] * taps.w[r];
        for (int d = r; d >= 1; --d) {
            const float a = row[clampi(x - d, 0, X - 1)], b = row[clampi(x + d, 0, X - 1)];
            tmp += ((double)a + (double)b) * taps.w[r - d];
        }
        const float m = (float)tmp;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c < C && ((chan_mask >> c) & 1u)) {
                float v = (float)img[((long)c * Zfull + zlo + z) * P + (long)y * X + x];
                if (airy) { v -= 10000.f; if (v < 0.f) v = 0.f; }
                const float pr = v * m;
                // np.max over z of a float32 array; first z initialises
                mx[c] = z == 0 ? pr : (pr > mx[c] ? pr : mx[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C && ((chan_mask >> c) & 1u)) proj[(long)c * P + (long)y * X + x] = (double)mx[c];
}

// ---- fast mask stage: the blurred one-hot mask is exactly zero more than 4 planes away from every chosen plane that
// feeds it, and adding exact zeros changes nothing, so only the planes inside [zmin-4, zmax+4] of the 17-wide windows
// are computed (same arithmetic, same order, bit-identical to the dense kernels above) -----------------------------------
constexpr int MASK_R = 8, MASK_W = 2 * MASK_R + 1, MASK_ZR = 4;

template <int SEG>
__global__ void __launch_bounds__(256) k_mask_y_sparse(const float *__restrict__ table, const int32_t *__restrict__ zsel, int Zs,
                                                       int Y, int X, Taps taps, float *__restrict__ out,
                                                       int32_t *__restrict__ zrange)
{
    extern __shared__ float sT[];  // Zs*Zs: T[z*Zs + z0]
    for (int i = threadIdx.x; i < Zs * Zs; i += blockDim.x) sT[i] = table[i];
    __syncthreads();
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= X) return;
    const int y0 = blockIdx.y * (SEG * MASK_W);
    const long P = (long)Y * X;
    int win[MASK_W];
#pragma unroll
    for (int i = 0; i < MASK_W; ++i) win[i] = zsel[(long)clampi(y0 - MASK_R + i, 0, Y - 1) * X + x];
    for (int s = 0; s < SEG; ++s) {
#pragma unroll
        for (int o = 0; o < MASK_W; ++o) {
            const int y = y0 + s * MASK_W + o;
            if (y < Y) {
                int zmin = win[0], zmax = win[0];
#pragma unroll
                for (int i = 1; i < MASK_W; ++i) { zmin = min(zmin, win[i]); zmax = max(zmax, win[i]); }
                zrange[(long)y * X + x] = zmin | (zmax << 16);
                const int za = max(zmin - MASK_ZR, 0), zb = min(zmax + MASK_ZR, Zs - 1);
                float *dst = out + (long)y * X + x;
                for (int z = 0; z < za; ++z) dst[(long)z * P] = 0.f;
                for (int z = za; z <= zb; ++z) {
                    const float *Tz = sT + z * Zs;
                    double tmp = (double)Tz[win[(o + MASK_R) % MASK_W]] * taps.w[MASK_R];
#pragma unroll
                    for (int d = MASK_R; d >= 1; --d)
                        tmp += ((double)Tz[win[(o + MASK_R - d) % MASK_W]] + (double)Tz[win[(o + MASK_R + d) % MASK_W]]) *
                               taps.w[MASK_R - d];
                    dst[(long)z * P] = (float)tmp;
                }
                for (int z = zb + 1; z < Zs; ++z) dst[(long)z * P] = 0.f;
            }
            win[o % MASK_W] = zsel[(long)clampi(y + MASK_R + 1, 0, Y - 1) * X + x];
        }
    }
}

template <int MAXC>
__global__ void __launch_bounds__(256) k_xpass_wmax_sparse(const float *__restrict__ ymask, const int32_t *__restrict__ zrange,
                                                           const uint16_t *__restrict__ img, int C, int Zfull, int zlo, int Zs,
                                                           int Y, int X, int airy, unsigned chan_mask, Taps taps,
                                                           double *__restrict__ proj)
{
    constexpr int N = 8 + 2 * MASK_R;  // 24 inputs for 8 outputs
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const int y = blockIdx.y;
    if (x0 >= X) return;
    const long P = (long)Y * X;
    const bool interior = x0 - MASK_R >= 0 && x0 + 8 + MASK_R <= X && (X & 3) == 0;
    int zmin = 1 << 30, zmax = -1;
    for (int i = 0; i < N; ++i) {
        const int r = zrange[(long)y * X + clampi(x0 - MASK_R + i, 0, X - 1)];
        zmin = min(zmin, r & 0xffff);
        zmax = max(zmax, r >> 16);
    }
    const int za = max(zmin - MASK_ZR, 0), zb = min(zmax + MASK_ZR, Zs - 1);
    float mx[MAXC][8];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) mx[c][k] = 0.f;
    for (int z = za; z <= zb; ++z) {
        const float *row = ymask + (long)z * P + (long)y * X;
        float v[N];
        if (interior) {
#pragma unroll
            for (int i = 0; i < N / 4; ++i) {
                const float4 f = *reinterpret_cast<const float4 *>(row + x0 - MASK_R + 4 * i);
                v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = row[clampi(x0 - MASK_R + i, 0, X - 1)];
        }
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double tmp = (double)v[k + MASK_R] * taps.w[MASK_R];
#pragma unroll
            for (int d = MASK_R; d >= 1; --d)
                tmp += ((double)v[k + MASK_R - d] + (double)v[k + MASK_R + d]) * taps.w[MASK_R - d];
            m[k] = (float)tmp;
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c < C && ((chan_mask >> c) & 1u)) {
                const uint16_t *ip = img + ((long)c * Zfull + zlo + z) * P + (long)y * X + x0;
                unsigned short pix[8];
                if (x0 + 8 <= X && (X & 7) == 0) {
                    const uint4 u = *reinterpret_cast<const uint4 *>(ip);
                    pix[0] = u.x & 0xffff; pix[1] = u.x >> 16; pix[2] = u.y & 0xffff; pix[3] = u.y >> 16;
                    pix[4] = u.z & 0xffff; pix[5] = u.z >> 16; pix[6] = u.w & 0xffff; pix[7] = u.w >> 16;
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) pix[k] = x0 + k < X ? ip[k] : 0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (x0 + k < X) {
                        float val = (float)pix[k];
                        if (airy) { val -= 10000.f; if (val < 0.f) val = 0.f; }
                        const float pr = val * m[k];
                        mx[c][k] = pr > mx[c][k] ? pr : mx[c][k];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C && ((chan_mask >> c) & 1u))
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (x0 + k < X) proj[(long)c * P + (long)y * X + x0 + k] = (double)mx[c][k];
}

// ---- P7 y + x pass and P8 fused: the blurred one-hot mask never reaches HBM -----------------------------------------------
// One block = FT_Y x FT_X outputs.  The chosen-plane map of the tile (+ 8-pixel halo, edges replicated) sits in LDS; plane
// by plane the block computes the y pass of the tile's columns into an LDS buffer (exact scipy order, zero outside a
// column's [zmin - 4, zmax + 4] exactly as the dense pass gives) and every thread finishes the x pass for its 8 outputs
// from that buffer, multiplies with the raw stack and keeps the per-channel maximum -- the arithmetic of k_mask_y_sparse
// + k_xpass_wmax_sparse, without writing the 503 MB mask volume and reading a third of it back.
constexpr int FT_Y = 16, FT_X = 128, FT_W = FT_X + 2 * MASK_R, FT_H = FT_Y + 2 * MASK_R;
template <int MAXC>
__global__ void __launch_bounds__(256) k_mask_wmax_fused(const float *__restrict__ table, const int32_t *__restrict__ zsel,
                                                         const uint16_t *__restrict__ img, int C, int Zfull, int zlo, int Zs, int Y,
                                                         int X, int airy, unsigned chan_mask, Taps taps, double *__restrict__ proj)
{
    extern __shared__ float sT[];                       // Zs * Zs: T[z * Zs + z0]
    __shared__ unsigned char zs[FT_H][FT_W];            // chosen plane, rows y0-8 .. y0+FT_Y+7, columns x0-8 .. x0+FT_X+7
    __shared__ unsigned char cmin[FT_Y][FT_W], cmax[FT_Y][FT_W];   // range of the 17-row window under every (row, column)
    __shared__ __attribute__((aligned(16))) float ybuf[2][FT_Y][FT_W];
    __shared__ int s_lo, s_hi;
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * FT_X, y0 = blockIdx.y * FT_Y;
    const long P = (long)Y * X;
    for (int i = t; i < Zs * Zs; i += 256) sT[i] = table[i];
    for (int i = t; i < FT_H * FT_W; i += 256) {
        const int r = i / FT_W, c = i - r * FT_W;
        zs[r][c] = (unsigned char)zsel[(long)clampi(y0 - MASK_R + r, 0, Y - 1) * X + clampi(x0 - MASK_R + c, 0, X - 1)];
    }
    if (t == 0) { s_lo = Zs; s_hi = -1; }
    __syncthreads();
    int lo = Zs, hi = -1;
    for (int i = t; i < FT_Y * FT_W; i += 256) {
        const int r = i / FT_W, c = i - r * FT_W;
        int a = zs[r][c], b = a;
#pragma unroll
        for (int k = 1; k < MASK_W; ++k) { const int v = zs[r + k][c]; a = min(a, v); b = max(b, v); }
        cmin[r][c] = (unsigned char)a; cmax[r][c] = (unsigned char)b;
        lo = min(lo, a); hi = max(hi, b);
    }
    for (int d = 32; d >= 1; d >>= 1) { lo = min(lo, __shfl_xor(lo, d, 64)); hi = max(hi, __shfl_xor(hi, d, 64)); }
    if ((t & 63) == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    __syncthreads();
    const int zaB = max(s_lo - MASK_ZR, 0), zbB = min(s_hi + MASK_ZR, Zs - 1);
    // this thread's outputs: row ty, columns cx .. cx+7 of the tile; their z range = the windows cx .. cx+23 of row ty
    const int ty = t >> 4, cx = (t & 15) * 8;
    const int gy = y0 + ty, gx = x0 + cx;
    const bool live = gy < Y && gx < X;
    int za = Zs, zb = -1;
    if (live) {
        for (int c = cx; c < cx + 8 + 2 * MASK_R; ++c) { za = min(za, (int)cmin[ty][c]); zb = max(zb, (int)cmax[ty][c]); }
        za = max(za - MASK_ZR, 0); zb = min(zb + MASK_ZR, Zs - 1);
    }
    float mx[MAXC][8];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) mx[c][k] = 0.f;
    for (int z = zaB; z <= zbB; ++z) {
        float(*yb)[FT_W] = ybuf[(z - zaB) & 1];
        const float *Tz = sT + z * Zs;
        for (int i = t; i < FT_Y * FT_W; i += 256) {     // y pass of plane z for the tile's columns
            const int r = i / FT_W, c = i - r * FT_W;
            float val = 0.f;
            if (z >= (int)cmin[r][c] - MASK_ZR && z <= (int)cmax[r][c] + MASK_ZR) {
                double tmp = (double)Tz[zs[r + MASK_R][c]] * taps.w[MASK_R];
#pragma unroll
                for (int d = MASK_R; d >= 1; --d)
                    tmp += ((double)Tz[zs[r + MASK_R - d][c]] + (double)Tz[zs[r + MASK_R + d][c]]) * taps.w[MASK_R - d];
                val = (float)tmp;
            }
            yb[r][c] = val;
        }
        __syncthreads();        // (the buffer of plane z-1 is free again only after the NEXT barrier: two buffers, one barrier per plane)
        if (live && z >= za && z <= zb) {
            float v[8 + 2 * MASK_R];
#pragma unroll
            for (int i = 0; i < (8 + 2 * MASK_R) / 4; ++i) {
                const float4 f = *reinterpret_cast<const float4 *>(&yb[ty][cx + 4 * i]);
                v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
            }
            float m[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                double tmp = (double)v[k + MASK_R] * taps.w[MASK_R];
#pragma unroll
                for (int d = MASK_R; d >= 1; --d)
                    tmp += ((double)v[k + MASK_R - d] + (double)v[k + MASK_R + d]) * taps.w[MASK_R - d];
                m[k] = (float)tmp;
            }
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if (c < C && ((chan_mask >> c) & 1u)) {
                    const uint16_t *ip = img + ((long)c * Zfull + zlo + z) * P + (long)gy * X + gx;
                    unsigned short pix[8];
                    if (gx + 8 <= X && (X & 7) == 0) {
                        const uint4 u = *reinterpret_cast<const uint4 *>(ip);
                        pix[0] = u.x & 0xffff; pix[1] = u.x >> 16; pix[2] = u.y & 0xffff; pix[3] = u.y >> 16;
                        pix[4] = u.z & 0xffff; pix[5] = u.z >> 16; pix[6] = u.w & 0xffff; pix[7] = u.w >> 16;
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) pix[k] = gx + k < X ? ip[k] : 0;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float val = (float)pix[k];
                        if (airy) { val -= 10000.f; if (val < 0.f) val = 0.f; }
                        const float pr = val * m[k];
                        mx[c][k] = pr > mx[c][k] ? pr : mx[c][k];
                    }
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C && ((chan_mask >> c) & 1u))
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (gx + k < X) proj[(long)c * P + (long)gy * X + gx + k] = (double)mx[c][k];
    }
}

// Configuration of the fast sigma-30 passes: 3 / 4 = the matrix-core kernels (tip_corr_mfma.h / tip_corr_mfma2.h), else
// variant * 10000 + (waves per block) * 100 + (outputs per lane and group) of the VALU kernel k_corr_long_fast.
// Measured on the 2048 x 2048 x 30 frame, one frame in flight, y / x pass: VALU 0.79 / 0.81 ms, 3: 0.67 / 0.69 ms,
// 4: 0.76 / 0.69 ms; whole classical pipeline 155 / 166.6 / 169.4 frames/s for (VALU, VALU) / (3, 3) / (3, 4), and
// 242.7 / 249.8 / 252.2 with four frames in flight.
#ifndef FAST_CFG_Y
#define FAST_CFG_Y 5
#endif
#ifndef FAST_CFG_X
#define FAST_CFG_X 5
#endif

static int cu_count()
{
    static int cus = 0;          // (same device model on every GPU of a node; a benign race writes the same value)
    if (!cus) {
        hipDeviceProp_t prop;
        cus = hipGetDeviceProperties(&prop, ctx().device) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}

template <int AXIS, int NW, int R, int VAR, int REM = 0>
static int launch_fast_cfg(const float *in, float *out, int Zs, int Y, int X, const TapsF &t)
{
    const int r = t.n >> 1;
    if (VAR != 0 && r % (R / 2) != REM) return fail(TIP_ERR_ARG, "fast pass: radius %d mod %d is not %d", r, R / 2, REM);
    const size_t lds = (size_t)(256 + 2 * r) * (AXIS == 1 ? 64 : 65) * sizeof(float);
    auto k = k_corr_long_fast<AXIS, 256, NW, R, VAR, REM>;
    TIP_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid = AXIS == 1 ? dim3(cdiv(X, 64), cdiv(Y, 256), Zs) : dim3(cdiv(Y, 64), cdiv(X, 256), Zs);
    TIP_LAUNCH(AXIS == 1 ? "score_fast_y" : "score_fast_x", k, grid, dim3(NW * 64), lds, in, out, Zs, Y, X, t);
    return TIP_OK;
}

// the same pass on the matrix cores (tip_corr_mfma.h)
template <int AXIS>
static int launch_mfma(const float *in, float *out, int Zs, int Y, int X, const TapsF &t)
{
    const int r = t.n >> 1;
    if (r < 1 || r > 127 || (16 + r) % MF_SEG) return fail(TIP_ERR_ARG, "mfma pass: radius %d (16 + r must be a multiple of %d)", r, MF_SEG);
    const int npos = MF_TO + 2 * r;
    const size_t lds = AXIS == 1 ? (size_t)npos * MF_LN * sizeof(float)
                                 : (size_t)MF_LN * (npos + 1 + (npos & 1)) * sizeof(float);
    auto k = k_corr_long_mfma<AXIS>;
    TIP_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles_pos = cdiv(AXIS == 1 ? Y : X, MF_TO), tiles_ln = cdiv(AXIS == 1 ? X : Y, MF_LN);
    const int ntiles = tiles_pos * tiles_ln * Zs;
    const int cus = cu_count();
    const int per_cu = tuning().mfma_blocks_per_cu;
    const int blocks = std::min(ntiles, per_cu * cus);  // persistent blocks, two per CU (LDS: 64 KB each); tuning hook: one
    TIP_LAUNCH(AXIS == 1 ? "score_fast_y" : "score_fast_x", k, dim3(blocks), dim3(MF_NW * 64), lds, in, out, Zs, Y, X, t, ntiles,
               tiles_pos, tiles_ln);
    return TIP_OK;
}

// ... with asynchronous global -> LDS copies and one double-buffered persistent block per CU (tip_corr_mfma2.h)
template <int AXIS>
static int launch_mfma2(const float *in, float *out, int Zs, int Y, int X, const TapsF &t)
{
    const int r = t.n >> 1;
    if (r < 8 || r > 120 || r % MF_SEG) return fail(TIP_ERR_ARG, "mfma pass: radius %d (a multiple of %d in [8, 120])", r, MF_SEG);
    const int npos = MF_TO + 2 * r;
    const int pitch = 513;                                // AXIS 2: 512 copied positions per line; odd: ds_read_b32 / ds_read2_b32 bank
                                                          // on (address / 4) mod 32 within each 32-lane half, and a half holds the
                                                          // 32 lines of one position (pitch 514 made lines l and l + 16 collide)
    const int bufsz = AXIS == 1 ? npos * MF_LN : MF_LN * pitch;
    const size_t lds = ((size_t)2 * bufsz + 2 * 127 + 64 + 64) * sizeof(float);   // + padded kernel + sink
    if (lds > 160 * 1024) return fail(TIP_ERR_ARG, "mfma pass: tile buffers exceed the LDS");
    auto k = k_corr_long_mfma2<AXIS>;
    TIP_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles_pos = cdiv(AXIS == 1 ? Y : X, MF_TO), tiles_ln = cdiv(AXIS == 1 ? X : Y, MF_LN);
    const long nt = (long)tiles_pos * tiles_ln * Zs;
    if (nt > 0x7fffffffL - 4096) return fail(TIP_ERR_ARG, "mfma pass: too many tiles");
    const int ntiles = (int)nt;
    const int cus = cu_count();
    const int blocks = std::min(ntiles, cus);             // persistent blocks, one per CU (two 62 KB tile buffers each)
    TIP_LAUNCH(AXIS == 1 ? "score_fast_y" : "score_fast_x", k, dim3(blocks), dim3(MF_NW * 64), lds, in, out, Zs, Y, X, t, ntiles,
               tiles_pos, tiles_ln, pitch, bufsz);
    return TIP_OK;
}

// ... on the fp16 matrix cores with split operands (tip_corr_f16.h): radius 120 (sigma 30), data bounded by the clip value
template <int AXIS>
static int launch_f16(const float *in, float *out, int Zs, int Y, int X, const TapsF &t, const ClipInfo *clip, int *range_flag)
{
    constexpr int R8 = 15, NPOS = HF_TO + 16 * R8, PITCH = ((NPOS / 8) | 1) * 8, WROWS = 2 * (R8 + 2) + 3;
    const size_t lds = (size_t)2 * HF_LN * PITCH * 2 + (size_t)2 * WROWS * 8 * 16;
    auto k = k_corr_long_f16<AXIS, R8>;
    static std::atomic<unsigned> attr_done[64];
    const int dev = ctx().device >= 0 && ctx().device < 64 ? ctx().device : 0;
    if (!(attr_done[dev].load(std::memory_order_acquire) & (1u << AXIS))) {
        TIP_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done[dev].fetch_or(1u << AXIS, std::memory_order_release);
    }
    const int tiles_ln = cdiv(AXIS == 1 ? X : Y, HF_LN), tiles_pos = cdiv(AXIS == 1 ? Y : X, HF_TO);
    if ((long)tiles_ln * tiles_pos > 0x7fffffffL || Zs > 65535) return fail(TIP_ERR_ARG, "f16 score pass: grid too large");
    const dim3 grid((unsigned)(tiles_ln * tiles_pos), (unsigned)Zs);
    const int xcd_bands = (AXIS == 2 && tiles_ln % 8 == 0) ? 1 : 0;
    TIP_LAUNCH(AXIS == 1 ? "score_fast_y" : "score_fast_x", k, grid, dim3(HF_NW * 64), lds, in, out, Zs, Y, X, t, &clip->p95, &clip->has, range_flag,
               tiles_ln, tiles_pos, xcd_bands);
    return TIP_OK;
}

template <int AXIS>
static int launch_fast(int cfg, const float *in, float *out, int Zs, int Y, int X, const TapsF &t, const ClipInfo *clip = nullptr,
                       int *range_flag = nullptr)
{
    const int r = t.n >> 1;
    if (cfg == 5 && (r != 120 || !clip || !range_flag)) cfg = AXIS == 1 ? 3 : 4;          // the fp16 tiles: sigma 30 on clipped data only
    if (cfg == 5) return launch_f16<AXIS>(in, out, Zs, Y, X, t, clip, range_flag);
    if ((cfg == 3 || cfg == 4) && (r < 8 || r > 120 || r % MF_SEG)) cfg = 11616;   // the MFMA tiles need radius % 8 == 0 (sigma 30: 120)
    if (cfg == 3) return launch_mfma<AXIS>(in, out, Zs, Y, X, t);
    if (cfg == 4) return launch_mfma2<AXIS>(in, out, Zs, Y, X, t);
    if (cfg == 10832 && r % 16 != 0 && r % 16 != 8) cfg = 11616;   // 32 outputs per lane: radius % 16 must be 0 or 8
    if (cfg / 10000 == 1 && cfg != 10832 && (r % 8)) cfg -= 10000;  // the packed variant needs radius % H == 0
    switch (cfg) {   // variant * 10000 + waves * 100 + outputs per lane
    case 1616: return launch_fast_cfg<AXIS, 16, 16, 0>(in, out, Zs, Y, X, t);
    case 11616: return launch_fast_cfg<AXIS, 16, 16, 1>(in, out, Zs, Y, X, t);
    case 10832:   // 8 waves, 32 outputs per lane: half the LDS window reads per output, 2 waves per SIMD
        return r % 16 == 0 ? launch_fast_cfg<AXIS, 8, 32, 1, 0>(in, out, Zs, Y, X, t) : launch_fast_cfg<AXIS, 8, 32, 1, 8>(in, out, Zs, Y, X, t);
    default: return fail(TIP_ERR_ARG, "unknown fast-pass configuration %d", cfg);
    }
}

// ---- certified argmax --------------------------------------------------------------------------------------------------
// The sigma-30 score is used for ONE thing: chosen_z = argmax_z(score) (sp.py:55-61).  So the score itself need not be
// exact -- only the argmax must be.  The fast passes (k_corr_long_fast) give S~ with |S~ - S| <= EPS * S for the exact
// float32 score S: each fast pass is within a = (1+u)^36 - 1 of the real-number sum (short partial sums, see tip_corr.h)
// and scipy's pass within 1.0001u of it (u = 2^-24, all terms non-negative), two passes compose to 2a + 2.1u < 74.2u;
// EPS = 96u leaves 29 % headroom.  (With one running sum per output a was 123u and EPS 320u: the uncertified pixels --
// and the cost of the exact fix-up -- scale with EPS.)
// A pixel is certified when its best fast score beats every other plane by more than both error bars; the few that
// are not (top two planes closer than ~1.2e-5 relative: the lines where the surface crosses between planes) are
// recomputed in exact scipy arithmetic from the z-passed volume, for the candidate planes only.
#define CERT_EPS (96.0f * 5.9604644775390625e-8f)
// absolute part of the fast passes' error: none for the float32 tiles (relative down to underflow); the fp16 tiles keep 22 bits
// relative down to 2^-36 of the clip value and are off by at most clip 2^-47 below that (tip_corr_f16.h) -- 2^-44 here
__device__ __forceinline__ float cert_abs(const ClipInfo *clip)
{
    return (clip && clip->has) ? clip->p95 * 5.684341886080802e-14f + 1e-30f : 1e-30f;
}

// four adjacent pixels per thread (one float4 per plane): enough bytes in flight to stream the score volume
__global__ void __launch_bounds__(256) k_argmax_certify(const float *__restrict__ score, int Z, long P, int *__restrict__ best_z,
                                                        int *__restrict__ unc_list, int *__restrict__ unc_count, const ClipInfo *clip)
{
    __shared__ int s_cnt[4], s_base;
    const float dabs = cert_abs(clip);
    const long p0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const bool vec = p0 + 4 <= P && (P & 3) == 0;
    float b1[4], b2[4];
    int z1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { b1[k] = -1.f; b2[k] = -1.f; z1[k] = 0; }
    if (p0 < P) {
        for (int z = 0; z < Z; ++z) {
            float v[4];
            if (vec) {
                const float4 f = *reinterpret_cast<const float4 *>(score + (long)z * P + p0);
                v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = p0 + k < P ? score[(long)z * P + p0 + k] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // scores are >= 0: the -1 start makes plane 0 the first maximum
                if (v[k] > b1[k]) { b2[k] = b1[k]; b1[k] = v[k]; z1[k] = z; }
                else if (v[k] > b2[k]) b2[k] = v[k];
            }
        }
    }
    unsigned unc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (p0 + k >= P) continue;
        best_z[p0 + k] = z1[k];
        // b1 == 0: every plane is exactly zero in the fast pass, hence (no underflow for uint16-derived data) in the exact one
        const bool certain = (b1[k] == 0.f) || (Z == 1) || (b1[k] * (1.f - CERT_EPS) - dabs > b2[k] * (1.f + CERT_EPS) + dabs);
        if (!certain) unc |= 1u << k;
    }
    // append with one atomic per block (not per pixel: same-address atomics serialise in L2)
    const int mine = __popc(unc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = mine;   // inclusive prefix sum over the wave
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_cnt[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_base = tot ? atomicAdd(unc_count, tot) : 0;
    }
    __syncthreads();
    if (mine) {
        int off = s_base + incl - mine;
        for (int w = 0; w < wave; ++w) off += s_cnt[w];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((unc >> k) & 1u) unc_list[off++] = (int)(p0 + k);
    }
}

constexpr int FIX_UL = 16;
__global__ void __launch_bounds__(256) k_argmax_exact_fix(const float *__restrict__ zvol, const float *__restrict__ score, int Z, int Y,
                                                          int X, Taps taps, const int *__restrict__ unc_list,
                                                          const int *__restrict__ unc_count, int *__restrict__ best_z, const ClipInfo *clip)
{
    const float dabs = cert_abs(clip);
    __shared__ float c[256];
    __shared__ float sfast[64];
    __shared__ float sexact[64];
    __shared__ int s_z;
    const int r = taps.n >> 1;  // 120
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  The list is in raster
    // order and neighbouring pixels read almost the same window of the volume, so every XCD gets one contiguous eighth
    // of the list instead of every eighth entry.
    const int total = *unc_count, xcd = blockIdx.x & 7, per = (total + 7) / 8;
    for (int sl = blockIdx.x >> 3; sl < per; sl += gridDim.x >> 3) {
        const int u = xcd * per + sl;
        if (u >= total) break;
        const int p = unc_list[u];
        const int y = p / X, x = p - y * X;
        const long P = (long)Y * X;
        if (threadIdx.x < Z) sfast[threadIdx.x] = score[(long)threadIdx.x * P + p];
        __syncthreads();
        float smax = sfast[0];
        for (int z = 1; z < Z; ++z) smax = fmaxf(smax, sfast[z]);
        const float thr = smax * (1.f - CERT_EPS) - dabs;
        for (int z = 0; z < Z; ++z) {
            const bool cand = sfast[z] * (1.f + CERT_EPS) + dabs >= thr;  // block-uniform
            if (!cand) { if (threadIdx.x == 0) sexact[z] = -1.f; continue; }
            const float *vol = zvol + (long)z * P;
            if (threadIdx.x < 2 * r + 1) {
                const int xx = clampi(x + (int)threadIdx.x - r, 0, X - 1);
                double tmp = (double)vol[(long)y * X + xx] * taps.w[r];
                // the sum is serial (scipy's order) but the loads are not: 2 * FIX_UL per trip in flight (the kernel is bound by
                // global-load latency, not by arithmetic)
                for (int d0 = r; d0 >= 1; d0 -= FIX_UL) {
                    float a[FIX_UL], b[FIX_UL];
#pragma unroll
                    for (int u = 0; u < FIX_UL; ++u) {
                        const int d = max(d0 - u, 1);
                        a[u] = vol[(long)clampi(y - d, 0, Y - 1) * X + xx];
                        b[u] = vol[(long)clampi(y + d, 0, Y - 1) * X + xx];
                    }
#pragma unroll
                    for (int u = 0; u < FIX_UL; ++u)
                        if (d0 - u >= 1) tmp += ((double)a[u] + (double)b[u]) * taps.w[r - (d0 - u)];
                }
                c[threadIdx.x] = (float)tmp;   // exact y pass, rounded to float32 like scipy's intermediate
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                double tmp = (double)c[r] * taps.w[r];
                for (int d = r; d >= 1; --d) tmp += ((double)c[r - d] + (double)c[r + d]) * taps.w[r - d];
                sexact[z] = (float)tmp;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            float b = -2.f; int bz = 0;
            for (int z = 0; z < Z; ++z)
                if (sexact[z] > b) { b = sexact[z]; bz = z; }
            best_z[p] = bz;
        }
        __syncthreads();
    }
}

// writes the z-maps from the certified plane index (same outputs as k_argmax_z)
__global__ void __launch_bounds__(256) k_emit_zmaps(const int *__restrict__ best_z, int Z, long P, int min_z, int atoh_shift,
                                                    int32_t *__restrict__ zsel, int32_t *__restrict__ zsel_atoh,
                                                    int64_t *__restrict__ zmap, int *__restrict__ err)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int cz = min_z + best_z[p];
    if (zmap) zmap[p] = cz;
    int ca = cz;
    if (atoh_shift != 0) { ca = cz + atoh_shift; ca = ca < 0 ? 0 : (ca > Z ? Z : ca); }
    if (cz >= Z || ca >= Z) {
        atomicOr(err, 1);
        zsel[p] = cz >= Z ? Z - 1 : cz;
        zsel_atoh[p] = ca >= Z ? Z - 1 : ca;
        return;
    }
    zsel[p] = cz;
    zsel_atoh[p] = ca;
}

// build_manifold with bin_size > 1 (sp.py:56-65): the (Yb, Xb) plane map of the binned score goes back to the frame through
// skimage.transform.resize(order 1, mode 'reflect') -- for 2-D arrays the bilinear warp of _warps_cy: source coordinate
// a * i + b (a = n_in / n_out, b = a / 2 - 1 / 2) evaluated in float32, corners floor / ceil with numpy 'reflect' (mirror
// without the edge: index -1 -> 1), top = (1 - dc) v00 + dc v01, bottom likewise, (1 - dr) top + dr bottom in double, float32
// result -- and np.round (half to even).  The float result agrees with skimage's to ~1e-6 (upstream's affine matrix comes out
// of a least-squares estimate, LAPACK-dependent in the last bit); the rounded maps equal the reference's on every golden,
// exact .5 ties included.  The atoh map is clip(plane + shift, 0, Z) BEFORE the resize, as upstream.
__device__ __forceinline__ int mirror_index(int n, int c)
{
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    c = (c < 0 ? -c : c) % p;
    return c > n - 1 ? p - c : c;
}
__global__ void __launch_bounds__(256) k_resize_round_zmaps(const int *__restrict__ bz, int Yb, int Xb, int Y, int X, int Z, int atoh_shift,
                                                            int32_t *__restrict__ zsel, int32_t *__restrict__ zsel_atoh,
                                                            int64_t *__restrict__ zmap, int *__restrict__ err)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    const double sy = (double)Yb / Y, sx = (double)Xb / X;
    const float ar = (float)sy, br = (float)(0.5 * sy - 0.5), ac = (float)sx, bc = (float)(0.5 * sx - 0.5);
    const float fr = ar * (float)y + br, fc = ac * (float)x + bc;
    const int r0 = (int)floorf(fr), c0 = (int)floorf(fc), r1 = (int)ceilf(fr), c1 = (int)ceilf(fc);
    const double dr = (double)(fr - (float)r0), dc = (double)(fc - (float)c0);
    const int y0 = mirror_index(Yb, r0), y1 = mirror_index(Yb, r1), x0 = mirror_index(Xb, c0), x1 = mirror_index(Xb, c1);
    int res[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        auto at = [&](int yy, int xx) -> double {
            int v = bz[(long)yy * Xb + xx];
            if (k == 1 && atoh_shift != 0) { v += atoh_shift; v = v < 0 ? 0 : (v > Z ? Z : v); }
            return (double)v;
        };
        const double top = (1.0 - dc) * at(y0, x0) + dc * at(y0, x1);
        const double bot = (1.0 - dc) * at(y1, x0) + dc * at(y1, x1);
        res[k] = (int)rintf((float)((1.0 - dr) * top + dr * bot));      // np.round: half to even (default rounding mode)
    }
    const long p = (long)y * X + x;
    if (zmap) zmap[p] = res[0];
    if (res[0] >= Z || res[1] >= Z) atomicOr(err, 1);
    zsel[p] = res[0] >= Z ? Z - 1 : res[0];
    zsel_atoh[p] = res[1] >= Z ? Z - 1 : res[1];
}

// ---- P4': bin_size > 1 (sp.py:39-65) -------------------------------------------------------------------------------
// skimage.measure.block_reduce(vol, (1, b, b), np.mean / np.var) in float32 with numpy's summation order: every row of
// a block (b contiguous samples, zeros beyond the frame) goes through numpy's pairwise_sum -- a running sum below 8
// elements, else eight running partial sums combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus the tail -- and the row
// sums are added up one after the other; mean = sum / float32(b*b); var = the same reduction of (x - mean)^2.
__device__ __forceinline__ float pw_row_sum(const float *__restrict__ row, int b, int valid, float mean, bool sq)
{
    auto at = [&](int i) -> float {
        const float v = i < valid ? row[i] : 0.f;
        if (!sq) return v;
        const float d = v - mean;
        return d * d;
    };
    if (b < 8) {
        float res = at(0);
        for (int i = 1; i < b; ++i) res += at(i);
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = at(j);
    int i = 8;
    for (; i < b - (b % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += at(i + j);
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < b; ++i) res += at(i);
    return res;
}

template <bool VAR>
__global__ void __launch_bounds__(256) k_block_reduce(const float *__restrict__ vol, float *__restrict__ out, int Z, int Y, int X,
                                                      int b, int Yb, int Xb)
{
    const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (long)Z * Yb * Xb) return;
    const int xb = (int)(o % Xb), yb = (int)((o / Xb) % Yb), z = (int)(o / ((long)Xb * Yb));
    const int x0 = xb * b, y0 = yb * b;
    const int valid = min(b, X - x0);
    const float *base = vol + ((long)z * Y + y0) * X + x0;
    const float cnt = (float)(b * b);
    float acc = 0.f;
    for (int r = 0; r < b; ++r) {
        const float row = pw_row_sum(base + (long)r * X, b, y0 + r < Y ? valid : 0, 0.f, false);
        acc = r == 0 ? row : acc + row;
    }
    const float mean = acc / cnt;
    if (!VAR) { out[o] = mean; return; }
    for (int r = 0; r < b; ++r) {
        const float row = pw_row_sum(base + (long)r * X, b, y0 + r < Y ? valid : 0, mean, true);
        acc = r == 0 ? row : acc + row;
    }
    out[o] = acc / cnt;
}

__global__ void __launch_bounds__(256) k_mul_f32(float *__restrict__ a, const float *__restrict__ b, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] * b[i];
}

// skimage.transform.resize(score, (Z, Y, X)) (order 1, mode 'reflect' -> scipy map_coordinates 'mirror'; the z factor is
// 1) fused with the first-maximum argmax over z.  One axis: coordinate f * (i + 0.5) - 0.5 in float64 with f = n_in /
// n_out, mirrored at both ends, weights (1 - t, 1 - (1 - t)); scipy adds the four corner terms (v * wy) * wx in float64 in
// the order (y0,x0), (y0,x1), (y1,x0), (y1,x1) and rounds to float32.
struct LinTap { int i0, i1; double w0, w1; };
__device__ __forceinline__ LinTap lin_tap(int i, int n_in, int n_out)
{
    LinTap t;
    if (n_in <= 1) { t.i0 = 0; t.i1 = 0; t.w0 = 1.0; t.w1 = 0.0; return t; }
    const double f = (double)n_in / (double)n_out;
    double c = f * ((double)i + 0.5) - 0.5;
    if (c < 0.0) c = -c;
    const double fl = floor(c);
    t.i0 = (int)fl;
    t.i1 = t.i0 + 1;
    if (t.i1 >= n_in) t.i1 = 2 * n_in - 2 - t.i1;
    t.w0 = 1.0 - (c - fl);
    t.w1 = 1.0 - t.w0;
    return t;
}

__global__ void __launch_bounds__(256) k_resize_argmax(const float *__restrict__ binned, int Z, int Yb, int Xb, int Y, int X,
                                                       int *__restrict__ best_z)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    const LinTap ty = lin_tap(y, Yb, Y), tx = lin_tap(x, Xb, X);
    float best = 0.f;
    int bi = 0;
    for (int z = 0; z < Z; ++z) {
        const float *pl = binned + (long)z * Yb * Xb;
        const double v00 = pl[(long)ty.i0 * Xb + tx.i0], v01 = pl[(long)ty.i0 * Xb + tx.i1];
        const double v10 = pl[(long)ty.i1 * Xb + tx.i0], v11 = pl[(long)ty.i1 * Xb + tx.i1];
        double t = (v00 * ty.w0) * tx.w0;
        t += (v01 * ty.w0) * tx.w1;
        t += (v10 * ty.w1) * tx.w0;
        t += (v11 * ty.w1) * tx.w1;
        const float s = (float)t;
        if (z == 0 || s > best) { best = s; bi = z; }
    }
    best_z[(long)y * X + x] = bi;
}

static int resolve_taps(const double *given, double sigma, int expect, Taps &t)
{
    double buf[256];
    if (!given) {
        int n = libm_taps(sigma, 4.0, buf, 255);
        if (n != expect) return fail(TIP_ERR_ARG, "internal: tap count %d != %d", n, expect);
        given = buf;
    }
    return make_taps(t, given, expect);
}

// build_continues_manifold on a device score (Zs, Y, X) -> plane index per pixel (int32, device); err bit 2: a pixel without
// a visited neighbour
static int manifold_dev(const float *score, int Zs, int Y, int X, int *bestz, int *err)
{
    Ctx &c = ctx();
    const long P = (long)Y * X, V = (long)Zs * P;
    if (V >= 0xffffffffL || Zs > 144 || Zs < 1) return fail(TIP_ERR_UNSUPPORTED, "build_manifold: stack too large");
    WsGuard ws;
    unsigned long long *key = ws.get<unsigned long long>(1);
    if (!key) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemsetAsync(key, 0, sizeof(unsigned long long), c.stream));
    TIP_HIP(hipMemsetAsync(bestz, 0xff, P * sizeof(int), c.stream));
    TIP_LAUNCH("argmax_first", k_argmax_first_f32, dim3((unsigned)std::min<long>(1024, cdiv(V, 256))), dim3(256), 0, score, V, key);
    int chunk = 1024;
    while (chunk > 64 && (size_t)4 * Zs * chunk > 150000) chunk >>= 1;
    const size_t lds = (size_t)4 * Zs * chunk;
    TIP_HIP(hipFuncSetAttribute((const void *)k_manifold, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    TIP_LAUNCH("manifold", k_manifold, dim3(1), dim3(MAN_T), lds, score, Zs, Y, X, (const unsigned long long *)key, (volatile int *)bestz,
               chunk, err);
    return TIP_OK;
}

int project_dev(const uint16_t *czyx, int C, int Z, int Y, int X, int zlo, int zhi, int min_z, int ref_ch, int method, int bin,
                int airyscan, int atoh_shift, const double *t05, const double *t1, const double *t2,
                const double *t30, double *proj, int64_t *zmap, const unsigned long long *hist_in = nullptr)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!czyx || !proj) return fail(TIP_ERR_ARG, "project: null pointer");
    if (C < 1 || C > 8) return fail(TIP_ERR_ARG, "project: 1..8 channels supported (got %d)", C);
    if (ref_ch < 0 || ref_ch >= C) return fail(TIP_ERR_INDEX, "project: reference_channel %d out of range", ref_ch);
    if (zlo < 0 || zhi > Z || zhi <= zlo) return fail(TIP_ERR_ARG, "project: bad z range [%d,%d) of %d", zlo, zhi, Z);
    if (Y < 1 || X < 1 || Y > 65535) return fail(TIP_ERR_ARG, "project: bad frame size %dx%d", Y, X);
    const int Zs = zhi - zlo;
    const long P = (long)Y * X, V = (long)Zs * P;
    const bool manifold = (method & TIP_PROJECT_MANIFOLD) != 0;      // sp.py:56-57: the spiral z-map instead of the argmax
    method &= ~TIP_PROJECT_MANIFOLD;
    if (manifold && hist_in) return fail(TIP_ERR_UNSUPPORTED, "project: build_manifold takes whole frames");
    if (manifold && (V >= 0xffffffffL || Zs > 144)) return fail(TIP_ERR_UNSUPPORTED, "project: build_manifold: stack too large");
    Taps k05, k1, k2, k30;
    int rc;
    if ((rc = resolve_taps(t05, 0.5, 5, k05)) || (rc = resolve_taps(t1, 1.0, 9, k1)) ||
        (rc = resolve_taps(t2, 2.0, 17, k2)) || (rc = resolve_taps(t30, 30.0, 241, k30)))
        return rc;

    WsGuard ws;
    float *A = ws.get<float>(V), *B = ws.get<float>(V);
    unsigned long long *hist = ws.get<unsigned long long>(65536);
    ClipInfo *clip = ws.get<ClipInfo>(1);
    int32_t *zsel = ws.get<int32_t>(P), *zsel_a = ws.get<int32_t>(P);
    float *ident = ws.get<float>((size_t)Zs * Zs), *table = ws.get<float>((size_t)Zs * Zs);
    int *err = ws.get<int>(1);
    int32_t *zrange = ws.get<int32_t>(P);
    if (!zrange) return TIP_ERR_NOMEM;
    if (!A || !B || !hist || !clip || !zsel || !zsel_a || !ident || !table || !err) return TIP_ERR_NOMEM;

    const uint16_t *ref = czyx + ((long)ref_ch * Z + zlo) * P;
    TIP_HIP(hipMemsetAsync(err, 0, sizeof(int), c.stream));
    if (hist_in) {   // a tile of a larger frame: the caller brings the whole frame's histogram
        if (bin > 1 && method == 2) return fail(TIP_ERR_UNSUPPORTED, "project: tiles with method multi_channel");
        TIP_LAUNCH("percentile95", k_percentile95, dim3(1), dim3(1024), 0, hist_in, clip, 0);
    } else {
        TIP_HIP(hipMemsetAsync(hist, 0, 65536 * sizeof(unsigned long long), c.stream));
        TIP_LAUNCH("hist_u16", k_hist_u16, dim3((unsigned)std::min<long>(cdiv(V, HIST_PER_BLOCK), cu_count())), dim3(1024), 0, ref, V, airyscan, hist);
        TIP_LAUNCH("percentile95", k_percentile95, dim3(1), dim3(1024), 0, (const unsigned long long *)hist, clip, 0);
    }

    const bool fast = (X % 4 == 0) && !tuning().project_generic;
    // P3: (0.5, 1, 1) of the clipped channel -> A_, then P4's z pass (0.5) of that -> B_
    auto short_blur = [&](const uint16_t *src, ClipInfo *ci, float *A_, float *B_) -> int {
        if (fast) {   // register-sliding kernels (each input loaded once per thread)
            Src4U16Clip su{src, airyscan, &ci->p95, &ci->has};
            TIP_LAUNCH("zpass_u16clip_x4", (k_zpass_r2_x4<Src4U16Clip>), dim3(cdiv(P / 4, 256)), dim3(256), 0, su, A_, Zs, P, k05);
            // (36-row segments; 72-row segments halve the re-read halo rows but measured slower -- 0.257 against 0.235 ms --
            // and so did four columns per thread with 16-byte accesses: the pass wants many independent row streams)
            TIP_LAUNCH("ypass_slide_r4", (k_ypass_slide<float, 4, 4>), dim3(cdiv(X, 256), cdiv(Y, 36), Zs), dim3(256), 0,
                       (const float *)A_, B_, Y, X, k1);
            TIP_LAUNCH("xpass_slide_r4", (k_xpass_slide<float, 4>), dim3(cdiv(cdiv(X, 8), 256), Y, Zs), dim3(256), 0, (const float *)B_,
                       A_, Y, X, k1);
            Src4F32 sf{A_};
            TIP_LAUNCH("zpass_f32_x4", (k_zpass_r2_x4<Src4F32>), dim3(cdiv(P / 4, 256)), dim3(256), 0, sf, B_, Zs, P, k05);
            return TIP_OK;
        }
        LoadU16Clip ld{src, P, (long)X, airyscan, ci};
        dim3 grid(cdiv(X, 256), Y, Zs), block(256);
        TIP_LAUNCH("corr_z_u16clip", (k_corr_generic<float, 0, LoadU16Clip>), grid, block, 0, ld, A_, Zs, Y, X, k05);
        int r2;
        if ((r2 = correlate1d_dev(A_, B_, 0, Zs, Y, X, 1, k1, 0))) return r2;
        if ((r2 = correlate1d_dev(B_, A_, 0, Zs, Y, X, 2, k1, 0))) return r2;
        return correlate1d_dev(A_, B_, 0, Zs, Y, X, 0, k05, 0);
    };
    // bin_size == 1 only needs the z-passed blur (B): the four short passes run as one kernel (tip_preblur.h)
    const bool fused_pre = fast && bin == 1 && !tuning().project_unfused_preblur;
    if (fused_pre) {
        ShortTaps s05, s1;
        for (int i = 0; i < 8; ++i) { s05.w[i] = i < 3 ? k05.w[i] : 0.0; s1.w[i] = i < 5 ? k1.w[i] : 0.0; }
        TIP_LAUNCH("preblur_fused", k_preblur_fused, dim3(cdiv(X, PB_X), cdiv(Y, PB_Y)), dim3(PB_T), 0, ref, airyscan,
                   (const float *)&clip->p95, (const int *)&clip->has, B, Zs, Y, X, s05, s1);
    } else if ((rc = short_blur(ref, clip, A, B))) return rc;
    if (bin > 1) {
        // P4' (sp.py:39-65): score on bin x bin blocks, resized back to the frame for the argmax
        const int Yb = cdiv(Y, bin), Xb = cdiv(X, bin);
        const long nb = (long)Zs * Yb * Xb;
        float *S1 = ws.get<float>(nb);
        int *bestz = ws.get<int>(P);
        if (!S1 || !bestz) return TIP_ERR_NOMEM;
        if (method == 0) {          // max_averages: block mean of the (exact) (0.5, 30, 30) blur
            if ((rc = correlate1d_dev(B, A, 0, Zs, Y, X, 1, k30, 0))) return rc;
            if ((rc = correlate1d_dev(A, B, 0, Zs, Y, X, 2, k30, 0))) return rc;
            TIP_LAUNCH("block_mean", (k_block_reduce<false>), dim3(cdiv(nb, 256)), dim3(256), 0, (const float *)B, S1, Zs, Y, X, bin, Yb, Xb);
        } else {                    // max_std / multi_channel: block variance of the (0.5, 1, 1) blur
            TIP_LAUNCH("block_var", (k_block_reduce<true>), dim3(cdiv(nb, 256)), dim3(256), 0, (const float *)A, S1, Zs, Y, X, bin, Yb, Xb);
            if (method == 2) {      // times the block mean of the next channel's (0.5, 30, 30) blur (sp.py:45-51)
                const uint16_t *other = czyx + ((long)((ref_ch + 1) % C) * Z + zlo) * P;
                float *S2 = ws.get<float>(nb);
                ClipInfo *clip2 = ws.get<ClipInfo>(1);
                if (!S2 || !clip2) return TIP_ERR_NOMEM;
                TIP_HIP(hipMemsetAsync(hist, 0, 65536 * sizeof(unsigned long long), c.stream));
                TIP_LAUNCH("hist_u16", k_hist_u16, dim3((unsigned)std::min<long>(cdiv(V, HIST_PER_BLOCK), cu_count())), dim3(1024), 0, other, V, airyscan, hist);
                TIP_LAUNCH("percentile95", k_percentile95, dim3(1), dim3(1024), 0, (const unsigned long long *)hist, clip2, 1);
                if ((rc = short_blur(other, clip2, A, B))) return rc;
                if ((rc = correlate1d_dev(B, A, 0, Zs, Y, X, 1, k30, 0))) return rc;
                if ((rc = correlate1d_dev(A, B, 0, Zs, Y, X, 2, k30, 0))) return rc;
                TIP_LAUNCH("block_mean", (k_block_reduce<false>), dim3(cdiv(nb, 256)), dim3(256), 0, (const float *)B, S2, Zs, Y, X, bin, Yb, Xb);
                TIP_LAUNCH("mul_f32", k_mul_f32, dim3(cdiv(nb, 256)), dim3(256), 0, S1, (const float *)S2, nb);
            }
        }
        if (manifold) {
            // sp.py:56-57, 63-65: the spiral on the BINNED score, then the plane maps resized to the frame and rounded
            int *bz = ws.get<int>((size_t)Yb * Xb);
            if (!bz) return TIP_ERR_NOMEM;
            if ((rc = manifold_dev((const float *)S1, Zs, Yb, Xb, bz, err))) return rc;
            TIP_LAUNCH("resize_round_zmaps", k_resize_round_zmaps, dim3(cdiv(X, 256), Y), dim3(256), 0, (const int *)bz, Yb, Xb, Y, X, Zs,
                       atoh_shift, zsel, zsel_a, zmap, err);
        } else {
        TIP_LAUNCH("resize_argmax", k_resize_argmax, dim3(cdiv(X, 256), Y), dim3(256), 0, (const float *)S1, Zs, Yb, Xb, Y, X, bestz);
        TIP_LAUNCH("emit_zmaps", k_emit_zmaps, dim3(cdiv(P, 256)), dim3(256), 0, (const int *)bestz, Zs, P, min_z, atoh_shift, zsel,
                   zsel_a, zmap, err);
        }
    } else {
    // P4 + P5: (0.5, 30, 30) score and its argmax
    const bool certified = fast && Zs <= 64 && !manifold && !tuning().project_exact_score;
    if (manifold) {
        // the spiral reads score VALUES (window argmaxes), so it gets the exact score; min_z is not added (sp.py:57)
        if ((rc = correlate1d_dev(B, A, 0, Zs, Y, X, 1, k30, 0))) return rc;
        if ((rc = correlate1d_dev(A, B, 0, Zs, Y, X, 2, k30, 0))) return rc;
        int *bestz = ws.get<int>(P);
        if (!bestz) return TIP_ERR_NOMEM;
        if ((rc = manifold_dev((const float *)B, Zs, Y, X, bestz, err))) return rc;
        TIP_LAUNCH("emit_zmaps", k_emit_zmaps, dim3(cdiv(P, 256)), dim3(256), 0, (const int *)bestz, Zs, P, 0, atoh_shift, zsel, zsel_a,
                   zmap, err);
    } else if (certified) {
        float *D = ws.get<float>(V);
        int *bestz = ws.get<int>(P), *unc = ws.get<int>(P), *uncn = ws.get<int>(1);
        if (!D || !bestz || !unc || !uncn) return TIP_ERR_NOMEM;
        TapsF f30;
        f30.n = k30.n;
        for (int i = 0; i < 256; ++i) f30.w[i] = (float)k30.w[i];
        const int r = k30.n >> 1;
        {   // fast y pass B -> A, fast x pass A -> D   (B, the exact z-passed volume, is kept for the exact fix-up)
            int cy = FAST_CFG_Y, cx = FAST_CFG_X;
            if (tuning().fast_cfg_y >= 0) { cy = tuning().fast_cfg_y; cx = tuning().fast_cfg_x >= 0 ? tuning().fast_cfg_x : cx; }  // tuning hook TIP_FAST_CFG: NW*100+NP per pass
            // (the fp16 tiles, cfg 5, take the clip value: their samples are scaled into fp16's range by it, and the volume B --
            //  convex combinations of clipped voxels -- is bounded by it; err bit 8 would report a sample beyond that range)
            if ((rc = launch_fast<1>(cy, (const float *)B, A, Zs, Y, X, f30, clip, err))) return rc;
            if ((rc = launch_fast<2>(cx, (const float *)A, D, Zs, Y, X, f30, clip, err))) return rc;
        }
        TIP_HIP(hipMemsetAsync(uncn, 0, sizeof(int), c.stream));
        TIP_LAUNCH("argmax_certify", k_argmax_certify, dim3(cdiv(cdiv(P, 4), 256)), dim3(256), 0, (const float *)D, Zs, P, bestz, unc, uncn,
                   (const ClipInfo *)clip);
        TIP_LAUNCH("argmax_exact_fix", k_argmax_exact_fix, dim3(8192), dim3(256), 0, (const float *)B, (const float *)D, Zs, Y, X,
                   k30, (const int *)unc, (const int *)uncn, bestz, (const ClipInfo *)clip);
        TIP_LAUNCH("emit_zmaps", k_emit_zmaps, dim3(cdiv(P, 256)), dim3(256), 0, (const int *)bestz, Zs, P, min_z, atoh_shift, zsel,
                   zsel_a, zmap, err);
        if (tuning().project_debug) {
            int hn = 0, he = 0;
            TIP_HIP(hipMemcpyAsync(&hn, uncn, sizeof(int), hipMemcpyDeviceToHost, c.stream));
            TIP_HIP(hipMemcpyAsync(&he, err, sizeof(int), hipMemcpyDeviceToHost, c.stream));
            TIP_HIP(hipStreamSynchronize(c.stream));
            fprintf(stderr, "certified argmax: %d of %ld pixels recomputed exactly\n", hn, P);
            if (he & 8) return fail(TIP_ERR_HIP, "score pass: a sample beyond the clip value's range reached the fp16 tiles");
        }
    } else {
        if ((rc = correlate1d_dev(B, A, 0, Zs, Y, X, 1, k30, 0))) return rc;
        if ((rc = correlate1d_dev(A, B, 0, Zs, Y, X, 2, k30, 0))) return rc;
        // P5
        TIP_LAUNCH("argmax_z", k_argmax_z, dim3(cdiv(P, 256)), dim3(256), 0, B, Zs, P, min_z, atoh_shift, zsel, zsel_a, zmap,
                   err);
    }
    }   // bin == 1
    // P6/P7 z pass as a Zs x Zs table (sigma 1 -> 9 taps), built with the same correlate kernel
    TIP_LAUNCH("identity", k_identity, dim3(cdiv((long)Zs * Zs, 256)), dim3(256), 0, ident, Zs);
    if ((rc = correlate1d_dev(ident, table, 0, Zs, Zs, 1, 0, k1, 1))) return rc;

    const unsigned all = (1u << C) - 1u;
    for (int pass = 0; pass < (atoh_shift != 0 ? 2 : 1); ++pass) {
        const int32_t *sel = pass == 0 ? zsel : zsel_a;
        unsigned cm = atoh_shift == 0 ? all : (pass == 0 ? (1u << ref_ch) : (all & ~(1u << ref_ch)));
        if (!cm) continue;
        if (fast && Zs <= 64 && !tuning().project_unfused_mask) {
            // (instantiated per channel count: the per-channel running maxima are registers, 8 channels' worth of them cost
            //  the two-channel case a third of its occupancy)
            const dim3 fgrid(cdiv(X, FT_X), cdiv(Y, FT_Y));
            const size_t flds = (size_t)Zs * Zs * sizeof(float);
            if (C <= 2) {
                TIP_LAUNCH("mask_wmax_fused", (k_mask_wmax_fused<2>), fgrid, dim3(256), flds, (const float *)table, sel, czyx, C, Z, zlo, Zs,
                           Y, X, airyscan, cm, k2, proj);
            } else if (C <= 4) {
                TIP_LAUNCH("mask_wmax_fused", (k_mask_wmax_fused<4>), fgrid, dim3(256), flds, (const float *)table, sel, czyx, C, Z, zlo, Zs,
                           Y, X, airyscan, cm, k2, proj);
            } else {
                TIP_LAUNCH("mask_wmax_fused", (k_mask_wmax_fused<8>), fgrid, dim3(256), flds, (const float *)table, sel, czyx, C, Z, zlo, Zs,
                           Y, X, airyscan, cm, k2, proj);
            }
        } else if (fast && Zs <= 64 && (long)Zs * Y < 2147483647L) {
            TIP_LAUNCH("mask_y_sparse", (k_mask_y_sparse<2>), dim3(cdiv(X, 256), cdiv(Y, 2 * MASK_W)), dim3(256),
                       (size_t)Zs * Zs * sizeof(float), (const float *)table, sel, Zs, Y, X, k2, A, zrange);
            TIP_LAUNCH("xpass_wmax_sparse", (k_xpass_wmax_sparse<8>), dim3(cdiv(cdiv(X, 8), 256), Y), dim3(256), 0, (const float *)A,
                       (const int32_t *)zrange, czyx, C, Z, zlo, Zs, Y, X, airyscan, cm, k2, proj);
        } else {
            LoadMaskTable ld{table, sel, Zs, (long)X};
            dim3 grid(cdiv(X, 256), Y, Zs), block(256);
            TIP_LAUNCH("mask_ypass", (k_corr_generic<float, 1, LoadMaskTable>), grid, block, 0, ld, A, Zs, Y, X, k2);
            TIP_LAUNCH("xpass_wmax", (k_xpass_wmax<8>), dim3(cdiv(X, 256), Y), dim3(256), 0, A, czyx, C, Z, zlo, Zs, Y, X,
                       airyscan, cm, k2, proj);
        }
    }
    if (min_z > 0 || atoh_shift > 0 || manifold) {
        int h = 0;
        TIP_HIP(hipMemcpyAsync(&h, err, sizeof(int), hipMemcpyDeviceToHost, c.stream));
        TIP_HIP(hipStreamSynchronize(c.stream));
        if (h & 2) return fail(TIP_ERR_HIP, "build_manifold: a pixel without a visited neighbour (upstream raises TypeError here)");
        if (h & 8) return fail(TIP_ERR_HIP, "score pass: a sample beyond the clip value's range reached the fp16 tiles");
        if (h)
            return fail(TIP_ERR_INDEX, "chosen z index out of bounds for the %d-plane mask (min_z=%d, atoh_shift=%d): "
                                       "the reference raises IndexError here (sp.py:62,68-69)", Zs, min_z, atoh_shift);
    }
    return TIP_OK;
}

}  // namespace tip

using namespace tip;

extern "C" {

static int project_host(const uint16_t *czyx, int c, int z, int y, int x, int zlo, int zhi, int min_z, int ref_ch, int method,
                        int bin, int airyscan, int atoh_shift, const double *t05, const double *t1, const double *t2,
                        const double *t30, double *proj, int64_t *zmap)
{
    Ctx &cx = ctx();
    if (!cx.stream) return TIP_ERR_HIP;
    if (!czyx || !proj) return fail(TIP_ERR_ARG, "tip_project_u16: null pointer");
    if (c < 1 || z < 1 || y < 1 || x < 1) return fail(TIP_ERR_ARG, "tip_project_u16: empty stack");
    const size_t nin = (size_t)c * z * y * x, P = (size_t)y * x;
    WsGuard ws;
    uint16_t *din = ws.get<uint16_t>(nin);
    double *dproj = ws.get<double>((size_t)c * P);
    int64_t *dz = ws.get<int64_t>(P);
    if (!din || !dproj || !dz) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(din, czyx, nin * 2, hipMemcpyHostToDevice, cx.stream));
    int rc = project_dev(din, c, z, y, x, zlo, zhi, min_z, ref_ch, method, bin, airyscan, atoh_shift, t05, t1, t2, t30, dproj, dz);
    if (rc) return rc;
    TIP_HIP(hipMemcpyAsync(proj, dproj, (size_t)c * P * 8, hipMemcpyDeviceToHost, cx.stream));
    if (zmap) TIP_HIP(hipMemcpyAsync(zmap, dz, P * 8, hipMemcpyDeviceToHost, cx.stream));
    TIP_HIP(hipStreamSynchronize(cx.stream));
    return TIP_OK;
}

static int check_binned(int method, int bin)
{
    if (method >= 0 && (method & TIP_PROJECT_MANIFOLD)) {
        method &= ~TIP_PROJECT_MANIFOLD;
    }
    if (method < 0 || method > 2) return fail(TIP_ERR_ARG, "projection: method %d (0 max_averages, 1 max_std, 2 multi_channel)", method);
    if (bin < 1 || bin > 128) return fail(TIP_ERR_UNSUPPORTED, "projection: bin_size %d (1..128 supported)", bin);
    return TIP_OK;
}

int tip_project_u16_dev(const uint16_t *czyx, int c, int z, int y, int x, int zlo, int zhi, int min_z, int ref_ch,
                        int airyscan, int atoh_shift, const double *t05, const double *t1, const double *t2,
                        const double *t30, double *proj, int64_t *zmap)
{
    return project_dev(czyx, c, z, y, x, zlo, zhi, min_z, ref_ch, 0, 1, airyscan, atoh_shift, t05, t1, t2, t30, proj, zmap);
}

int tip_project_u16(const uint16_t *czyx, int c, int z, int y, int x, int zlo, int zhi, int min_z, int ref_ch,
                    int airyscan, int atoh_shift, const double *t05, const double *t1, const double *t2,
                    const double *t30, double *proj, int64_t *zmap)
{
    return project_host(czyx, c, z, y, x, zlo, zhi, min_z, ref_ch, 0, 1, airyscan, atoh_shift, t05, t1, t2, t30, proj, zmap);
}

int tip_hist_u16_box_dev(const uint16_t *czyx, int c, int z, int y, int x, int ch, int z0, int z1, int y0, int y1, int x0, int x1,
                         int airyscan, unsigned long long *hist_dev)
{
    Ctx &cx = ctx();
    if (!cx.stream) return TIP_ERR_HIP;
    if (!czyx || !hist_dev) return fail(TIP_ERR_ARG, "tip_hist_u16_box_dev: null pointer");
    if (ch < 0 || ch >= c || z0 < 0 || z1 > z || z1 <= z0 || y0 < 0 || y1 > y || y1 <= y0 || x0 < 0 || x1 > x || x1 <= x0)
        return fail(TIP_ERR_ARG, "tip_hist_u16_box_dev: box outside the (%d,%d,%d,%d) stack", c, z, y, x);
    const long n = (long)(z1 - z0) * (y1 - y0) * (x1 - x0);
    const uint16_t *plane = czyx + (long)ch * z * y * x;
    TIP_LAUNCH("hist_u16_box", k_hist_u16_box, dim3(cdiv(n, HIST_PER_BLOCK)), dim3(1024), 0, plane, y, x, z0, y0, x0, z1 - z0, y1 - y0,
               x1 - x0, airyscan, hist_dev);
    return TIP_OK;
}

int tip_project_u16_hist_dev(const uint16_t *czyx, int c, int z, int y, int x, int zlo, int zhi, int min_z, int ref_ch,
                             int airyscan, int atoh_shift, const double *t05, const double *t1, const double *t2,
                             const double *t30, const unsigned long long *hist_dev, double *proj, int64_t *zmap)
{
    if (!hist_dev) return fail(TIP_ERR_ARG, "tip_project_u16_hist_dev: null histogram");
    return project_dev(czyx, c, z, y, x, zlo, zhi, min_z, ref_ch, 0, 1, airyscan, atoh_shift, t05, t1, t2, t30, proj, zmap, hist_dev);
}

int tip_project_u16_binned_dev(const uint16_t *czyx, int c, int z, int y, int x, int zlo, int zhi, int min_z, int ref_ch,
                               int method, int bin_size, int airyscan, int atoh_shift, const double *t05, const double *t1,
                               const double *t2, const double *t30, double *proj, int64_t *zmap)
{
    int rc = check_binned(method, bin_size);
    if (rc) return rc;
    return project_dev(czyx, c, z, y, x, zlo, zhi, min_z, ref_ch, method, bin_size, airyscan, atoh_shift, t05, t1, t2, t30, proj, zmap);
}

int tip_project_u16_binned(const uint16_t *czyx, int c, int z, int y, int x, int zlo, int zhi, int min_z, int ref_ch,
                           int method, int bin_size, int airyscan, int atoh_shift, const double *t05, const double *t1,
                           const double *t2, const double *t30, double *proj, int64_t *zmap)
{
    int rc = check_binned(method, bin_size);
    if (rc) return rc;
    return project_host(czyx, c, z, y, x, zlo, zhi, min_z, ref_ch, method, bin_size, airyscan, atoh_shift, t05, t1, t2, t30, proj, zmap);
}

// sp.py:87-165 on its own: score float32 (z, y, x) host -> int64 (y, x) plane map
// ---- diagnostics of the fp16 score tiles (tests) ---------------------------------------------------------------------------------
__global__ void k_set_clip(ClipInfo *c, float v) { c->has = 1; c->p95 = v; c->p95d = (double)v; }

// one sigma-30 pass of tip_corr_f16.h on HOST float32 volumes (z, y, x): axis 1 = along y, 2 = along x; taps = 241 symmetric float64
// weights (radius 120); clip = the bound of the data (the scaling follows from it).  flag: bit 8 = a sample beyond the range.
int tip_score_pass_f16(const float *in, float *out, int z, int y, int x, int axis, const double *taps, int ntaps, float clip, int *flag)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!in || !out || !taps || !flag || z < 1 || y < 1 || x < 1 || (axis != 1 && axis != 2) || ntaps != 241 || !(clip > 0.f))
        return fail(TIP_ERR_ARG, "tip_score_pass_f16: bad arguments (radius 120 only)");
    const size_t V = (size_t)z * y * x;
    WsGuard ws;
    float *di = ws.get<float>(V), *dout = ws.get<float>(V);
    ClipInfo *ci = ws.get<ClipInfo>(1);
    int *df = ws.get<int>(1);
    if (!di || !dout || !ci || !df) return TIP_ERR_NOMEM;
    TapsF t;
    t.n = ntaps;
    for (int i = 0; i < 256; ++i) t.w[i] = i < ntaps ? (float)taps[i] : 0.f;
    TIP_HIP(hipMemcpyAsync(di, in, V * 4, hipMemcpyHostToDevice, c.stream));
    TIP_HIP(hipMemsetAsync(df, 0, sizeof(int), c.stream));
    hipLaunchKernelGGL(k_set_clip, dim3(1), dim3(1), 0, c.stream, ci, clip);
    int rc = axis == 1 ? launch_f16<1>(di, dout, z, y, x, t, ci, df) : launch_f16<2>(di, dout, z, y, x, t, ci, df);
    if (rc) return rc;
    TIP_HIP(hipMemcpyAsync(out, dout, V * 4, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(flag, df, sizeof(int), hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

// How v_mfma_f32_32x32x16_f16 rounds, on THIS device (the certified bound of tip_corr_f16.h counts two roundings per instruction:
// the sixteen products enter the accumulator as two exactly-summed halves).  Every row of A / column of B is the same; case t has
// C = c[t] and product k = a[t][k] b[t][k]; out[t] = any output element.
__global__ void k_mfma_f16_probe(const float *a, const float *b, const float *c, float *out, int ncase)
{
    const int h = threadIdx.x >> 5;
    for (int t = 0; t < ncase; ++t) {
        f16x8 fa, fb;
        for (int e = 0; e < 8; ++e) { fa[e] = (_Float16)a[t * 16 + 8 * h + e]; fb[e] = (_Float16)b[t * 16 + 8 * h + e]; }
        f32x16 acc;
        for (int q = 0; q < 16; ++q) acc[q] = c[t];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
        if (threadIdx.x == 0) out[t] = acc[0];
    }
}
int tip_mfma_f16_probe(const float *a, const float *b, const float *cvals, float *out, int ncase)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!a || !b || !cvals || !out || ncase < 1 || ncase > 256) return fail(TIP_ERR_ARG, "tip_mfma_f16_probe: bad arguments");
    WsGuard ws;
    float *da = ws.get<float>((size_t)ncase * 16), *db = ws.get<float>((size_t)ncase * 16), *dc = ws.get<float>(ncase), *dout = ws.get<float>(ncase);
    if (!da || !db || !dc || !dout) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(da, a, (size_t)ncase * 64, hipMemcpyHostToDevice, c.stream));
    TIP_HIP(hipMemcpyAsync(db, b, (size_t)ncase * 64, hipMemcpyHostToDevice, c.stream));
    TIP_HIP(hipMemcpyAsync(dc, cvals, (size_t)ncase * 4, hipMemcpyHostToDevice, c.stream));
    hipLaunchKernelGGL(k_mfma_f16_probe, dim3(1), dim3(64), 0, c.stream, (const float *)da, (const float *)db, (const float *)dc, dout, ncase);
    TIP_HIP(hipMemcpyAsync(out, dout, (size_t)ncase * 4, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_build_manifold_f32(const float *score, int z, int y, int x, int64_t *chosen)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!score || !chosen || z < 1 || y < 1 || x < 1) return fail(TIP_ERR_ARG, "tip_build_manifold_f32: bad arguments");
    const size_t P = (size_t)y * x, V = P * z;
    WsGuard ws;
    float *ds = ws.get<float>(V);
    int *bz = ws.get<int>(P), *err = ws.get<int>(1);
    if (!ds || !bz || !err) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(ds, score, V * 4, hipMemcpyHostToDevice, c.stream));
    TIP_HIP(hipMemsetAsync(err, 0, sizeof(int), c.stream));
    int rc = manifold_dev(ds, z, y, x, bz, err);
    if (rc) return rc;
    std::vector<int> hb(P);
    int herr = 0;
    TIP_HIP(hipMemcpyAsync(hb.data(), bz, P * 4, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    if (herr) return fail(TIP_ERR_HIP, "build_manifold: a pixel without a visited neighbour (upstream raises TypeError here)");
    for (size_t i = 0; i < P; ++i) chosen[i] = hb[i];
    return TIP_OK;
}

}  // extern "C"
