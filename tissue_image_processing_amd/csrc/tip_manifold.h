// tip_manifold.h -- build_continues_manifold (sp.py:87-165): the z-map grown as a square spiral around the score's
// global maximum.  A pixel's plane comes from the planes of its (up to two) already-visited neighbours -- the argmax of the
// score in a 3-plane window around a single neighbour's plane (or two equal ones), in a 2-plane window when they are one
// plane apart, else their mean -- so the spiral is one long dependency chain: upstream walks it in Python, 4 M steps for
// a 2048^2 frame.
//
// Parallel form.  Inside one straight run of a ring (a ring = 5 runs: right edge lower half, bottom, left, top, right edge
// upper half) every pixel depends on constants (planes of inner rings / earlier runs, read from the map) and on ONE
// variable, the plane s of its predecessor in the run.  So each pixel is a FUNCTION F_i: s -> plane over the Z possible
// planes (a Z-entry table), a run is the composition F_n o ... o F_1 o F_0 with F_0 constant, and compositions are
// associative: a Hillis-Steele scan over the tables gives every pixel's plane in log2(run length) steps.  One workgroup
// walks the rings (they ARE sequential), scanning 1024 (512 when Z > 40) pixels at a time in LDS.
//
// Reproduced quirks of the reference (goldens pin them): the "up" neighbour of row 0 is the LAST row (Python's index -1),
// read like any other neighbour (it counts when it has been visited); "down" exists for row < R - 1, "left" for col > 0,
// "right" for col < C - 1; left / right are only consulted while fewer than two neighbours have been found; the mean of
// two planes is truncated (float stored into an integer array); the window argmax takes the first maximum.
#pragma once
#include "tip_internal.h"

namespace tip {

// first global maximum of a float32 array: key = (sortable value bits) << 32 | (~index): atomicMax keeps the largest
// value and, among equal values, the smallest index (np.argmax)
__global__ void __launch_bounds__(256) k_argmax_first_f32(const float *__restrict__ v, long n, unsigned long long *__restrict__ key)
{
    unsigned long long best = 0ULL;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned int b = __float_as_uint(v[i] + 0.0f);
        b = (b >> 31) ? ~b : (b | 0x80000000u);
        const unsigned long long k = ((unsigned long long)b << 32) | (unsigned long long)(0xffffffffu - (unsigned int)i);
        best = k > best ? k : best;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(best, d, 64);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(key, best);
}

constexpr int MAN_T = 1024;

__global__ void __launch_bounds__(MAN_T) k_manifold(const float *__restrict__ score, int Z, int R, int C,
                                                    const unsigned long long *__restrict__ start_key, volatile int *chosen,
                                                    int chunk, int *__restrict__ err)
{
    extern __shared__ unsigned char man_lds[];
    unsigned char *b3 = man_lds;                       // [Z][chunk] argmax plane of the 3-plane window around z
    unsigned char *b2 = b3 + (size_t)Z * chunk;        // [Z][chunk] argmax plane of the window [z, z + 2)
    unsigned char *ta = b2 + (size_t)Z * chunk;        // [Z][chunk] function tables, ping
    unsigned char *tb = ta + (size_t)Z * chunk;        //            pong
    const int tid = threadIdx.x;
    const long P = (long)R * C;
    const unsigned int sidx = 0xffffffffu - (unsigned int)(*start_key & 0xffffffffULL);
    const int sp = (int)(sidx / P), sr = (int)((sidx % P) / C), sc = (int)(sidx % C);
    if (tid == 0) chosen[(long)sr * C + sc] = sp;
    __threadfence();
    __syncthreads();
    const int dmax = max(max(sc, sr), max(C - 1 - sc, R - 1 - sr));
    for (int d = 1; d <= dmax; ++d) {
        for (int seg = 0; seg < 5; ++seg) {
            int r0, c0, dr = 0, dc = 0, n0;
            bool exists;
            switch (seg) {
            case 0: r0 = sr; c0 = sc + d; dr = 1; n0 = d + 1; exists = c0 < C; break;              // right edge, lower half
            case 1: r0 = sr + d; c0 = sc + d - 1; dc = -1; n0 = 2 * d; exists = r0 < R; break;     // bottom edge, leftwards
            case 2: r0 = sr + d - 1; c0 = sc - d; dr = -1; n0 = 2 * d; exists = c0 >= 0; break;    // left edge, upwards
            case 3: r0 = sr - d; c0 = sc - d + 1; dc = 1; n0 = 2 * d; exists = r0 >= 0; break;     // top edge, rightwards
            default: r0 = sr - d + 1; c0 = sc + d; dr = 1; n0 = d - 1; exists = c0 < C; break;     // right edge, upper half
            }
            if (!exists || n0 < 1) continue;
            // the steps k of the run that lie inside the image
            const int m0 = dr ? r0 : c0, st = dr ? dr : dc, lim = dr ? R : C;
            int k_lo, k_hi;
            if (st > 0) { k_lo = max(0, -m0); k_hi = min(n0 - 1, lim - 1 - m0); }
            else { k_lo = max(0, m0 - (lim - 1)); k_hi = min(n0 - 1, m0); }
            if (k_lo > k_hi) continue;
            // the left edge may end on row 0, whose "up" neighbour is the last row -- possibly the first pixel of this very
            // run: that pixel is evaluated on its own, after the rest of the run is in the map
            const int split = (seg == 2 && k_hi == r0 && k_lo < k_hi) ? k_hi : -1;
            for (int part = 0; part < (split >= 0 ? 2 : 1); ++part) {
                const int pa = part == 0 ? k_lo : split, pb = part == 0 ? (split >= 0 ? split - 1 : k_hi) : k_hi;
                for (int ks = pa; ks <= pb; ks += chunk) {
                    const int cnt = min(chunk, pb - ks + 1);
                    // ---- per pixel: window tables of its score column, neighbours, function table ----------------------
                    if (tid < cnt) {
                        const int k = ks + tid, r = r0 + k * dr, c = c0 + k * dc;
                        const float *col = score + (long)r * C + c;
                        float prev = 0.f, cur = col[0];
                        for (int z = 0; z < Z; ++z) {
                            const float nxt = z + 1 < Z ? col[(long)(z + 1) * P] : 0.f;
                            int a = z > 0 ? z - 1 : z;
                            float av = z > 0 ? prev : cur;
                            if (z > 0 && cur > av) { a = z; av = cur; }
                            if (z + 1 < Z && nxt > av) { a = z + 1; }
                            b3[(size_t)z * chunk + tid] = (unsigned char)a;
                            b2[(size_t)z * chunk + tid] = (unsigned char)((z + 1 < Z && nxt > cur) ? z + 1 : z);
                            prev = cur; cur = nxt;
                        }
                        // neighbours in upstream's order: up (row 0 wraps to the last row), down, left, right
                        int nb[4];
                        nb[0] = chosen[(long)(r > 0 ? r - 1 : R - 1) * C + c];
                        nb[1] = r < R - 1 ? chosen[(long)(r + 1) * C + c] : -1;
                        nb[2] = c > 0 ? chosen[(long)r * C + c - 1] : -1;
                        nb[3] = c < C - 1 ? chosen[(long)r * C + c + 1] : -1;
                        // the predecessor in the run (inside this chunk) is the variable
                        const int ps = tid == 0 ? -1 : (dr > 0 ? 0 : (dr < 0 ? 1 : (dc < 0 ? 3 : 2)));
                        for (int s = 0; s < Z; ++s) {
                            int n1 = -1, n2 = -1;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int vj = j == ps ? s : nb[j];
                                if (vj >= 0) { if (n1 < 0) n1 = vj; else if (n2 < 0) n2 = vj; }
                            }
                            int res;
                            if (n1 < 0) { res = 0; atomicOr(err, 2); }          // (upstream would raise: None - 1)
                            else if (n2 < 0 || n1 == n2) res = b3[(size_t)min(n1, Z - 1) * chunk + tid];
                            else if (n1 - n2 == 1 || n2 - n1 == 1) res = b2[(size_t)min(n1, n2) * chunk + tid];
                            else res = (n1 + n2) >> 1;
                            ta[(size_t)s * chunk + tid] = (unsigned char)res;
                            if (ps < 0 && s == 0) {          // constant function: one evaluation fills the table
                                for (int s2 = 1; s2 < Z; ++s2) ta[(size_t)s2 * chunk + tid] = (unsigned char)res;
                                break;
                            }
                        }
                    }
                    __syncthreads();
                    // ---- inclusive scan of function composition: after it, table i = F_i o ... o F_0 (constant) ---------
                    unsigned char *src = ta, *dst = tb;
                    for (int off = 1; off < cnt; off <<= 1) {
                        if (tid < cnt) {
                            if (tid >= off) {
                                for (int s = 0; s < Z; ++s)
                                    dst[(size_t)s * chunk + tid] = src[(size_t)src[(size_t)s * chunk + tid - off] * chunk + tid];
                            } else {
                                for (int s = 0; s < Z; ++s) dst[(size_t)s * chunk + tid] = src[(size_t)s * chunk + tid];
                            }
                        }
                        __syncthreads();
                        unsigned char *t = src; src = dst; dst = t;
                    }
                    if (tid < cnt) {
                        const int k = ks + tid, r = r0 + k * dr, c = c0 + k * dc;
                        chosen[(long)r * C + c] = src[tid];          // (row s = 0 of the composed, constant table)
                    }
                    __threadfence();
                    __syncthreads();
                }
            }
        }
    }
}

}  // namespace tip
