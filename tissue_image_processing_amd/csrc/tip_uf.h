// tip_uf.h -- lock-free union-find over a 2-D grid (4-connectivity), roots = smallest linear index of a
// component = its raster-first pixel, which is what makes raster-order label numbering a prefix sum.
#pragma once
#include "tip_internal.h"
#include <algorithm>

namespace tip {

__device__ __forceinline__ int uf_find(const int *parent, int x)
{
    int p = parent[x];
    while (p != x) { x = p; p = parent[x]; }
    return x;
}

// parent pointers only ever decrease, so concurrent unions stay acyclic
__device__ __forceinline__ void uf_unite(int *parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }  // a > b: hang a under b
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

static __global__ void __launch_bounds__(256) k_uf_init(int *__restrict__ parent, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = (int)i;
}

// every pixel unites with its "same" left and upper neighbours
template <typename Same>
__global__ void __launch_bounds__(256) k_uf_merge(Same s, int *__restrict__ parent, int Y, int X)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    const int i = y * X + x;
    if (!s.valid(i)) return;
    if (x > 0 && s.valid(i - 1) && s.same(i, i - 1)) uf_unite(parent, i, i - 1);
    if (y > 0 && s.valid(i - X) && s.same(i, i - X)) uf_unite(parent, i, i - X);
}

// Tiled variant (frames of at least a few tiles).  Large components -- the thousands of cell-interior blobs of a U-Net boundary
// image, a few hundred pixels each -- cost the one-level kernel long chains of global atomics (0.44 ms per 2048^2 frame); here a block
// first solves its UF_T x UF_T tile in LDS (the same lock-free union, local indices: raster order inside a tile is the order of the
// global indices, so "hang the larger under the smaller" keeps the raster-first pixel as root at both levels), writes global parents,
// and a second kernel unites across the tile borders only.
constexpr int UF_T = 32;
__device__ __forceinline__ int uf_find_lds(const int *lp, int x)
{
    int p = lp[x];
    while (p != x) { x = p; p = lp[x]; }
    return x;
}
__device__ __forceinline__ void uf_unite_lds(int *lp, int a, int b)
{
    for (;;) {
        a = uf_find_lds(lp, a);
        b = uf_find_lds(lp, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&lp[a], b);
        if (old == a) return;
        a = old;
    }
}
template <typename Same>
__global__ void __launch_bounds__(256) k_uf_tiles(Same s, int *__restrict__ parent, int Y, int X)
{
    __shared__ int lp[UF_T * UF_T];
    const int x0 = blockIdx.x * UF_T, y0 = blockIdx.y * UF_T;
    for (int l = threadIdx.x; l < UF_T * UF_T; l += 256) lp[l] = l;
    __syncthreads();
    for (int l = threadIdx.x; l < UF_T * UF_T; l += 256) {
        const int lx = l % UF_T, ly = l / UF_T, x = x0 + lx, y = y0 + ly;
        if (x >= X || y >= Y) continue;
        const int i = y * X + x;
        if (!s.valid(i)) continue;
        if (lx > 0 && s.valid(i - 1) && s.same(i, i - 1)) uf_unite_lds(lp, l, l - 1);
        if (ly > 0 && s.valid(i - X) && s.same(i, i - X)) uf_unite_lds(lp, l, l - UF_T);
    }
    __syncthreads();
    for (int l = threadIdx.x; l < UF_T * UF_T; l += 256) {
        const int lx = l % UF_T, ly = l / UF_T, x = x0 + lx, y = y0 + ly;
        if (x >= X || y >= Y) continue;
        const int r = uf_find_lds(lp, l);
        parent[y * X + x] = (y0 + r / UF_T) * X + x0 + r % UF_T;
    }
}
// ... and across the tile borders: the pixels of a tile's first column / first row with their left / upper neighbours
template <typename Same>
__global__ void __launch_bounds__(256) k_uf_borders(Same s, int *__restrict__ parent, int Y, int X)
{
    // blockIdx.y = 0: vertical borders (x a multiple of UF_T, every y), 1: horizontal borders (y a multiple of UF_T, every x)
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) {
        const int nbx = (X - 1) / UF_T;                       // borders at x = UF_T, 2 UF_T, ...
        if (t >= (long)nbx * Y) return;
        const int y = (int)(t / nbx), x = ((int)(t % nbx) + 1) * UF_T;
        const int i = y * X + x;
        if (s.valid(i) && s.valid(i - 1) && s.same(i, i - 1)) uf_unite(parent, i, i - 1);
    } else {
        const int nby = (Y - 1) / UF_T;
        if (t >= (long)nby * X) return;
        const int x = (int)(t % X), y = ((int)(t / X) + 1) * UF_T;
        const int i = y * X + x;
        if (s.valid(i) && s.valid(i - X) && s.same(i, i - X)) uf_unite(parent, i, i - X);
    }
}

static __global__ void __launch_bounds__(256) k_uf_flatten(int *__restrict__ parent, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = uf_find(parent, (int)i);
}

template <typename Same>
int uf_components(Same s, int *parent, int Y, int X)
{
    const long n = (long)Y * X;
    if (Y >= 2 * UF_T && X >= 2 * UF_T && !tuning().uf_one_level) {
        TIP_LAUNCH("uf_tiles", (k_uf_tiles<Same>), dim3(cdiv(X, UF_T), cdiv(Y, UF_T)), dim3(256), 0, s, parent, Y, X);
        const long nb = std::max<long>((long)((X - 1) / UF_T) * Y, (long)((Y - 1) / UF_T) * X);
        TIP_LAUNCH("uf_borders", (k_uf_borders<Same>), dim3(cdiv(nb, 256), 2), dim3(256), 0, s, parent, Y, X);
    } else {
        TIP_LAUNCH("uf_init", k_uf_init, dim3(cdiv(n, 256)), dim3(256), 0, parent, n);
        TIP_LAUNCH("uf_merge", (k_uf_merge<Same>), dim3(cdiv(X, 256), Y), dim3(256), 0, s, parent, Y, X);
    }
    TIP_LAUNCH("uf_flatten", k_uf_flatten, dim3(cdiv(n, 256)), dim3(256), 0, parent, n);
    return TIP_OK;
}

int exclusive_scan_i32(const int *in, int *out, long n, int *total_dev);

}  // namespace tip
