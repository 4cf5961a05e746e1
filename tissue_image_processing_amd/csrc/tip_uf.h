// tip_uf.h -- lock-free union-find over a 2-D grid (4-connectivity), roots = smallest linear index of a
// component = its raster-first pixel, which is what makes raster-order label numbering a prefix sum.
#pragma once
#include "tip_internal.h"

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

static __global__ void __launch_bounds__(256) k_uf_flatten(int *__restrict__ parent, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = uf_find(parent, (int)i);
}

template <typename Same>
int uf_components(Same s, int *parent, int Y, int X)
{
    const long n = (long)Y * X;
    TIP_LAUNCH("uf_init", k_uf_init, dim3(cdiv(n, 256)), dim3(256), 0, parent, n);
    TIP_LAUNCH("uf_merge", (k_uf_merge<Same>), dim3(cdiv(X, 256), Y), dim3(256), 0, s, parent, Y, X);
    TIP_LAUNCH("uf_flatten", k_uf_flatten, dim3(cdiv(n, 256)), dim3(256), 0, parent, n);
    return TIP_OK;
}

int exclusive_scan_i32(const int *in, int *out, long n, int *total_dev);

}  // namespace tip
