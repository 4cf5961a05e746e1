// tip_label.hip -- device-wide scan, union-find connected components (raster-order numbering), rank filters.
//
//   label4:      skimage.measure.label(connectivity=1, background=bg)   (ti.py:2922, 3470)
//                == scipy.ndimage.label on a boolean image               (watershed markers, _watershed.py:76)
//   rank filter: scipy.ndimage.maximum_filter / minimum_filter and grey erosion / dilation with a flat
//                footprint (ti.py:1822,2081,2969,4079-4084; pl.py:170-193; bim.py:468-472 via threshold_local)
#include "tip_internal.h"
#include "tip_uf.h"

namespace tip {

// ---------------------------------------------------------------------------------------------------------
// exclusive scan of int32 (three kernels: per-block scan, scan of block sums, add)
// ---------------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8, SCAN_THREADS = 256, SCAN_BLOCK = SCAN_ITEMS * SCAN_THREADS;

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_block(const int *__restrict__ in, int *__restrict__ out, long n,
                                                             int *__restrict__ bsum)
{
    __shared__ int wsum[SCAN_THREADS / 64];
    const long base = (long)blockIdx.x * SCAN_BLOCK + (long)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    // wave inclusive scan of s
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int run = woff + inc - s;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == SCAN_THREADS - 1) bsum[blockIdx.x] = run;
}

__global__ void __launch_bounds__(1024) k_scan_sums(int *__restrict__ bsum, int nb, int *__restrict__ total)
{
    __shared__ int buf[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? bsum[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int t = threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        const int incl = buf[threadIdx.x];
        const int c = carry;
        if (i < nb) bsum[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

// ... with the scan of the block sums folded in (few blocks: every block adds up the sums in front of it itself -- one launch less;
// the last block writes the total)
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_add_own(int *__restrict__ out, long n, const int *__restrict__ bsum, int nb,
                                                               int *__restrict__ total)
{
    __shared__ int wsum[SCAN_THREADS / 64];
    int s = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += SCAN_THREADS) s += bsum[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) off += wsum[w];
    if (total && (int)blockIdx.x == nb - 1 && threadIdx.x == 0) *total = off + bsum[nb - 1];
    const long base = (long)blockIdx.x * SCAN_BLOCK + (long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) out[base + i] += off;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_add(int *__restrict__ out, long n, const int *__restrict__ bsum)
{
    const int off = bsum[blockIdx.x];
    const long base = (long)blockIdx.x * SCAN_BLOCK + (long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) out[base + i] += off;
}

int exclusive_scan_i32(const int *in, int *out, long n, int *total_dev)
{
    const int nb = cdiv(n, SCAN_BLOCK);
    WsGuard ws;
    int *bsum = ws.get<int>(nb);
    if (!bsum) return TIP_ERR_NOMEM;
    TIP_LAUNCH("scan_block", k_scan_block, dim3(nb), dim3(SCAN_THREADS), 0, in, out, n, bsum);
    if (nb <= 4096) {          // (up to 8.4 M elements: a block reads at most 16 KB of block sums out of L2)
        TIP_LAUNCH("scan_add_own", k_scan_add_own, dim3(nb), dim3(SCAN_THREADS), 0, out, n, (const int *)bsum, nb, total_dev);
        return TIP_OK;
    }
    TIP_LAUNCH("scan_sums", k_scan_sums, dim3(1), dim3(1024), 0, bsum, nb, total_dev);
    TIP_LAUNCH("scan_add", k_scan_add, dim3(nb), dim3(SCAN_THREADS), 0, out, n, (const int *)bsum);
    return TIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// connected components with raster-order numbering
// ---------------------------------------------------------------------------------------------------------
struct SameI32 {
    const int32_t *in;
    int32_t bg;
    __device__ __forceinline__ bool valid(int i) const { return in[i] != bg; }
    __device__ __forceinline__ bool same(int i, int j) const { return in[i] == in[j]; }
};

__global__ void __launch_bounds__(256) k_root_flags_i32(SameI32 s, const int *__restrict__ parent, int *__restrict__ flag,
                                                        long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (s.valid((int)i) && parent[i] == (int)i) ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_relabel_i32(SameI32 s, const int *__restrict__ parent,
                                                     const int *__restrict__ rank, int32_t *__restrict__ out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = s.valid((int)i) ? rank[parent[i]] + 1 : 0;
}

int label4_dev(const int32_t *in, int32_t bg, int32_t *out, int Y, int X, int32_t *n_labels_host)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!in || !out) return fail(TIP_ERR_ARG, "label4: null pointer");
    if (Y < 1 || X < 1 || (long)Y * X > 2147483647L) return fail(TIP_ERR_ARG, "label4: bad shape %dx%d", Y, X);
    const long n = (long)Y * X;
    WsGuard ws;
    int *parent = ws.get<int>(n), *flag = ws.get<int>(n), *rank = ws.get<int>(n), *total = ws.get<int>(1);
    if (!parent || !flag || !rank || !total) return TIP_ERR_NOMEM;
    SameI32 s{in, bg};
    int rc = uf_components(s, parent, Y, X);
    if (rc) return rc;
    TIP_LAUNCH("ccl_root_flags", k_root_flags_i32, dim3(cdiv(n, 256)), dim3(256), 0, s, (const int *)parent, flag, n);
    if ((rc = exclusive_scan_i32(flag, rank, n, total))) return rc;
    TIP_LAUNCH("ccl_relabel", k_relabel_i32, dim3(cdiv(n, 256)), dim3(256), 0, s, (const int *)parent,
               (const int *)rank, out, n);
    if (n_labels_host) {
        int h = 0;
        TIP_HIP(hipMemcpyAsync(&h, total, sizeof(int), hipMemcpyDeviceToHost, c.stream));
        TIP_HIP(hipStreamSynchronize(c.stream));
        *n_labels_host = h;
    }
    return TIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// rank filters
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n)
{
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// footprint_kind 0: ky x kx rectangle (window i-k/2 .. i-k/2+k-1, scipy origin 0); 1: 3x3 cross without centre
template <typename T>
__global__ void __launch_bounds__(256) k_rank2d(const T *__restrict__ in, T *__restrict__ out, int Y, int X, int ky, int kx,
                                                int fp, int reflect, int is_max)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    const int oy = ky / 2, ox = kx / 2;
    bool have = false;
    T best = 0;
    for (int j = 0; j < ky; ++j)
        for (int i = 0; i < kx; ++i) {
            if (fp == 1 && !(((j == 1) ^ (i == 1)))) continue;  // cross: exactly one of (row, col) is the middle
            int yy = y - oy + j, xx = x - ox + i;
            T v;
            if (reflect) {
                v = in[(long)reflect_idx(yy, Y) * X + reflect_idx(xx, X)];
            } else {
                v = (yy < 0 || yy >= Y || xx < 0 || xx >= X) ? (T)0 : in[(long)yy * X + xx];
            }
            if (!have) { best = v; have = true; }
            else if (is_max ? (v > best) : (v < best)) best = v;
        }
    out[(long)y * X + x] = best;
}

int rankfilter2d_dev(const void *in, void *out, int dtype, int Y, int X, int ky, int kx, int fp, int border, int is_max)
{
    if (!in || !out || in == out) return fail(TIP_ERR_ARG, "rankfilter2d: null or aliased pointers");
    if (Y < 1 || X < 1 || Y > 65535) return fail(TIP_ERR_ARG, "rankfilter2d: bad shape");
    if (ky < 1 || kx < 1 || ky > 31 || kx > 31) return fail(TIP_ERR_ARG, "rankfilter2d: window %dx%d", ky, kx);
    if (fp == 1 && (ky != 3 || kx != 3)) return fail(TIP_ERR_ARG, "cross footprint is 3x3");
    if (fp != 0 && fp != 1) return fail(TIP_ERR_ARG, "footprint_kind %d", fp);
    dim3 grid(cdiv(X, 256), Y), block(256);
    if (dtype == 1)
        TIP_LAUNCH("rank2d_f64", k_rank2d<double>, grid, block, 0, (const double *)in, (double *)out, Y, X, ky, kx, fp, border,
                   is_max);
    else if (dtype == 2)
        TIP_LAUNCH("rank2d_i32", k_rank2d<int32_t>, grid, block, 0, (const int32_t *)in, (int32_t *)out, Y, X, ky, kx, fp,
                   border, is_max);
    else
        return fail(TIP_ERR_ARG, "rankfilter2d: dtype %d (1=f64, 2=i32)", dtype);
    return TIP_OK;
}

// bim.py:464-473: thr = imgthresh * max(block x block, reflect) (float64); out = img < thr ? 0 : img
__global__ void __launch_bounds__(256) k_local_threshold(const double *__restrict__ img, double *__restrict__ out, int Y, int X,
                                                         double imgthresh, int block)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    const int o = block / 2;
    double best = img[(long)reflect_idx(y - o, Y) * X + reflect_idx(x - o, X)];
    for (int j = 0; j < block; ++j)
        for (int i = 0; i < block; ++i) {
            const double v = img[(long)reflect_idx(y - o + j, Y) * X + reflect_idx(x - o + i, X)];
            if (v > best) best = v;
        }
    const double thr = imgthresh * best;
    const double v = img[(long)y * X + x];
    out[(long)y * X + x] = v < thr ? 0.0 : v;
}

__global__ void __launch_bounds__(256) k_update_labels(const int32_t *__restrict__ in, int32_t *__restrict__ out, int Y, int X)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    int32_t v = in[(long)y * X + x];
    if (v < 0) {
        int32_t best = 0;  // zero padding takes part in the maximum (mode='constant')
        bool have = false;
        for (int j = -1; j <= 1; ++j)
            for (int i = -1; i <= 1; ++i) {
                const int yy = y + j, xx = x + i;
                const int32_t q = (yy < 0 || yy >= Y || xx < 0 || xx >= X) ? 0 : in[(long)yy * X + xx];
                if (!have) { best = q; have = true; }
                else if (q > best) best = q;
            }
        v = best;
    }
    out[(long)y * X + x] = v;
}

__global__ void __launch_bounds__(256) k_lut_gather(const int32_t *__restrict__ labels, const int64_t *__restrict__ lut,
                                                    int64_t n_lut, int64_t *__restrict__ out, long n, int *__restrict__ err)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = labels[i];
    if (l < 0 || l >= n_lut) { atomicOr(err, 1); out[i] = 0; return; }
    out[i] = lut[l];
}

// T3 lookup (ti.py:2081-2090): value of maximum_filter(labels, (3,3), mode='constant') at rounded query points
__global__ void __launch_bounds__(256) k_lookup_max3(const int32_t *__restrict__ lab, int Y, int X, const int64_t *__restrict__ qy,
                                                     const int64_t *__restrict__ qx, long n, int32_t *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long y = qy[i], x = qx[i];
    if (y < 0 || y >= Y || x < 0 || x >= X) { out[i] = -1; return; }  // invalid location (ti.py:2086-2087)
    int32_t best = 0;
    bool have = false;
    for (int j = -1; j <= 1; ++j)
        for (int k = -1; k <= 1; ++k) {
            const long yy = y + j, xx = x + k;
            const int32_t q = (yy < 0 || yy >= Y || xx < 0 || xx >= X) ? 0 : lab[yy * X + xx];
            if (!have) { best = q; have = true; }
            else if (q > best) best = q;
        }
    out[i] = best;
}

}  // namespace tip

using namespace tip;

extern "C" {

int tip_label4_i32_dev(const int32_t *in, int32_t bg, int32_t *out, int y, int x, int32_t *n_labels_host)
{
    return label4_dev(in, bg, out, y, x, n_labels_host);
}

int tip_label4_i32(const int32_t *in, int32_t bg, int32_t *out, int y, int x, int32_t *n_labels)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!in || !out) return fail(TIP_ERR_ARG, "tip_label4_i32: null pointer");
    if (y < 1 || x < 1) return fail(TIP_ERR_ARG, "tip_label4_i32: empty image");
    const size_t bytes = (size_t)y * x * 4;
    WsGuard ws;
    int32_t *din = ws.get<int32_t>((size_t)y * x), *dout = ws.get<int32_t>((size_t)y * x);
    if (!din || !dout) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, c.stream));
    int32_t nl = 0;
    int rc = label4_dev(din, bg, dout, y, x, &nl);
    if (rc) return rc;
    TIP_HIP(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    if (n_labels) *n_labels = nl;
    return TIP_OK;
}

int tip_rankfilter2d_dev(const void *in, void *out, int dtype, int y, int x, int ky, int kx, int footprint_kind,
                         int border_mode, int is_max)
{
    return rankfilter2d_dev(in, out, dtype, y, x, ky, kx, footprint_kind, border_mode, is_max);
}

int tip_rankfilter2d(const void *in, void *out, int dtype, int y, int x, int ky, int kx, int footprint_kind,
                     int border_mode, int is_max)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!in || !out) return fail(TIP_ERR_ARG, "tip_rankfilter2d: null pointer");
    if (dtype != 1 && dtype != 2) return fail(TIP_ERR_ARG, "tip_rankfilter2d: dtype %d (1=f64, 2=i32)", dtype);
    if (y < 1 || x < 1) return fail(TIP_ERR_ARG, "tip_rankfilter2d: empty image");
    const size_t bytes = (size_t)y * x * (dtype == 1 ? 8 : 4);
    WsGuard ws;
    char *din = ws.get<char>(bytes), *dout = ws.get<char>(bytes);
    if (!din || !dout) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, c.stream));
    int rc = rankfilter2d_dev(din, dout, dtype, y, x, ky, kx, footprint_kind, border_mode, is_max);
    if (rc) return rc;
    TIP_HIP(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_local_threshold_f64_dev(const double *img, double *out, int y, int x, double imgthresh, int block)
{
    if (!img || !out || img == out) return fail(TIP_ERR_ARG, "tip_local_threshold_f64_dev: null or aliased pointers");
    if (block < 1 || block > 63) return fail(TIP_ERR_ARG, "block size %d", block);
    if (block % 2 == 0) block += 1;  // bim.py:466-467
    if (y < 1 || x < 1 || y > 65535) return fail(TIP_ERR_ARG, "bad shape");
    TIP_LAUNCH("local_threshold", k_local_threshold, dim3(cdiv(x, 256), y), dim3(256), 0, img, out, y, x, imgthresh, block);
    return TIP_OK;
}

int tip_update_labels_i32(int32_t *labels, int y, int x)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || y < 1 || x < 1 || y > 65535) return fail(TIP_ERR_ARG, "tip_update_labels_i32: bad arguments");
    const size_t bytes = (size_t)y * x * 4;
    WsGuard ws;
    int32_t *din = ws.get<int32_t>((size_t)y * x), *dout = ws.get<int32_t>((size_t)y * x);
    if (!din || !dout) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(din, labels, bytes, hipMemcpyHostToDevice, c.stream));
    TIP_LAUNCH("update_labels", k_update_labels, dim3(cdiv(x, 256), y), dim3(256), 0, (const int32_t *)din, dout, y, x);
    TIP_HIP(hipMemcpyAsync(labels, dout, bytes, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_lookup_max3_i32_dev(const int32_t *labels, int y, int x, const int64_t *qy_host, const int64_t *qx_host, int64_t n,
                            int32_t *out_host)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || y < 1 || x < 1 || n < 0) return fail(TIP_ERR_ARG, "tip_lookup_max3_i32_dev: bad arguments");
    if (n == 0) return TIP_OK;
    if (!qy_host || !qx_host || !out_host) return fail(TIP_ERR_ARG, "tip_lookup_max3_i32_dev: null pointer");
    WsGuard ws;
    int64_t *dy = ws.get<int64_t>(n), *dx = ws.get<int64_t>(n);
    int32_t *dout = ws.get<int32_t>(n);
    if (!dy || !dx || !dout) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(dy, qy_host, n * 8, hipMemcpyHostToDevice, c.stream));
    TIP_HIP(hipMemcpyAsync(dx, qx_host, n * 8, hipMemcpyHostToDevice, c.stream));
    TIP_LAUNCH("lookup_max3", k_lookup_max3, dim3(cdiv(n, 256)), dim3(256), 0, labels, y, x, (const int64_t *)dy,
               (const int64_t *)dx, (long)n, dout);
    TIP_HIP(hipMemcpyAsync(out_host, dout, n * 4, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_lut_gather_i32(const int32_t *labels, const int64_t *lut, int64_t n_lut, int64_t *out, int64_t n)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || !lut || !out || n < 1 || n_lut < 1) return fail(TIP_ERR_ARG, "tip_lut_gather_i32: bad arguments");
    WsGuard ws;
    int32_t *dl = ws.get<int32_t>(n);
    int64_t *dlut = ws.get<int64_t>(n_lut), *dout = ws.get<int64_t>(n);
    int *err = ws.get<int>(1);
    if (!dl || !dlut || !dout || !err) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(dl, labels, n * 4, hipMemcpyHostToDevice, c.stream));
    TIP_HIP(hipMemcpyAsync(dlut, lut, n_lut * 8, hipMemcpyHostToDevice, c.stream));
    TIP_HIP(hipMemsetAsync(err, 0, 4, c.stream));
    TIP_LAUNCH("lut_gather", k_lut_gather, dim3(cdiv(n, 256)), dim3(256), 0, (const int32_t *)dl, (const int64_t *)dlut, n_lut,
               dout, (long)n, err);
    int h = 0;
    TIP_HIP(hipMemcpyAsync(out, dout, n * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(&h, err, 4, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    if (h) return fail(TIP_ERR_INDEX, "label outside the lookup table (index out of bounds)");
    return TIP_OK;
}

}  // extern "C"
