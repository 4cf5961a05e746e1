// tip_props.hip -- per-cell reductions over an int32 label map.
//
//   regionprops: skimage.measure.regionprops_table(labels, [label, area, perimeter, centroid, bbox]) (ti.py:891)
//                + intensity sums for intensity_mean (ti.py:2353)
//   neighbor pairs: the relation Tissue.find_neighbors evaluates with labels[dilated == i] (ti.py:1822-1835)
#include "tip_internal.h"

namespace tip {

__device__ __forceinline__ int lab_at(const int32_t *lab, int Y, int X, int y, int x)
{
    return (y < 0 || y >= Y || x < 0 || x >= X) ? 0 : lab[(long)y * X + x];
}

// border pixel of region l: carries l and has a 4-neighbour that does not (image edge counts as outside),
// i.e. image - binary_erosion(image, cross, border_value=0) of skimage.measure.perimeter
__device__ __forceinline__ int is_border(const int32_t *lab, int Y, int X, int y, int x, int l)
{
    if (y < 0 || y >= Y || x < 0 || x >= X) return 0;
    if (lab[(long)y * X + x] != l) return 0;
    if (y == 0 || lab[(long)(y - 1) * X + x] != l) return 1;
    if (y == Y - 1 || lab[(long)(y + 1) * X + x] != l) return 1;
    if (x == 0 || lab[(long)y * X + x - 1] != l) return 1;
    if (x == X - 1 || lab[(long)y * X + x + 1] != l) return 1;
    return 0;
}

struct PropAcc {
    unsigned long long area, sy, sx, p0, p1, p2;
    int ymin, ymax, xmin, xmax;
    double isum;
};

struct PropOut {
    unsigned long long *area, *sumy, *sumx, *pc;  // pc[3*l+k]
    int *bbox;                                    // bbox[4*l+{0..3}] = min_row, min_col, max_row+1, max_col+1
    double *isum;                                 // nullable
};

__device__ __forceinline__ void flush(const PropOut &o, int l, const PropAcc &a, bool has_i)
{
    const int k = l - 1;
    atomicAdd(&o.area[k], a.area);
    atomicAdd(&o.sumy[k], a.sy);
    atomicAdd(&o.sumx[k], a.sx);
    if (a.p0) atomicAdd(&o.pc[3 * k], a.p0);
    if (a.p1) atomicAdd(&o.pc[3 * k + 1], a.p1);
    if (a.p2) atomicAdd(&o.pc[3 * k + 2], a.p2);
    atomicMin(&o.bbox[4 * k], a.ymin);
    atomicMin(&o.bbox[4 * k + 1], a.xmin);
    atomicMax(&o.bbox[4 * k + 2], a.ymax + 1);
    atomicMax(&o.bbox[4 * k + 3], a.xmax + 1);
    if (has_i) atomicAdd(&o.isum[k], a.isum);
}

// each thread walks RUN consecutive pixels of a row and flushes one set of atomics per label run
constexpr int PROP_RUN = 16;
__global__ void __launch_bounds__(256) k_regionprops(const int32_t *__restrict__ lab, const double *__restrict__ inten, int Y,
                                                     int X, int nlab, PropOut o)
{
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * PROP_RUN, y = blockIdx.y;
    if (x0 >= X) return;
    const bool has_i = inten != nullptr;
    int cur = 0;
    PropAcc a;
    for (int x = x0; x < min(x0 + PROP_RUN, X); ++x) {
        const int l = lab[(long)y * X + x];
        if (l != cur) {
            if (cur > 0 && cur <= nlab) flush(o, cur, a, has_i);
            cur = l;
            a.area = a.sy = a.sx = a.p0 = a.p1 = a.p2 = 0;
            a.ymin = a.ymax = y;
            a.xmin = a.xmax = x;
            a.isum = 0.0;
        }
        if (l <= 0 || l > nlab) continue;
        a.area += 1;
        a.sy += (unsigned long long)y;
        a.sx += (unsigned long long)x;
        a.xmax = x;
        if (has_i) a.isum += inten[(long)y * X + x];
        if (is_border(lab, Y, X, y, x, l)) {
            int code = 1;
            code += 2 * (is_border(lab, Y, X, y - 1, x, l) + is_border(lab, Y, X, y + 1, x, l) +
                         is_border(lab, Y, X, y, x - 1, l) + is_border(lab, Y, X, y, x + 1, l));
            code += 10 * (is_border(lab, Y, X, y - 1, x - 1, l) + is_border(lab, Y, X, y - 1, x + 1, l) +
                          is_border(lab, Y, X, y + 1, x - 1, l) + is_border(lab, Y, X, y + 1, x + 1, l));
            if (code == 5 || code == 7 || code == 15 || code == 17 || code == 25 || code == 27) a.p0++;
            else if (code == 21 || code == 33) a.p1++;
            else if (code == 13 || code == 23) a.p2++;
        }
    }
    if (cur > 0 && cur <= nlab) flush(o, cur, a, has_i);
}

__global__ void __launch_bounds__(256) k_props_init(PropOut o, int nlab, int Y, int X)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nlab) return;
    o.area[k] = o.sumy[k] = o.sumx[k] = 0;
    o.pc[3 * k] = o.pc[3 * k + 1] = o.pc[3 * k + 2] = 0;
    o.bbox[4 * k] = Y; o.bbox[4 * k + 1] = X; o.bbox[4 * k + 2] = 0; o.bbox[4 * k + 3] = 0;
    if (o.isum) o.isum[k] = 0.0;
}

__global__ void __launch_bounds__(256) k_props_finish(PropOut o, int nlab, int64_t *__restrict__ bbox64)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 4 * nlab) return;
    bbox64[k] = o.bbox[k];
}

int regionprops_dev(const int32_t *labels, const double *intensity, int Y, int X, int n, int64_t *area, int64_t *bbox4,
                    int64_t *sumy, int64_t *sumx, int64_t *pc3, double *isum)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || !area || !bbox4 || !sumy || !sumx || !pc3) return fail(TIP_ERR_ARG, "regionprops: null pointer");
    if ((intensity == nullptr) != (isum == nullptr)) return fail(TIP_ERR_ARG, "regionprops: intensity and isum go together");
    if (Y < 1 || X < 1 || Y > 65535 || n < 0) return fail(TIP_ERR_ARG, "regionprops: bad shape");
    if (n == 0) return TIP_OK;
    WsGuard ws;
    int *bbox32 = ws.get<int>((size_t)4 * n);
    if (!bbox32) return TIP_ERR_NOMEM;
    PropOut o{(unsigned long long *)area, (unsigned long long *)sumy, (unsigned long long *)sumx,
              (unsigned long long *)pc3, bbox32, isum};
    TIP_LAUNCH("props_init", k_props_init, dim3(cdiv(n, 256)), dim3(256), 0, o, n, Y, X);
    TIP_LAUNCH("regionprops", k_regionprops, dim3(cdiv(cdiv(X, PROP_RUN), 256), Y), dim3(256), 0, labels, intensity, Y, X, n, o);
    TIP_LAUNCH("props_finish", k_props_finish, dim3(cdiv(4L * n, 256)), dim3(256), 0, o, n, bbox4);
    return TIP_OK;
}

// ---- neighbour pairs: hash set of (hi<<32 | lo) keys -------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

__global__ void __launch_bounds__(256) k_neighbor_pairs(const int32_t *__restrict__ lab, int Y, int X,
                                                        unsigned long long *__restrict__ table, unsigned long long tmask,
                                                        int32_t *__restrict__ pairs, long long cap,
                                                        unsigned long long *__restrict__ count)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    const int l = lab[(long)y * X + x];
    if (l <= 0) return;
    int m = 0;  // zero padding takes part (mode='constant'); labels are >= 0 on this path
#pragma unroll
    for (int j = -2; j <= 2; ++j)
#pragma unroll
        for (int i = -2; i <= 2; ++i) {
            const int q = lab_at(lab, Y, X, y + j, x + i);
            m = q > m ? q : m;
        }
    if (m == l) return;
    const unsigned long long key = ((unsigned long long)(unsigned)m << 32) | (unsigned)l;
    unsigned long long h = mix64(key) & tmask;
    for (;;) {
        const unsigned long long cur = table[h];
        if (cur == key) return;
        if (cur == 0ULL) {
            const unsigned long long old = atomicCAS(&table[h], 0ULL, key);
            if (old == 0ULL) {
                const unsigned long long slot = atomicAdd(count, 1ULL);
                if ((long long)slot < cap) { pairs[2 * slot] = m; pairs[2 * slot + 1] = l; }
                return;
            }
            if (old == key) return;
        }
        h = (h + 1) & tmask;
    }
}

int neighbor_pairs_dev(const int32_t *labels, int Y, int X, int32_t *pairs_dev, int64_t cap, int64_t *n_pairs_host)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || !pairs_dev || !n_pairs_host || cap < 1) return fail(TIP_ERR_ARG, "neighbor_pairs: bad arguments");
    if (Y < 1 || X < 1 || Y > 65535) return fail(TIP_ERR_ARG, "neighbor_pairs: bad shape");
    unsigned long long tsize = 1024;
    while (tsize < (unsigned long long)cap * 2) tsize <<= 1;
    WsGuard ws;
    unsigned long long *table = ws.get<unsigned long long>(tsize), *count = ws.get<unsigned long long>(1);
    if (!table || !count) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemsetAsync(table, 0, tsize * 8, c.stream));
    TIP_HIP(hipMemsetAsync(count, 0, 8, c.stream));
    TIP_LAUNCH("neighbor_pairs", k_neighbor_pairs, dim3(cdiv(X, 256), Y), dim3(256), 0, labels, Y, X, table, tsize - 1,
               pairs_dev, (long long)cap, count);
    unsigned long long h = 0;
    TIP_HIP(hipMemcpyAsync(&h, count, 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    *n_pairs_host = (int64_t)h;
    if ((int64_t)h > cap) return fail(TIP_ERR_OVERFLOW, "neighbor_pairs: %lld pairs exceed capacity %lld", (long long)h, (long long)cap);
    return TIP_OK;
}

}  // namespace tip

using namespace tip;

extern "C" {

int tip_regionprops_i32_dev(const int32_t *labels, const double *intensity, int y, int x, int n, int64_t *area,
                            int64_t *bbox4, int64_t *sumy, int64_t *sumx, int64_t *pc3, double *isum)
{
    return regionprops_dev(labels, intensity, y, x, n, area, bbox4, sumy, sumx, pc3, isum);
}

int tip_regionprops_i32(const int32_t *labels, const double *intensity, int y, int x, int n, int64_t *area, int64_t *bbox4,
                        int64_t *sumy, int64_t *sumx, int64_t *pc3, double *isum)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || y < 1 || x < 1 || n < 0) return fail(TIP_ERR_ARG, "tip_regionprops_i32: bad arguments");
    if (n == 0) return TIP_OK;
    const size_t P = (size_t)y * x;
    WsGuard ws;
    int32_t *dl = ws.get<int32_t>(P);
    double *di = intensity ? ws.get<double>(P) : nullptr;
    int64_t *da = ws.get<int64_t>(n), *db = ws.get<int64_t>((size_t)4 * n), *dsy = ws.get<int64_t>(n),
            *dsx = ws.get<int64_t>(n), *dp = ws.get<int64_t>((size_t)3 * n);
    double *dis = intensity ? ws.get<double>(n) : nullptr;
    if (!dl || !da || !db || !dsy || !dsx || !dp || (intensity && (!di || !dis))) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(dl, labels, P * 4, hipMemcpyHostToDevice, c.stream));
    if (intensity) TIP_HIP(hipMemcpyAsync(di, intensity, P * 8, hipMemcpyHostToDevice, c.stream));
    int rc = regionprops_dev(dl, di, y, x, n, da, db, dsy, dsx, dp, dis);
    if (rc) return rc;
    TIP_HIP(hipMemcpyAsync(area, da, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(bbox4, db, (size_t)n * 32, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(sumy, dsy, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(sumx, dsx, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(pc3, dp, (size_t)n * 24, hipMemcpyDeviceToHost, c.stream));
    if (isum) TIP_HIP(hipMemcpyAsync(isum, dis, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_neighbor_pairs_i32_dev(const int32_t *labels, int y, int x, int32_t *pairs_dev, int64_t cap, int64_t *n_pairs_host)
{
    return neighbor_pairs_dev(labels, y, x, pairs_dev, cap, n_pairs_host);
}

int tip_neighbor_pairs_i32(const int32_t *labels, int y, int x, int32_t *pairs, int64_t cap, int64_t *n_pairs)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || !pairs || !n_pairs || y < 1 || x < 1 || cap < 1) return fail(TIP_ERR_ARG, "tip_neighbor_pairs_i32: bad arguments");
    const size_t P = (size_t)y * x;
    WsGuard ws;
    int32_t *dl = ws.get<int32_t>(P), *dp = ws.get<int32_t>((size_t)2 * cap);
    if (!dl || !dp) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(dl, labels, P * 4, hipMemcpyHostToDevice, c.stream));
    int rc = neighbor_pairs_dev(dl, y, x, dp, cap, n_pairs);
    if (rc) return rc;
    TIP_HIP(hipMemcpyAsync(pairs, dp, (size_t)(*n_pairs) * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

}  // extern "C"
