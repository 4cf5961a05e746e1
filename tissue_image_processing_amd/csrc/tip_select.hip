// tip_select.hip -- exact order statistics per label (and of a whole frame) by most-significant-digit radix select.
//
//   calc_cell_types (ti.py:2338-2391) asks, for every cell, np.percentile(intensity[cell], 100 - p) and, for the frame,
//   np.percentile(intensity, 99): order statistics k and k+1 of the pixel values plus a linear interpolation.  The host
//   used to sort all pixels by (label, value).  Here the values are mapped to sortable 64-bit keys and the k-th key of
//   every label is found digit by digit: 8 passes over the pixels, each histogramming one byte of the keys that still
//   match the label's prefix (256 bins per label), each followed by a per-label pick of the bin that holds rank k.  The
//   (k+1)-th statistic is the same key when that key occurs often enough, else the smallest larger key (one more pass).
//   The interpolation stays on the host (numpy's arithmetic, a few thousand values).
#include "tip_internal.h"

namespace tip {

__device__ __forceinline__ unsigned long long sel_enc(double d)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(d + 0.0);   // (-0.0 sorts with +0.0, like numpy's <)
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double sel_dec(unsigned long long e)
{
    unsigned long long b = (e >> 63) ? (e & 0x7fffffffffffffffULL) : ~e;
    return __longlong_as_double((long long)b);
}

// add 1 to hist[slot] for every active lane, lanes with equal slots combined into one atomic (neighbouring pixels mostly
// share label and leading digits: a wave touches a handful of distinct slots)
__device__ __forceinline__ void wave_hist_add(bool active, long slot, unsigned int *__restrict__ hist)
{
    unsigned long long todo = __ballot(active);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const long ls = __shfl(slot, leader, 64);
        const unsigned long long same = __ballot(active && slot == ls) & todo;
        if (lane == leader) atomicAdd(&hist[ls], (unsigned int)__popcll(same));
        todo &= ~same;
    }
}

// one digit: labels == nullptr -> a single pseudo-label 0 for every pixel (whole-frame statistic)
__global__ void __launch_bounds__(256) k_sel_hist(const int32_t *__restrict__ labels, const double *__restrict__ img, long n, int nlab,
                                                  const unsigned long long *__restrict__ prefix, int shift,
                                                  unsigned int *__restrict__ hist)
{
    __shared__ unsigned int sh[256];
    const bool whole = labels == nullptr;
    if (whole) { sh[threadIdx.x] = 0; __syncthreads(); }
    const long i0 = (long)blockIdx.x * blockDim.x * 8 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long i = i0 + (long)u * blockDim.x;
        bool active = i < n;
        int l = 0;
        if (active && !whole) { l = labels[i] - 1; active = l >= 0 && l < nlab; }
        unsigned long long key = 0;
        if (active) {
            key = sel_enc(img[i]);
            if (shift < 56 && (key >> (shift + 8)) != (prefix[l] >> (shift + 8))) active = false;
        }
        const int bin = (int)((key >> shift) & 255ULL);
        if (whole) { if (active) atomicAdd(&sh[bin], 1u); }
        else wave_hist_add(active, (long)l * 256 + bin, hist);
    }
    if (whole) {
        __syncthreads();
        if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
    }
}

// per label: the bin holding rank[l]; prefix gets the digit, rank becomes the rank inside the bin; on the last digit
// `room` = how many more copies of the selected key follow the chosen one (so that rank+1 can be answered)
__global__ void __launch_bounds__(256) k_sel_pick(unsigned int *__restrict__ hist, long long *__restrict__ rank,
                                                  unsigned long long *__restrict__ prefix, int shift, int nlab,
                                                  long long *__restrict__ room)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlab) return;
    unsigned int *h = hist + (long)l * 256;
    long long r = rank[l];
    if (r < 0) return;            // label without pixels
    long long cum = 0;
    int pick = 255;
    for (int b = 0; b < 256; ++b) {
        const long long c = h[b];
        if (cum + c > r) { pick = b; break; }
        cum += c;
    }
    if (shift == 0) room[l] = (long long)h[pick] - (r - cum) - 1;
    prefix[l] |= (unsigned long long)pick << shift;
    rank[l] = r - cum;
    for (int b = 0; b < 256; ++b) h[b] = 0;
}

// smallest key strictly above the selected one, per label
__global__ void __launch_bounds__(256) k_sel_next(const int32_t *__restrict__ labels, const double *__restrict__ img, long n, int nlab,
                                                  const unsigned long long *__restrict__ prefix, unsigned long long *__restrict__ above)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int l = 0;
    if (labels) { l = labels[i] - 1; if (l < 0 || l >= nlab) return; }
    const unsigned long long key = sel_enc(img[i]);
    if (key > prefix[l] && key < above[l]) atomicMin(&above[l], key);
}

__global__ void __launch_bounds__(256) k_sel_emit(const unsigned long long *__restrict__ prefix, const unsigned long long *__restrict__ above,
                                                  const long long *__restrict__ room, const long long *__restrict__ rank0, int nlab,
                                                  double *__restrict__ lo, double *__restrict__ hi)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlab) return;
    if (rank0[l] < 0) { lo[l] = 0.0; hi[l] = 0.0; return; }
    const double a = sel_dec(prefix[l]);
    lo[l] = a;
    hi[l] = room[l] > 0 ? a : (above[l] == ~0ULL ? a : sel_dec(above[l]));
}

// lo[l] = value of rank ranks[l] (0-based) among the pixels of label l + 1, hi[l] = value of rank ranks[l] + 1 (= lo when
// there is none); ranks[l] < 0: label absent.  labels == nullptr: nlab must be 1 (the whole frame).
int label_order_stats_dev(const int32_t *labels, const double *img, long n, int nlab, const long long *ranks_host, double *lo_host,
                          double *hi_host)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!img || !ranks_host || !lo_host || !hi_host || n < 1 || nlab < 1) return fail(TIP_ERR_ARG, "order_stats: bad arguments");
    if (!labels && nlab != 1) return fail(TIP_ERR_ARG, "order_stats: whole-frame mode takes one rank");
    WsGuard ws;
    unsigned int *hist = ws.get<unsigned int>((size_t)nlab * 256);
    unsigned long long *prefix = ws.get<unsigned long long>(nlab), *above = ws.get<unsigned long long>(nlab);
    long long *rank = ws.get<long long>(nlab), *rank0 = ws.get<long long>(nlab), *room = ws.get<long long>(nlab);
    double *lo = ws.get<double>(nlab), *hi = ws.get<double>(nlab);
    if (!hist || !prefix || !above || !rank || !rank0 || !room || !lo || !hi) return TIP_ERR_NOMEM;
    hipStream_t s = c.stream;
    TIP_HIP(hipMemsetAsync(hist, 0, (size_t)nlab * 256 * 4, s));
    TIP_HIP(hipMemsetAsync(prefix, 0, (size_t)nlab * 8, s));
    TIP_HIP(hipMemsetAsync(above, 0xff, (size_t)nlab * 8, s));
    TIP_HIP(hipMemsetAsync(room, 0, (size_t)nlab * 8, s));
    TIP_HIP(hipMemcpyAsync(rank, ranks_host, (size_t)nlab * 8, hipMemcpyHostToDevice, s));
    TIP_HIP(hipMemcpyAsync(rank0, ranks_host, (size_t)nlab * 8, hipMemcpyHostToDevice, s));
    for (int shift = 56; shift >= 0; shift -= 8) {
        TIP_LAUNCH("sel_hist", k_sel_hist, dim3(cdiv(n, 256 * 8)), dim3(256), 0, labels, img, n, nlab, (const unsigned long long *)prefix, shift,
                   hist);
        TIP_LAUNCH("sel_pick", k_sel_pick, dim3(cdiv(nlab, 256)), dim3(256), 0, hist, rank, prefix, shift, nlab, room);
    }
    TIP_LAUNCH("sel_next", k_sel_next, dim3(cdiv(n, 256)), dim3(256), 0, labels, img, n, nlab, (const unsigned long long *)prefix, above);
    TIP_LAUNCH("sel_emit", k_sel_emit, dim3(cdiv(nlab, 256)), dim3(256), 0, (const unsigned long long *)prefix, (const unsigned long long *)above,
               (const long long *)room, (const long long *)rank0, nlab, lo, hi);
    TIP_HIP(hipMemcpyAsync(lo_host, lo, (size_t)nlab * 8, hipMemcpyDeviceToHost, s));
    TIP_HIP(hipMemcpyAsync(hi_host, hi, (size_t)nlab * 8, hipMemcpyDeviceToHost, s));
    TIP_HIP(hipStreamSynchronize(s));
    return TIP_OK;
}

}  // namespace tip

using namespace tip;

extern "C" {

int tip_label_order_stats_f64(const int32_t *labels, const double *img, int y, int x, int nlab, const int64_t *ranks, double *lo,
                              double *hi)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!img || y < 1 || x < 1 || nlab < 1 || !ranks || !lo || !hi) return fail(TIP_ERR_ARG, "tip_label_order_stats_f64: bad arguments");
    const size_t P = (size_t)y * x;
    WsGuard ws;
    int32_t *dl = labels ? ws.get<int32_t>(P) : nullptr;
    double *di = ws.get<double>(P);
    if ((labels && !dl) || !di) return TIP_ERR_NOMEM;
    if (labels) TIP_HIP(hipMemcpyAsync(dl, labels, P * 4, hipMemcpyHostToDevice, c.stream));
    TIP_HIP(hipMemcpyAsync(di, img, P * 8, hipMemcpyHostToDevice, c.stream));
    return label_order_stats_dev(dl, di, (long)P, nlab, (const long long *)ranks, lo, hi);
}

}  // extern "C"
