// tip_unet.hip -- epilogue of the U-Net's convolutions (pl.py:31-37: Conv2D + bias -> ReLU -> BatchNormalization at
// inference = per-channel scale and shift), fused into ONE in-place pass over the NHWC activation.
//
// The convolutions themselves stay with PyTorch-ROCm / MIOpen (north_star: PyTorch only for the U-Net's conv path).  Left
// to torch, the epilogue is four full passes over activations of up to 2.1 GB (bias add, relu, multiply, add): 14 % of the
// network's time at 2048^2.  Same float32 operations in the same order (the library is built with -ffp-contract=off, so
// the multiply and the add round separately like torch's two kernels): bit-identical, one read + one write instead of four.
// Runs on the stream the caller names (torch's current stream), so no cross-stream synchronisation is needed.
#include "tip_internal.h"
#include "tip_unet_conv.h"
#include <algorithm>
#include <atomic>

namespace tip {

__global__ void __launch_bounds__(256) k_bias_relu_affine_f32(float *__restrict__ x, const float *__restrict__ bias,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              long n4, int C)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<float4 *>(x)[i];
    const int c = (int)((i * 4) % C);                       // C % 4 == 0: the four lanes are channels c .. c+3
    const float4 b = *reinterpret_cast<const float4 *>(bias + c);
    const float4 s = *reinterpret_cast<const float4 *>(scale + c);
    const float4 t = *reinterpret_cast<const float4 *>(shift + c);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
    v.x = v.x * s.x + t.x; v.y = v.y * s.y + t.y; v.z = v.z * s.z + t.z; v.w = v.w * s.w + t.w;
    reinterpret_cast<float4 *>(x)[i] = v;
}

// ---- the post-network tail (pl.py:167-194) as one submission on the library's stream ------------------------------------
int rankfilter2d_dev(const void *in, void *out, int dtype, int Y, int X, int ky, int kx, int fp, int border, int is_max);   // tip_label.hip
int watershed_dev(const double *img, int32_t *labels, int Y, int X, int wsl, int32_t *flags_host);                          // tip_watershed.hip

// HC_B = 255 * (p > thr)  (pl.py:168), p read with a row pitch (the un-padded view of the network's output)
template <typename T>
__global__ void __launch_bounds__(256) k_tail_threshold(const T *__restrict__ p, long ld, int Y, int X, T thr, double *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x < X) out[(long)y * X + x] = p[(long)y * ld + x] > thr ? 255.0 : 0.0;
}

__global__ void __launch_bounds__(256) k_tail_sub(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}


// ---- U1: prepare_image (pl.py:21-29, 90-122) on a device-resident (C, A, B) float64 image ----------------------------------------
// normalize_channel clips every channel to its [1st, 99th] percentile and scales to [0, 1]; prepare_image transposes to (B, A) and
// pads in front to the network's extents.  The percentiles are np.percentile's: order statistics k, k + 1 of the channel plus
// numpy's lerp.  They come from a most-significant-digit radix select over sortable 64-bit keys (eight byte-wide passes, both
// ranks of every channel in one launch; the same scheme as tip_select.hip) -- no sort -- and never leave the device: the
// normalise / transpose / pad pass reads them from memory.
constexpr int PREP_MAXC = 8;
struct PrepState {                      // per (channel, which percentile): [c][0] = the 1st, [c][1] = the 99th
    unsigned int hist[PREP_MAXC][2][256];
    unsigned long long prefix[PREP_MAXC][2], above[PREP_MAXC][2];
    long long rank[PREP_MAXC][2], room[PREP_MAXC][2];
    double per[PREP_MAXC][2], clipv[PREP_MAXC][2];      // percentile values; the values written over the clipped pixels
};

__device__ __forceinline__ unsigned long long prep_enc(double d)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(d + 0.0);   // (-0.0 sorts with +0.0, like numpy's <)
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double prep_dec(unsigned long long e)
{
    unsigned long long b = (e >> 63) ? (e & 0x7fffffffffffffffULL) : ~e;
    return __longlong_as_double((long long)b);
}

__global__ void __launch_bounds__(64) k_prep_reset(PrepState *st, int C, long long r1, long long r99)
{
    const int t = threadIdx.x;
    for (int i = t; i < C * 2 * 256; i += 64) (&st->hist[0][0][0])[i] = 0;
    if (t < C * 2) {
        const int c = t >> 1, w = t & 1;
        st->prefix[c][w] = 0; st->above[c][w] = ~0ULL; st->room[c][w] = 0;
        st->rank[c][w] = w ? r99 : r1;
    }
}

// one digit of both selects of channel blockIdx.y; the plane is dense (n consecutive doubles from img + c * cstride)
__global__ void __launch_bounds__(256) k_prep_hist(const double *__restrict__ img, long cstride, long n, PrepState *st, int shift)
{
    __shared__ unsigned int sh[2][256];
    const int c = blockIdx.y;
    sh[0][threadIdx.x] = 0; sh[1][threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long p0 = st->prefix[c][0], p1 = st->prefix[c][1];
    const double *src = img + (long)c * cstride;
    const long i0 = (long)blockIdx.x * 2048 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long i = i0 + u * 256;
        if (i < n) {
            const unsigned long long key = prep_enc(src[i]);
            const int bin = (int)((key >> shift) & 255ULL);
            const bool m0 = shift == 56 || (key >> (shift + 8)) == (p0 >> (shift + 8));
            const bool m1 = shift == 56 || (key >> (shift + 8)) == (p1 >> (shift + 8));
            if (m0) atomicAdd(&sh[0][bin], 1u);
            if (m1) atomicAdd(&sh[1][bin], 1u);
        }
    }
    __syncthreads();
    if (sh[0][threadIdx.x]) atomicAdd(&st->hist[c][0][threadIdx.x], sh[0][threadIdx.x]);
    if (sh[1][threadIdx.x]) atomicAdd(&st->hist[c][1][threadIdx.x], sh[1][threadIdx.x]);
}

// the bin holding the rank: prefix gets the digit, rank becomes the rank inside the bin; on the last digit `room` = how many more
// copies of the selected key follow (so that rank + 1 can be answered)
__global__ void __launch_bounds__(64) k_prep_pick(PrepState *st, int C, int shift)
{
    const int t = threadIdx.x;
    if (t >= C * 2) return;
    const int c = t >> 1, w = t & 1;
    unsigned int *h = st->hist[c][w];
    const long long r = st->rank[c][w];
    long long cum = 0;
    int pick = 255;
    for (int b = 0; b < 256; ++b) {
        const long long cnt = h[b];
        if (cum + cnt > r) { pick = b; break; }
        cum += cnt;
    }
    if (shift == 0) st->room[c][w] = (long long)h[pick] - (r - cum) - 1;
    st->prefix[c][w] |= (unsigned long long)pick << shift;
    st->rank[c][w] = r - cum;
    for (int b = 0; b < 256; ++b) h[b] = 0;
}

// smallest key strictly above each selected one
__global__ void __launch_bounds__(256) k_prep_next(const double *__restrict__ img, long cstride, long n, PrepState *st)
{
    const int c = blockIdx.y;
    const unsigned long long p0 = st->prefix[c][0], p1 = st->prefix[c][1];
    unsigned long long a0 = ~0ULL, a1 = ~0ULL;
    const double *src = img + (long)c * cstride;
    for (long i = (long)blockIdx.x * 2048 + threadIdx.x, e = min(n, ((long)blockIdx.x + 1) * 2048); i < e; i += 256) {
        const unsigned long long key = prep_enc(src[i]);
        if (key > p0 && key < a0) a0 = key;
        if (key > p1 && key < a1) a1 = key;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o0 = __shfl_xor(a0, d, 64), o1 = __shfl_xor(a1, d, 64);
        a0 = o0 < a0 ? o0 : a0;
        a1 = o1 < a1 ? o1 : a1;
    }
    if ((threadIdx.x & 63) == 0) {
        if (a0 != ~0ULL) atomicMin(&st->above[c][0], a0);
        if (a1 != ~0ULL) atomicMin(&st->above[c][1], a1);
    }
}

// np.percentile's 'linear' lerp (numpy/lib/function_base.py _lerp: lo + diff * g, and hi - diff * (1 - g) from g >= 0.5) and the
// values normalize_channel writes over the clipped pixels, which take the INPUT's dtype (pl.py:26-27 assign into a copy of the
// image): kind 0 = float64 (as is), 1 = float32 (rounded), 2 = integer (truncated)
__global__ void __launch_bounds__(64) k_prep_finish(PrepState *st, int C, double g1, double g99, int kind)
{
    const int t = threadIdx.x;
    if (t >= C * 2) return;
    const int c = t >> 1, w = t & 1;
    const double lo = prep_dec(st->prefix[c][w]);
    const double hi = st->room[c][w] > 0 ? lo : (st->above[c][w] == ~0ULL ? lo : prep_dec(st->above[c][w]));
    const double g = w ? g99 : g1;
    const double diff = hi - lo;
    double res = lo + diff * g;
    if (g >= 0.5) res = hi - diff * (1.0 - g);
    st->per[c][w] = res;
    st->clipv[c][w] = kind == 2 ? trunc(res) : (kind == 1 ? (double)(float)res : res);
}

// out[c][pb + b][pa + a] = normalised in[c][a][b]; A_CONTIG: the input's a index has unit stride (threads run along a on both
// sides), else its b index has (a 32 x 32 tile turns through LDS)
template <bool A_CONTIG>
__global__ void __launch_bounds__(256) k_prep_normalize(const double *__restrict__ img, long cstride, long sa, long sb, int A, int B,
                                                        const PrepState *__restrict__ st, int single, float *__restrict__ out, int Ap,
                                                        int Bp, int pa, int pb)
{
    __shared__ float tile[32][33];
    const int c = blockIdx.z, a0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const double per1 = st->per[c][0], per99 = st->per[c][1], lo = st->clipv[c][0], hi = st->clipv[c][1];
    const double den = per99 - per1;
    const float per1f = (float)per1, denf = (float)den;
    const double *src = img + (long)c * cstride;
    float *dst = out + (long)c * Ap * Bp;
    auto norm = [&](double v) -> float {
        double cl = v > per99 ? hi : v;
        cl = v < per1 ? lo : cl;
        if (single) return ((float)cl - per1f) / denf;       // numpy 1.x: float32 array (op) float64 scalar stays float32
        return (float)((cl - per1) / den);
    };
    if (A_CONTIG) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = a0 + tx, b = b0 + ty + k * 8;
            if (a < A && b < B) dst[(long)(pb + b) * Ap + pa + a] = norm(src[(long)a * sa + (long)b * sb]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = a0 + ty + k * 8, b = b0 + tx;
            if (a < A && b < B) tile[ty + k * 8][tx] = norm(src[(long)a * sa + (long)b * sb]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = a0 + tx, b = b0 + ty + k * 8;
            if (a < A && b < B) dst[(long)(pb + b) * Ap + pa + a] = tile[tx][ty + k * 8];
        }
    }
}


// The tail's whole morphology (pl.py:168-193) in ONE kernel.  Every image of the chain is two-valued, so it runs on bytes in LDS:
//   A = p > thr -> B = dilation 5x5 -> closed = erosion 5x5 -> HC = erosion 7x7 -> D = closed - HC -> boundary = dilation 5x5.
// skimage / scipy filter with 'reflect' borders; a symmetric window on a reflect-extended image yields the reflect-extension of
// the filtered image, so the chain of reflect-bordered filters equals the chain on the reflect-extended input: a block loads its
// 64 x 64 tile with a 9-pixel halo (2 + 2 + 3 + 2) through reflected coordinates and runs the separable passes on shrinking
// margins.  As four generic float64 rank-filter launches + a subtraction this was 0.58 ms per 2048^2 frame.
constexpr int TM_T = 64, TM_H = 9, TM_W = TM_T + 2 * TM_H, TM_P = TM_W + 2;
template <typename T>
__global__ void __launch_bounds__(256) k_tail_morph(const T *__restrict__ p, long ld, int Y, int X, T thr, double *__restrict__ hc,
                                                    double *__restrict__ bd)
{
    __shared__ unsigned char s0[TM_W][TM_P], s1[TM_W][TM_P], s2[TM_W][TM_P];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TM_T - TM_H, y0 = blockIdx.y * TM_T - TM_H;
    auto refl = [](int i, int n) {                 // scipy 'reflect' (d c b a | a b c d | d c b a), any overshoot
        const int period = 2 * n;
        i %= period;
        if (i < 0) i += period;
        return i < n ? i : period - 1 - i;
    };
    for (int i = tid; i < TM_W * TM_W; i += 256) {
        const int r = i / TM_W, c = i - r * TM_W;
        s0[r][c] = p[(long)refl(y0 + r, Y) * ld + refl(x0 + c, X)] > thr ? 1 : 0;
    }
    __syncthreads();
    // separable window passes over the largest region whose window fits the array (what they compute from cells outside the valid
    // margin never reaches the interior: the margins add up to the halo)
    auto row_pass = [&](unsigned char (*src)[TM_P], unsigned char (*dst)[TM_P], int rad, bool is_max) {
        const int wc = TM_W - 2 * rad;
        for (int i = tid; i < TM_W * wc; i += 256) {
            const int r = i / wc, c = rad + i - r * wc;
            unsigned v = src[r][c - rad];
            for (int d = -rad + 1; d <= rad; ++d) v = is_max ? (v | src[r][c + d]) : (v & src[r][c + d]);
            dst[r][c] = (unsigned char)v;
        }
        __syncthreads();
    };
    auto col_pass = [&](unsigned char (*src)[TM_P], unsigned char (*dst)[TM_P], int rad, bool is_max) {
        const int hr = TM_W - 2 * rad;
        for (int i = tid; i < hr * TM_W; i += 256) {
            const int r = rad + i / TM_W, c = i % TM_W;
            unsigned v = src[r - rad][c];
            for (int d = -rad + 1; d <= rad; ++d) v = is_max ? (v | src[r + d][c]) : (v & src[r + d][c]);
            dst[r][c] = (unsigned char)v;
        }
        __syncthreads();
    };
    row_pass(s0, s1, 2, true);  col_pass(s1, s0, 2, true);      // B = dilation 5x5            (valid margin 2)
    row_pass(s0, s1, 2, false); col_pass(s1, s2, 2, false);     // closed = erosion 5x5 in s2  (4)
    row_pass(s2, s1, 3, false); col_pass(s1, s0, 3, false);     // HC = erosion 7x7 in s0      (7)
    for (int i = tid; i < TM_W * TM_W; i += 256) {              // D = closed - HC in s1; HC out
        const int r = i / TM_W, c = i - r * TM_W;
        s1[r][c] = s2[r][c] & (s0[r][c] ^ 1);                  // (erosion never exceeds its input: closed - HC is 255 or 0)
        const int y = y0 + r, x = x0 + c;
        if (r >= TM_H && r < TM_H + TM_T && c >= TM_H && c < TM_H + TM_T && y < Y && x < X) hc[(long)y * X + x] = s0[r][c] ? 255.0 : 0.0;
    }
    __syncthreads();
    row_pass(s1, s2, 2, true);  col_pass(s2, s0, 2, true);      // boundary = dilation 5x5 in s0 (9)
    for (int i = tid; i < TM_T * TM_T; i += 256) {
        const int r = TM_H + i / TM_T, c = TM_H + i % TM_T;
        const int y = y0 + r, x = x0 + c;
        if (y < Y && x < X) bd[(long)y * X + x] = s0[r][c] ? 255.0 : 0.0;
    }
}

}  // namespace tip

using namespace tip;

extern "C" {

// x: n float32 values of a channels-last (NHWC-contiguous) activation with C channels, updated in place:
// x = relu(x + bias[c]) * scale[c] + shift[c].  `stream`: the hipStream_t to launch on, taken as is -- 0 is HIP's null
// stream, which is what torch.cuda.current_stream() is unless the caller changed it.
int tip_bias_relu_affine_f32_dev(float *x, const float *bias, const float *scale, const float *shift, long n, int c, void *stream)
{
    Ctx &cx = ctx();
    if (!cx.stream) return TIP_ERR_HIP;
    if (!x || !bias || !scale || !shift || n < 0 || c < 4 || (c & 3) || (n % c)) return fail(TIP_ERR_ARG, "bias_relu_affine: bad arguments");
    if (n == 0) return TIP_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bias_relu_affine_f32, dim3(cdiv(n / 4, 256)), dim3(256), 0, s, x, bias, scale, shift, n / 4, c);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TIP_ERR_HIP, "launch bias_relu_affine: %s", hipGetErrorString(e));
    return TIP_OK;
}

// Which device does this runtime think `p` lives on?  >= 0: the ordinal; negative: `p` is not a device pointer THIS copy of
// the HIP runtime knows.  A caller that hands over torch pointers and torch's stream (tip_bias_relu_affine_f32_dev,
// tip_unet_tail_dev) checks once that torch and this library share one runtime: PyTorch wheels bundle their own
// libamdhip64, and two loaded runtimes do not know each other's allocations or stream handles.
int tip_pointer_device(const void *p)
{
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return TIP_ERR_ARG; }
    if (a.type != hipMemoryTypeDevice) return TIP_ERR_ARG;
    return a.device;
}

// Ordering edges between the calling thread's library stream and a foreign HIP stream (torch's current stream).  Neither
// blocks the host.  tip_wait_stream: work submitted to the library AFTER the call starts after everything queued on
// `stream` BEFORE the call -- call it after allocating (from torch's caching allocator) every buffer the library is going
// to write: the allocator hands out blocks whose previous owner's kernels may still be queued on that stream.
int tip_wait_stream(void *stream)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!c.edge_event) TIP_HIP(hipEventCreateWithFlags(&c.edge_event, hipEventDisableTiming));
    TIP_HIP(hipEventRecord(c.edge_event, (hipStream_t)stream));
    TIP_HIP(hipStreamWaitEvent(c.stream, c.edge_event, 0));
    return TIP_OK;
}

// ... and the other direction: `stream` waits for everything submitted to the library so far.
int tip_stream_wait_tip(void *stream)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!c.edge_event) TIP_HIP(hipEventCreateWithFlags(&c.edge_event, hipEventDisableTiming));
    TIP_HIP(hipEventRecord(c.edge_event, c.stream));
    TIP_HIP(hipStreamWaitEvent((hipStream_t)stream, c.edge_event, 0));
    return TIP_OK;
}

// pl.py:167-194 on a device-resident class-0 probability map p (y rows of x values, row pitch ld elements; dtype 0 =
// float32, 1 = float64): HC_B = 255 (p > thr) -> 5x5 closing (the reference's 101 iterations are idempotent) -> HC = 7x7
// erosion -> boundary = 5x5 dilation of (closed - HC) -> watershed(boundary, watershed_line=True).  labels (int32) and hc
// (float64) are caller-owned device buffers of y * x elements.  Everything runs on the library's stream in library
// workspaces.  The boundary image is {0, 255} by construction; anything else means a corrupted intermediate and is an
// error, not a slow flood.
int tip_unet_tail_dev(const void *p, int dtype, long ld, int y, int x, double thr, int32_t *labels, double *hc, int32_t *flags_host)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!p || !labels || !hc || y < 1 || x < 1 || ld < x || (dtype != 0 && dtype != 1)) return fail(TIP_ERR_ARG, "tip_unet_tail_dev: bad arguments");
    const long n = (long)y * x;
    WsGuard ws;
    double *b = ws.get<double>(n);
    if (!b) return TIP_ERR_NOMEM;
    const dim3 mgrid(cdiv(x, TM_T), cdiv(y, TM_T));
    if (tuning().unet_tail_unfused) {        // the same chain as separate launches (tests compare the two)
        double *a = ws.get<double>(n), *d = ws.get<double>(n);
        if (!a || !d) return TIP_ERR_NOMEM;
        if (dtype == 0)
            TIP_LAUNCH("tail_threshold", k_tail_threshold<float>, dim3(cdiv(x, 256), y), dim3(256), 0, (const float *)p, ld, y, x, (float)thr, a);
        else
            TIP_LAUNCH("tail_threshold", k_tail_threshold<double>, dim3(cdiv(x, 256), y), dim3(256), 0, (const double *)p, ld, y, x, thr, a);
        int rc0;
        if ((rc0 = rankfilter2d_dev(a, b, 1, y, x, 5, 5, 0, 1, 1))) return rc0;     // dilation 5x5, reflect
        if ((rc0 = rankfilter2d_dev(b, a, 1, y, x, 5, 5, 0, 1, 0))) return rc0;     // erosion 5x5 -> closed
        if ((rc0 = rankfilter2d_dev(a, hc, 1, y, x, 7, 7, 0, 1, 0))) return rc0;    // HC = erosion 7x7
        TIP_LAUNCH("tail_sub", k_tail_sub, dim3(cdiv(n, 256)), dim3(256), 0, (const double *)a, (const double *)hc, d, n);
        if ((rc0 = rankfilter2d_dev(d, b, 1, y, x, 5, 5, 0, 1, 1))) return rc0;     // boundary = dilation 5x5
    } else if (dtype == 0) {
        TIP_LAUNCH("tail_morph", k_tail_morph<float>, mgrid, dim3(256), 0, (const float *)p, ld, y, x, (float)thr, hc, b);
    } else {
        TIP_LAUNCH("tail_morph", k_tail_morph<double>, mgrid, dim3(256), 0, (const double *)p, ld, y, x, thr, hc, b);
    }
    int rc;
    int32_t flags = 0;
    if ((rc = watershed_dev(b, labels, y, x, 1, &flags))) return rc;
    if (flags_host) *flags_host = flags;
    if (c.last_ws_other != 0)
        return fail(TIP_ERR_HIP, "tip_unet_tail_dev: the boundary image is not two-valued (%ld other values): corrupted intermediate", c.last_ws_other);
    return TIP_OK;
}

// ---- the network's layers on split bf16 planes (tip_unet_conv.h).  All of them launch on the stream the caller names
// (torch's current stream: the buffers are torch tensors), like tip_bias_relu_affine_f32_dev. -----------------------------------
static int unet_launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TIP_ERR_HIP, "launch %s: %s", what, hipGetErrorString(e));
    return TIP_OK;
}

// pl.py:90-122 + 21-29 for a device-resident image: img = (c, a, b) float64 with element strides (cstride, sa, sb), every channel
// plane dense (sa == 1 && sb == a, or sb == 1 && sa == b); kind = dtype of the ORIGINAL image (0 float64, 1 float32, 2 integer:
// normalize_channel's clip values take it).  out = (c, bp, ap) float32, zero-filled in front: out[c][bp - b + j][ap - a + i] =
// normalised img[c][i][j].  Launches on `stream` (torch's current stream: the buffers are torch tensors) and returns at once.
int tip_unet_prepare_f64_dev(const double *img, int c, int a, int b, long cstride, long sa, long sb, int kind, float *out, int ap, int bp,
                             void *stream)
{
    Ctx &cx = ctx();
    if (!cx.stream) return TIP_ERR_HIP;
    if (!img || !out || c < 1 || c > PREP_MAXC || a < 1 || b < 1 || ap < a || bp < b || kind < 0 || kind > 2)
        return fail(TIP_ERR_ARG, "tip_unet_prepare_f64_dev: bad arguments");
    const bool a_contig = sa == 1 && sb == a, b_contig = sb == 1 && sa == b;
    if (!a_contig && !b_contig) return fail(TIP_ERR_UNSUPPORTED, "tip_unet_prepare_f64_dev: every channel plane must be dense");
    if (!cx.prep_ws) TIP_HIP(hipMalloc(&cx.prep_ws, sizeof(PrepState)));
    PrepState *st = (PrepState *)cx.prep_ws;
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)a * b;
    // numpy: virtual index (n - 1) q, previous = floor, gamma = the fraction (function_base.py _quantile / _lerp)
    auto split = [&](double q, long long &prev, double &g) {
        const double virt = (double)(n - 1) * q;
        prev = (long long)floor(virt);
        g = virt - (double)prev;
        if (prev < 0) prev = 0;
        if (prev > n - 1) prev = n - 1;
    };
    long long r1, r99;
    double g1, g99;
    split(1.0 / 100.0, r1, g1);
    split(99.0 / 100.0, r99, g99);
    hipLaunchKernelGGL(k_prep_reset, dim3(1), dim3(64), 0, s, st, c, r1, r99);
    const dim3 hgrid((unsigned)cdiv(n, 2048), (unsigned)c);
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(k_prep_hist, hgrid, dim3(256), 0, s, img, cstride, n, st, shift);
        hipLaunchKernelGGL(k_prep_pick, dim3(1), dim3(64), 0, s, st, c, shift);
    }
    hipLaunchKernelGGL(k_prep_next, hgrid, dim3(256), 0, s, img, cstride, n, st);
    hipLaunchKernelGGL(k_prep_finish, dim3(1), dim3(64), 0, s, st, c, g1, g99, kind);
    if (ap != a || bp != b) TIP_HIP(hipMemsetAsync(out, 0, (size_t)c * ap * bp * sizeof(float), s));
    const dim3 ngrid((unsigned)cdiv(a, 32), (unsigned)cdiv(b, 32), (unsigned)c);
    if (a_contig)
        hipLaunchKernelGGL(k_prep_normalize<true>, ngrid, dim3(256), 0, s, img, cstride, sa, sb, a, b, (const PrepState *)st, kind == 1 ? 1 : 0, out, ap, bp, ap - a, bp - b);
    else
        hipLaunchKernelGGL(k_prep_normalize<false>, ngrid, dim3(256), 0, s, img, cstride, sa, sb, a, b, (const PrepState *)st, kind == 1 ? 1 : 0, out, ap, bp, ap - a, bp - b);
    return unet_launch_check("unet_prepare");
}

int tip_unet_conv_dev(const tip_unet_conv_desc *d, void *stream)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!d || !d->in0 || !d->weights || !d->bias || (!d->out && !d->head_out)) return fail(TIP_ERR_ARG, "tip_unet_conv_dev: null pointer");
    if (d->planes != 2 && d->planes != 3) return fail(TIP_ERR_ARG, "tip_unet_conv_dev: planes must be 2 or 3");
    if (d->format != 0 && d->format != 1) return fail(TIP_ERR_ARG, "tip_unet_conv_dev: format is 0 (bf16 pieces) or 1 (fp16 pieces)");
    if (d->format == 1 && (d->planes != 2 || !(d->acc_scale > 0.f))) return fail(TIP_ERR_ARG, "tip_unet_conv_dev: fp16 pieces come in two planes with a positive acc_scale");
    if (d->h < 8 || d->w < UC_TW || d->h % 8 || d->w % UC_TW)
        return fail(TIP_ERR_UNSUPPORTED, "tip_unet_conv_dev: the grid %dx%d is not a multiple of the 8x%d pixel tile", d->h, d->w, UC_TW);
    if (d->c0 < UC_KC || d->c0 % UC_KC || d->c1 < 0 || d->c1 % UC_KC || (d->c1 > 0 && !d->in1) || d->cout < UC_BN || d->cout % UC_BN)
        return fail(TIP_ERR_UNSUPPORTED, "tip_unet_conv_dev: channels (%d + %d -> %d) must be multiples of %d / %d", d->c0, d->c1, d->cout, UC_KC, UC_BN);
    if (d->ntaps < 1 || d->ntaps > 9 || d->sy < 1 || d->sx < 1 || d->oy < 0 || d->ox < 0 ||
        (d->h - 1) * d->sy + d->oy >= d->out_h || (d->w - 1) * d->sx + d->ox >= d->out_w)
        return fail(TIP_ERR_ARG, "tip_unet_conv_dev: bad taps / output mapping");
    if ((d->scale == nullptr) != (d->shift == nullptr)) return fail(TIP_ERR_ARG, "tip_unet_conv_dev: scale and shift come together");
    // a tile's halo window ((tile rows + 2) image rows of one plane) is addressed through 32-bit buffer offsets; tensors may be any size
    if (18L * d->w * (d->c0 > d->c1 ? d->c0 : d->c1) * 2 >= (1L << 32) - 65536)
        return fail(TIP_ERR_UNSUPPORTED, "tip_unet_conv_dev: 18 rows of %d pixels x %d channels exceed the 4 GB a buffer resource addresses",
                    d->w, d->c0 > d->c1 ? d->c0 : d->c1);
    ConvParams p;
    p.in0 = (const uint16_t *)d->in0; p.in1 = (const uint16_t *)(d->c1 > 0 ? d->in1 : d->in0);
    p.c0 = d->c0; p.c1 = d->c1; p.H = d->h; p.W = d->w;
    p.w = (const uint16_t *)d->weights; p.ntaps = d->ntaps;
    for (int t = 0; t < 9; ++t) {
        p.dy[t] = t < d->ntaps ? d->dy[t] : 0; p.dx[t] = t < d->ntaps ? d->dx[t] : 0;
        if (p.dy[t] < -1 || p.dy[t] > 1 || p.dx[t] < -1 || p.dx[t] > 1) return fail(TIP_ERR_ARG, "tip_unet_conv_dev: tap offsets are -1, 0 or 1");
    }
    p.cout = d->cout; p.bias = d->bias; p.scale = d->scale; p.shift = d->shift;
    p.out = (uint16_t *)d->out; p.outH = d->out_h; p.outW = d->out_w; p.sy = d->sy; p.sx = d->sx; p.oy = d->oy; p.ox = d->ox;
    p.pool_out = (uint16_t *)d->pool_out;
    p.acc_scale = d->format == 1 ? d->acc_scale : 1.f;
    for (int t = 0; t < 9; ++t) p.nmask[t] = d->tf ? (d->nmask[t] & 15) : 15;
    if (d->tf && (d->planes != 2 || d->ntaps != 4 || d->sy != 2 || d->sx != 2 || d->oy != 0 || d->ox != 0 || d->scale || d->pool_out || d->head_out ||
                  d->out_h != 2 * d->h || d->out_w != 2 * d->w))
        return fail(TIP_ERR_ARG, "tip_unet_conv_dev: the fused transposed convolution takes four offset taps, two planes, bias only, a 2h x 2w output");
#ifdef UC_TRACE
    static unsigned long long *trace_dev = nullptr;
    if (!trace_dev) { TIP_HIP(hipMalloc(&trace_dev, 1024)); }
    TIP_HIP(hipMemsetAsync(trace_dev, 0, 1024, (hipStream_t)stream));
    p.trace = trace_dev;
#endif
    p.head_w = d->head_w; p.head_b = d->head_b; p.head_out = d->head_out;
    if (d->head_out && (!d->head_w || !d->head_b || d->cout != UC_BN || !d->scale || d->pool_out || d->sy != 1 || d->sx != 1 || d->oy != 0 || d->ox != 0 ||
                        d->out_h != d->h || d->out_w != d->w))
        return fail(TIP_ERR_ARG, "tip_unet_conv_dev: the fused head needs a 128-channel Conv2D + BatchNorm layer with the plain output mapping");
    if (d->pool_out && (d->sy != 1 || d->sx != 1 || d->oy != 0 || d->ox != 0 || d->out_h != d->h || d->out_w != d->w))
        return fail(TIP_ERR_ARG, "tip_unet_conv_dev: pool_out needs the plain output mapping");
    // 16-row tiles (one 512-thread workgroup per CU) where the grid allows: half the weight copies per MFMA, and LDS for five
    // weight buffers (copies four steps ahead) when the stencil has >= 4 taps; two pieces only (LDS)
    const bool tf = d->tf != 0;
    const int t8 = tuning().unet_tile8;
    // (measured per layer at 2048^2: the short K loops of the 128-channel 3x3 layers gain 1-3 % from two workgroups per CU -- one's
    // epilogue behind the other's products -- every other layer is faster with the shared weight tile of the 16-row workgroup)
    const bool want8 = t8 == 1 || (t8 < 0 && d->ntaps == 9 && d->c0 + d->c1 <= 128);
    const int th = (d->planes == 2 && d->h % 16 == 0 && !want8) ? 16 : 8;
    const int dist = (th == 16 && d->ntaps >= 4) ? 4 : 2;
    const int da = (th == 16 && d->ntaps <= 2) ? 2 : 1;       // one- and two-tap stencils: activation tiles two chunks ahead
    const int threads = th * 32, hp = UC_HW * (th + 2);
    const int ntiles = (d->h / th) * (d->w / UC_TW), nblks = d->cout / UC_BN;
    p.xcd_map = (tuning().unet_xcd_map && nblks > 1 && ntiles % 8 == 0) ? 1 : 0;
    const dim3 grid = p.xcd_map ? dim3((unsigned)(ntiles * nblks)) : dim3(ntiles, nblks);
    const int a_per = (d->planes * ((hp * 2 + 63) / 64) * 64 + threads - 1) / threads;     // (a plane's halo tile padded to whole waves, as in the kernel)
    // several steps per barrier: 3x3 stencils on 16-row tiles; three (a chunk's nine taps in three iterations, nine weight buffers)
    // or two (six buffers; needs an even number of steps)
    const int spb_want = tuning().unet_spb;
    const bool multi = th == 16 && dist == 4 && d->ntaps == 9 && spb_want > 1;
    const int spb = !multi ? 1 : (spb_want >= 3 ? 3 : ((((d->c0 + d->c1) / UC_KC) & 1) == 0 ? 2 : 1));
    const size_t lds = (size_t)(da + 1) * a_per * threads * 16 + (size_t)(spb > 1 ? 3 * spb : dist + 1) * d->planes * 256 * 16;
    hipStream_t s = (hipStream_t)stream;
    const int which = tf ? (th == 16 ? 7 : 8)
                         : d->planes == 3 ? 3 : (th == 16 ? (dist == 4 ? (spb == 3 ? 6 : (spb == 2 ? 5 : 2)) : (da == 2 ? 4 : 1)) : 0);
    // the >64 KB dynamic LDS attribute is set once per (device, kernel): one atomic bit each
    static std::atomic<unsigned> attr_done[64];
    const unsigned bit = 1u << (which + (d->format ? 9 : 0));
    const int dev = c.device >= 0 && c.device < 64 ? c.device : 0;
#define UC_FLAVOUR(W, ...)                                                                                                                  \
    case W: {                                                                                                                               \
        if (!(attr_done[dev].load(std::memory_order_acquire) & bit)) {                                                                      \
            TIP_HIP(hipFuncSetAttribute((const void *)(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                \
            attr_done[dev].fetch_or(bit, std::memory_order_release);                                                                        \
        }                                                                                                                                   \
        hipLaunchKernelGGL((__VA_ARGS__), grid, dim3(threads), lds, s, p);                                                                  \
    } break;
    if (d->format == 0) {
        switch (which) {
            UC_FLAVOUR(3, k_unet_conv<3, 8, 2>)
            UC_FLAVOUR(6, k_unet_conv<2, 16, 4, 1, 3>)
            UC_FLAVOUR(5, k_unet_conv<2, 16, 4, 1, 2>)
            UC_FLAVOUR(2, k_unet_conv<2, 16, 4>)
            UC_FLAVOUR(4, k_unet_conv<2, 16, 2, 2>)
            UC_FLAVOUR(1, k_unet_conv<2, 16, 2>)
            UC_FLAVOUR(0, k_unet_conv<2, 8, 2>)
            UC_FLAVOUR(7, k_unet_conv<2, 16, 4, 1, 1, false, true>)
            UC_FLAVOUR(8, k_unet_conv<2, 8, 2, 1, 1, false, true>)
        }
    } else {
        switch (which) {
            UC_FLAVOUR(6, k_unet_conv<2, 16, 4, 1, 3, true>)
            UC_FLAVOUR(5, k_unet_conv<2, 16, 4, 1, 2, true>)
            UC_FLAVOUR(2, k_unet_conv<2, 16, 4, 1, 1, true>)
            UC_FLAVOUR(4, k_unet_conv<2, 16, 2, 2, 1, true>)
            UC_FLAVOUR(1, k_unet_conv<2, 16, 2, 1, 1, true>)
            UC_FLAVOUR(0, k_unet_conv<2, 8, 2, 1, 1, true>)
            UC_FLAVOUR(7, k_unet_conv<2, 16, 4, 1, 1, true, true>)
            UC_FLAVOUR(8, k_unet_conv<2, 8, 2, 1, 1, true, true>)
        }
    }
#undef UC_FLAVOUR
#ifdef UC_TRACE
    {
        unsigned long long tr[128];
        TIP_HIP(hipStreamSynchronize(s));
        TIP_HIP(hipMemcpy(tr, trace_dev, sizeof tr, hipMemcpyDeviceToHost));
        for (int k = 0; k < th / 2; ++k) {
            const unsigned long long *t = tr + 16 * k;
            const double n = (double)t[4];
            fprintf(stderr, "UC_TRACE th %d spb %d taps %d cin %d cout %d grid %dx%d wave %d simd %d: per STEP (shader clocks; sums over the loop / steps): copies %.0f, products %.0f (first MFMA out after %.0f), "
                            "vmcnt wait %.0f, barrier %.0f, total %.0f; steps %.0f; whole loop %.0f, prologue %.0f, epilogue %.0f clocks\n",
                    th, spb, d->ntaps, d->c0 + d->c1, d->cout, d->h, d->w, k, (int)((t[7] >> 4) & 3), t[0] / n, t[2] / n, t[1] / n, t[6] / n, (t[3] - t[6]) / n, t[5] / n, n, (double)t[5], (double)t[8], (double)t[9]);
        }
    }
#endif
    return unet_launch_check("unet_conv");
}

int tip_unet_conv_first_dev(const float *in, int h, int w, const float *wgt, const float *bias, const float *scale, const float *shift,
                            void *out, int planes, int format, void *stream)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!in || !wgt || !bias || !scale || !shift || !out || h < 1 || w < 1 || w % FIRST_RUN || ((long)h * w) % FIRST_PIX || (planes != 2 && planes != 3) ||
        (format != 0 && !(format == 1 && planes == 2)))
        return fail(TIP_ERR_ARG, "tip_unet_conv_first_dev: bad arguments (w must be a multiple of 32, h * w of 256)");
    const dim3 grid((unsigned)((long)h * w / FIRST_PIX));
    hipStream_t s = (hipStream_t)stream;
    if (format == 1) hipLaunchKernelGGL((k_unet_conv_first<2, true>), grid, dim3(256), 0, s, in, h, w, wgt, bias, scale, shift, (uint16_t *)out);
    else if (planes == 2) hipLaunchKernelGGL(k_unet_conv_first<2>, grid, dim3(256), 0, s, in, h, w, wgt, bias, scale, shift, (uint16_t *)out);
    else hipLaunchKernelGGL(k_unet_conv_first<3>, grid, dim3(256), 0, s, in, h, w, wgt, bias, scale, shift, (uint16_t *)out);
    return unet_launch_check("unet_conv_first");
}

int tip_unet_pool2_dev(const void *in, int h, int w, int ch, int planes, int format, void *out, void *stream)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!in || !out || h < 2 || w < 2 || (h & 1) || (w & 1) || ch < 8 || ch % 8 || (planes != 2 && planes != 3) || (format != 0 && !(format == 1 && planes == 2)))
        return fail(TIP_ERR_ARG, "tip_unet_pool2_dev: bad arguments");
    const long n = (long)(h / 2) * (w / 2) * (ch / 8);
    hipStream_t s = (hipStream_t)stream;
    if (format == 1) hipLaunchKernelGGL((k_unet_pool2<2, true>), dim3(cdiv(n, 256)), dim3(256), 0, s, (const uint16_t *)in, h, w, ch, (uint16_t *)out);
    else if (planes == 2) hipLaunchKernelGGL(k_unet_pool2<2>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const uint16_t *)in, h, w, ch, (uint16_t *)out);
    else hipLaunchKernelGGL(k_unet_pool2<3>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const uint16_t *)in, h, w, ch, (uint16_t *)out);
    return unet_launch_check("unet_pool2");
}

int tip_unet_head_dev(const void *in, long npix, const float *wgt, const float *bias, float *out, int planes, int format, int logits, void *stream)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!in || !wgt || !bias || !out || npix < 1 || (planes != 2 && planes != 3) || (format != 0 && !(format == 1 && planes == 2)))
        return fail(TIP_ERR_ARG, "tip_unet_head_dev: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(npix * 8, 256));
    if (format == 1) hipLaunchKernelGGL((k_unet_head<2, true>), grid, dim3(256), 0, s, (const uint16_t *)in, npix, wgt, bias, out, logits);
    else if (planes == 2) hipLaunchKernelGGL(k_unet_head<2>, grid, dim3(256), 0, s, (const uint16_t *)in, npix, wgt, bias, out, logits);
    else hipLaunchKernelGGL(k_unet_head<3>, grid, dim3(256), 0, s, (const uint16_t *)in, npix, wgt, bias, out, logits);
    return unet_launch_check("unet_head");
}

}  // extern "C"
