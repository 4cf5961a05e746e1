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
    double *a = ws.get<double>(n), *b = ws.get<double>(n), *d = ws.get<double>(n);
    if (!a || !b || !d) return TIP_ERR_NOMEM;
    if (dtype == 0)
        TIP_LAUNCH("tail_threshold", k_tail_threshold<float>, dim3(cdiv(x, 256), y), dim3(256), 0, (const float *)p, ld, y, x, (float)thr, a);
    else
        TIP_LAUNCH("tail_threshold", k_tail_threshold<double>, dim3(cdiv(x, 256), y), dim3(256), 0, (const double *)p, ld, y, x, thr, a);
    int rc;
    if ((rc = rankfilter2d_dev(a, b, 1, y, x, 5, 5, 0, 1, 1))) return rc;     // dilation 5x5, reflect
    if ((rc = rankfilter2d_dev(b, a, 1, y, x, 5, 5, 0, 1, 0))) return rc;     // erosion 5x5 -> closed
    if ((rc = rankfilter2d_dev(a, hc, 1, y, x, 7, 7, 0, 1, 0))) return rc;    // HC = erosion 7x7
    TIP_LAUNCH("tail_sub", k_tail_sub, dim3(cdiv(n, 256)), dim3(256), 0, (const double *)a, (const double *)hc, d, n);
    if ((rc = rankfilter2d_dev(d, b, 1, y, x, 5, 5, 0, 1, 1))) return rc;     // boundary = dilation 5x5
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

int tip_unet_conv_dev(const tip_unet_conv_desc *d, void *stream)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!d || !d->in0 || !d->weights || !d->bias || !d->out) return fail(TIP_ERR_ARG, "tip_unet_conv_dev: null pointer");
    if (d->planes != 2 && d->planes != 3) return fail(TIP_ERR_ARG, "tip_unet_conv_dev: planes must be 2 or 3");
    if (d->h < 8 || d->w < UC_TW || d->h % 8 || d->w % UC_TW)
        return fail(TIP_ERR_UNSUPPORTED, "tip_unet_conv_dev: the grid %dx%d is not a multiple of the 8x%d pixel tile", d->h, d->w, UC_TW);
    if (d->c0 < UC_KC || d->c0 % UC_KC || d->c1 < 0 || d->c1 % UC_KC || (d->c1 > 0 && !d->in1) || d->cout < UC_BN || d->cout % UC_BN)
        return fail(TIP_ERR_UNSUPPORTED, "tip_unet_conv_dev: channels (%d + %d -> %d) must be multiples of %d / %d", d->c0, d->c1, d->cout, UC_KC, UC_BN);
    if (d->ntaps < 1 || d->ntaps > 9 || d->sy < 1 || d->sx < 1 || d->oy < 0 || d->ox < 0 ||
        (d->h - 1) * d->sy + d->oy >= d->out_h || (d->w - 1) * d->sx + d->ox >= d->out_w)
        return fail(TIP_ERR_ARG, "tip_unet_conv_dev: bad taps / output mapping");
    if ((d->scale == nullptr) != (d->shift == nullptr)) return fail(TIP_ERR_ARG, "tip_unet_conv_dev: scale and shift come together");
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
    if (!c.zero_page) {     // source of halo pixels outside the image (the copies go global -> LDS, a register zero cannot be written)
        TIP_HIP(hipMalloc(&c.zero_page, 256));
        TIP_HIP(hipMemset(c.zero_page, 0, 256));
    }
    p.zeros = (const uint16_t *)c.zero_page;
    p.pool_out = (uint16_t *)d->pool_out;
    if (d->pool_out && (d->sy != 1 || d->sx != 1 || d->oy != 0 || d->ox != 0 || d->out_h != d->h || d->out_w != d->w))
        return fail(TIP_ERR_ARG, "tip_unet_conv_dev: pool_out needs the plain output mapping");
    // 16-row tiles (one 512-thread workgroup per CU) where the grid allows: half the weight copies per MFMA, and LDS for five
    // weight buffers (copies four steps ahead) when the stencil has >= 4 taps; two pieces only (LDS)
    const int th = (d->planes == 2 && d->h % 16 == 0 && !tuning().unet_tile8) ? 16 : 8;
    const int dist = (th == 16 && d->ntaps >= 4) ? 4 : 2;
    const int da = (th == 16 && d->ntaps <= 2) ? 2 : 1;       // one- and two-tap stencils: activation tiles two chunks ahead
    const int threads = th * 32, hp = UC_HW * (th + 2);
    const dim3 grid((d->h / th) * (d->w / UC_TW), d->cout / UC_BN);
    const int a_per = (d->planes * hp * 2 + threads - 1) / threads;
    const size_t lds = (size_t)(da + 1) * a_per * threads * 16 + (size_t)(dist + 1) * d->planes * 256 * 16;
    hipStream_t s = (hipStream_t)stream;
    static bool attr_set[5] = {false, false, false, false, false};
    const int which = d->planes == 3 ? 3 : (th == 16 ? (dist == 4 ? 2 : (da == 2 ? 4 : 1)) : 0);
    const void *fn = which == 3 ? (const void *)k_unet_conv<3, 8, 2>
                   : which == 2 ? (const void *)k_unet_conv<2, 16, 4>
                   : which == 4 ? (const void *)k_unet_conv<2, 16, 2, 2>
                   : which == 1 ? (const void *)k_unet_conv<2, 16, 2> : (const void *)k_unet_conv<2, 8, 2>;
    if (!attr_set[which]) { TIP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set[which] = true; }
    if (which == 3) hipLaunchKernelGGL((k_unet_conv<3, 8, 2>), grid, dim3(threads), lds, s, p);
    else if (which == 2) hipLaunchKernelGGL((k_unet_conv<2, 16, 4>), grid, dim3(threads), lds, s, p);
    else if (which == 4) hipLaunchKernelGGL((k_unet_conv<2, 16, 2, 2>), grid, dim3(threads), lds, s, p);
    else if (which == 1) hipLaunchKernelGGL((k_unet_conv<2, 16, 2>), grid, dim3(threads), lds, s, p);
    else hipLaunchKernelGGL((k_unet_conv<2, 8, 2>), grid, dim3(threads), lds, s, p);
    return unet_launch_check("unet_conv");
}

int tip_unet_conv_first_dev(const float *in, int h, int w, const float *wgt, const float *bias, const float *scale, const float *shift,
                            void *out, int planes, void *stream)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!in || !wgt || !bias || !scale || !shift || !out || h < 1 || w < 1 || ((long)h * w) % 16 || (planes != 2 && planes != 3))
        return fail(TIP_ERR_ARG, "tip_unet_conv_first_dev: bad arguments");
    const dim3 grid((unsigned)((long)h * w / 16));
    hipStream_t s = (hipStream_t)stream;
    if (planes == 2) hipLaunchKernelGGL(k_unet_conv_first<2>, grid, dim3(256), 0, s, in, h, w, wgt, bias, scale, shift, (uint16_t *)out);
    else hipLaunchKernelGGL(k_unet_conv_first<3>, grid, dim3(256), 0, s, in, h, w, wgt, bias, scale, shift, (uint16_t *)out);
    return unet_launch_check("unet_conv_first");
}

int tip_unet_pool2_dev(const void *in, int h, int w, int ch, int planes, void *out, void *stream)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!in || !out || h < 2 || w < 2 || (h & 1) || (w & 1) || ch < 8 || ch % 8 || (planes != 2 && planes != 3))
        return fail(TIP_ERR_ARG, "tip_unet_pool2_dev: bad arguments");
    const long n = (long)(h / 2) * (w / 2) * (ch / 8);
    hipStream_t s = (hipStream_t)stream;
    if (planes == 2) hipLaunchKernelGGL(k_unet_pool2<2>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const uint16_t *)in, h, w, ch, (uint16_t *)out);
    else hipLaunchKernelGGL(k_unet_pool2<3>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const uint16_t *)in, h, w, ch, (uint16_t *)out);
    return unet_launch_check("unet_pool2");
}

int tip_unet_head_dev(const void *in, long npix, const float *wgt, const float *bias, float *out, int planes, int logits, void *stream)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!in || !wgt || !bias || !out || npix < 1 || (planes != 2 && planes != 3)) return fail(TIP_ERR_ARG, "tip_unet_head_dev: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(npix * 8, 256));
    if (planes == 2) hipLaunchKernelGGL(k_unet_head<2>, grid, dim3(256), 0, s, (const uint16_t *)in, npix, wgt, bias, out, logits);
    else hipLaunchKernelGGL(k_unet_head<3>, grid, dim3(256), 0, s, (const uint16_t *)in, npix, wgt, bias, out, logits);
    return unet_launch_check("unet_head");
}

}  // extern "C"
