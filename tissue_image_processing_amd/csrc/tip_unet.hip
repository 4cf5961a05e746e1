// tip_unet.hip -- epilogue of the U-Net's convolutions (pl.py:31-37: Conv2D + bias -> ReLU -> BatchNormalization at
// inference = per-channel scale and shift), fused into ONE in-place pass over the NHWC activation.
//
// The convolutions themselves stay with PyTorch-ROCm / MIOpen (north_star: PyTorch only for the U-Net's conv path).  Left
// to torch, the epilogue is four full passes over activations of up to 2.1 GB (bias add, relu, multiply, add): 14 % of the
// network's time at 2048^2.  Same float32 operations in the same order (the library is built with -ffp-contract=off, so
// the multiply and the add round separately like torch's two kernels): bit-identical, one read + one write instead of four.
// Runs on the stream the caller names (torch's current stream), so no cross-stream synchronisation is needed.
#include "tip_internal.h"

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

}  // extern "C"
