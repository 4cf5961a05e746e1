// mfma_rate.hip -- what the fp32 matrix pipe sustains on this chip: back-to-back v_mfma_f32_32x32x2_f32 from registers
// only (no LDS, no memory), 1 / 2 / 4 waves per SIMD, one or two independent accumulator chains per wave.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) k_mfma(float *out, int iters, float a0, float b0)
{
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int q = 0; q < 16; ++q) s += acc[c][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
static void run(int blocks_per_cu, int cus, int iters)
{
    float *out;
    const int blocks = blocks_per_cu * cus;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * (double)iters * 8 * CHAINS * 4096.0;   // 4 waves per block, 2 * 32 * 32 * 2 flop per MFMA
    printf("chains %d, %d waves/SIMD: %.3f ms, %.1f TFLOP/s (%.1f %% of 157.3)\n", CHAINS, blocks_per_cu, ms, flop / ms / 1e9,
           100.0 * flop / ms / 1e9 / 157.3);
    hipFree(out);
}

// co-issue probe: 512-thread blocks, one per CU; waves 0-3 (one per SIMD) run the MFMA chain, waves 4-7 (their SIMD partners)
// run `valu_per_mfma` dependent-free v_fma_f32 per MFMA slot (0 = partners idle): does vector work on the same SIMD cost
// matrix throughput?
__global__ void __launch_bounds__(512) k_mix(float *out, int iters, int valu_iters, float a0, float b0)
{
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc;
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        float a = a0 + threadIdx.x * 1e-6f, b = b0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        for (int q = 0; q < 16; ++q) s += acc[q];
    } else {
        float x[8];
        for (int q = 0; q < 8; ++q) x[q] = a0 + q + threadIdx.x;
        for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = __builtin_fmaf(x[q], b0, a0);
        }
        for (int q = 0; q < 8; ++q) s += x[q];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static void run_mix(int cus, int iters, int valu_iters)
{
    float *out;
    hipMalloc(&out, (size_t)cus * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mix, dim3(cus), dim3(512), 0, 0, out, 10, 10, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix, dim3(cus), dim3(512), 0, 0, out, iters, valu_iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)cus * 4 * (double)iters * 8 * 4096.0, vf = (double)cus * 4 * 64.0 * (double)valu_iters * 64 * 2;
    printf("mix: MFMA iters %d, VALU iters %d: %.3f ms -> MFMA %.1f TFLOP/s if it alone set the time, VALU %.1f TFLOP/s\n", iters, valu_iters,
           ms, mf / ms / 1e9, vf / ms / 1e9);
    hipFree(out);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, %d MHz\n", p.name, cus, p.clockRate / 1000);
    for (int w : {1, 2, 4}) { run<1>(w, cus, 40000 / w); run<2>(w, cus, 20000 / w); }
    // a long run: does the rate sag with time (power / clock)?
    run<2>(2, cus, 400000);
    // MFMA waves alone (one per SIMD), then with a VALU partner wave doing 64 v_fma per 8 MFMAs (8 per MFMA), 128, 256
    run_mix(cus, 40000, 0);
    run_mix(cus, 40000, 40000);        // 64 fma per 8 MFMAs: 256 of 512 cycles of vector issue
    run_mix(cus, 40000, 80000);
    run_mix(cus, 0, 40000);            // VALU alone
    return 0;
}
