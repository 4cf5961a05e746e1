// mfma_f16_rate.hip -- the fp16 twin of mfma_bf16_rate.hip (11-bit significands toggle more multiplier bits than bf16's 8: does the
// power-limited clock differ?) plus a check that fp16 SUBNORMAL operands are multiplied, not flushed (the f16x3 split of
// csrc/tip_unet_conv.h relies on it).  What the fp16 matrix pipe sustains on this chip and at which shader clock: back-to-back
// v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 from registers only (no LDS, no memory), 1 / 2 waves per SIMD, eight
// independent accumulators per wave.  Wave 0 of block 0 reads s_memtime (shader clock ticks) and s_memrealtime (100 MHz) around
// its loop: their ratio is the clock the CU really ran at under this load.  Operand values: 0 = all-zero bits, 1 = random bits
// (data-dependent power).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_f16_rate.hip -o /tmp/mfma_f16_rate && /tmp/mfma_f16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>   // 0: 32x32x16 (8 accumulators of 16 registers), 1: 16x16x32 (8 accumulators of 4 registers)
__global__ void __launch_bounds__(256) k_mfma(float *out, unsigned long long *clk, int iters, const uint4 *seed)
{
    const uint4 sa = seed[threadIdx.x & 63], sb = seed[64 + (threadIdx.x & 63)];
    const f16x8 a = __builtin_bit_cast(f16x8, sa), b = __builtin_bit_cast(f16x8, sb);
    f32x16 acc[8];
    f32x4 acd[8];
    for (int c = 0; c < 8; ++c) {
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
        for (int q = 0; q < 4; ++q) acd[c][q] = 0.f;
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (SHAPE == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
                else acd[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acd[c], 0, 0, 0);
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int c = 0; c < 8; ++c) {
        for (int q = 0; q < 16; ++q) s += acc[c][q];
        for (int q = 0; q < 4; ++q) s += acd[c][q];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int SHAPE>
static void run(int blocks_per_cu, int cus, int iters, int random_bits)
{
    float *out;
    unsigned long long *clk, hclk[2];
    uint4 *seed, hseed[128];
    const int blocks = blocks_per_cu * cus;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipMalloc(&clk, 16);
    hipMalloc(&seed, sizeof hseed);
    srand(1);
    for (int i = 0; i < 128; ++i) {
        unsigned w[4];
        for (int j = 0; j < 4; ++j) {
            // random mantissas and signs with exponents near 1.0 (finite, no denormals): ~ random fp16 pairs
            const unsigned lo = 0x3800u | (rand() & 0x87ffu), hi = 0x3800u | (rand() & 0x87ffu);
            w[j] = random_bits ? (lo | (hi << 16)) : 0u;
        }
        hseed[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    hipMemcpy(seed, hseed, sizeof hseed, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, clk, 100, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, seed);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    const double per = SHAPE == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    const double flop = (double)blocks * 4 * (double)iters * 24 * per;
    const double mhz = (double)hclk[0] / ((double)hclk[1] / 100.0);    // shader ticks per microsecond of the 100 MHz counter
    const double cyc_per_mfma = (double)hclk[0] / ((double)iters * 24 * blocks_per_cu);   // per SIMD: blocks_per_cu waves share it
    printf("%s, %d wave(s)/SIMD, %s operands: %.3f ms, %.0f TFLOP/s (%.1f %% of 2500), shader clock %.0f MHz, %.1f clocks per MFMA per SIMD\n",
           SHAPE == 0 ? "32x32x16" : "16x16x32", blocks_per_cu, random_bits ? "random" : "zero", ms, flop / ms / 1e9,
           100.0 * flop / ms / 1e9 / 2500.0, mhz, cyc_per_mfma);
    hipFree(out); hipFree(clk); hipFree(seed);
}

// one MFMA with A = 2^-20 (an fp16 subnormal: bits 0x0010) in every element and B = 1024: each output is 16 x 2^-10 = 2^-6 unless
// subnormal inputs are flushed (then 0); and the conversion float -> fp16 of 2^-20 must give 0x0010, not 0
__global__ void k_denorm(float *out, unsigned *cvt)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = __builtin_bit_cast(_Float16, (unsigned short)0x0010); b[i] = (_Float16)1024.0f; }
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    out[threadIdx.x] = acc[0];
    volatile float tiny = 9.5367431640625e-07f;     // 2^-20
    const _Float16 h = (_Float16)tiny;
    if (threadIdx.x == 0) cvt[0] = __builtin_bit_cast(unsigned short, h);
}

int main()
{
    {
        float *o, ho[64];
        unsigned *c, hc = 0;
        hipMalloc(&o, 256); hipMalloc(&c, 4);
        hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, o, c);
        hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost);
        hipMemcpy(&hc, c, 4, hipMemcpyDeviceToHost);
        printf("fp16 subnormal operand (2^-20) x 1024 summed over K = 16: %.9g (expected 0.015625 = 2^-6; 0 would mean flushed inputs); "
               "v_cvt_f16_f32(2^-20) = 0x%04x (expected 0x0010)\n", ho[0], hc);
    }
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clockRate %d kHz\n", p.name, cus, p.clockRate);
    for (int rnd = 0; rnd < 2; ++rnd) {
        run<0>(1, cus, 20000, rnd);
        run<0>(2, cus, 10000, rnd);
        run<1>(1, cus, 40000, rnd);
        run<1>(2, cus, 20000, rnd);
    }
    // a long run: does the clock sag as the chip warms up?
    run<0>(2, cus, 100000, 1);
    return 0;
}
