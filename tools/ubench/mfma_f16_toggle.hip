// mfma_f16_toggle.hip -- how much of the matrix pipe's power goes into CHANGING operands?  mfma_f16_rate.hip multiplies the same two
// register operands over and over (the multiplier inputs never toggle); a convolution brings a new operand with every instruction.
// Same loop (back-to-back v_mfma_f32_32x32x16_f16 from registers, 2 waves per SIMD, random fp16 bits), four A and four B operands held in
// registers, and the order in which the 16 (A, B) pairs of an iteration are issued:
//   0  one pair only            (a0 b0 sixteen times: the rate micro-benchmark)
//   1  B fixed, A cycles        (a0 b0, a1 b0, a2 b0, a3 b0, ...)
//   2  both change every time   (a0 b0, a1 b1, a2 b2, a3 b3, a0 b1, ...)
//   3  outer product, A-major   (a0 b0, a0 b1, a0 b2, a0 b3, a1 b0, ...: one operand changes per instruction)
// and operand statistics: dense random / every second K element of A zero / the three low mantissa bits of A zero.
// The chip runs at its power cap, so the rate IS the energy per instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_f16_toggle.hip -o /tmp/mfma_f16_toggle && /tmp/mfma_f16_toggle
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int ORDER>
__global__ void __launch_bounds__(256) k_mfma(float *out, unsigned long long *clk, int iters, const uint4 *seed)
{
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(f16x8, seed[i * 64 + (threadIdx.x & 63)]);
        b[i] = __builtin_bit_cast(f16x8, seed[(4 + i) * 64 + (threadIdx.x & 63)]);
    }
    f32x16 acc[8];
    for (int c = 0; c < 8; ++c)
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int ia = ORDER == 0 ? 0 : ORDER == 1 ? (u & 3) : ORDER == 2 ? (u & 3) : (u >> 2);
            const int ib = ORDER == 0 ? 0 : ORDER == 1 ? 0 : ORDER == 2 ? ((u + (u >> 2)) & 3) : (u & 3);
            acc[u & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[ib], acc[u & 7], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int c = 0; c < 8; ++c)
        for (int q = 0; q < 16; ++q) s += acc[c][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

static const char *ORDER_NAME[4] = {"one pair only", "B fixed, A cycles", "both change", "outer product (one changes)"};
static const char *STAT_NAME[3] = {"dense random", "every second K element of A zero", "three low mantissa bits of A zero"};

template <int ORDER>
static void run(int cus, int iters, int stat)
{
    float *out;
    unsigned long long *clk, hclk[2];
    uint4 *seed, hseed[8 * 64];
    const int blocks = 2 * cus;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipMalloc(&clk, 16);
    hipMalloc(&seed, sizeof hseed);
    srand(1);
    for (int i = 0; i < 8 * 64; ++i) {
        unsigned w[4];
        for (int j = 0; j < 4; ++j) {
            unsigned lo = 0x3800u | (rand() & 0x87ffu), hi = 0x3800u | (rand() & 0x87ffu);
            if (i < 4 * 64) {       // A operands
                if (stat == 1) hi = 0;
                if (stat == 2) { lo &= 0xfff8u; hi &= 0xfff8u; }
            }
            w[j] = lo | (hi << 16);
        }
        hseed[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    hipMemcpy(seed, hseed, sizeof hseed, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<ORDER>, dim3(blocks), dim3(256), 0, 0, out, clk, 100, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<ORDER>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, seed);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    const double flop = (double)blocks * 4 * (double)iters * 16 * 2.0 * 32 * 32 * 16;
    const double mhz = (double)hclk[0] / ((double)hclk[1] / 100.0);
    printf("%-30s %-36s %8.3f ms  %5.0f TFLOP/s  shader clock %4.0f MHz  %.1f clocks per MFMA per SIMD\n", ORDER_NAME[ORDER], STAT_NAME[stat], ms,
           flop / ms / 1e9, mhz, (double)hclk[0] / ((double)iters * 16 * 2));
    hipFree(out); hipFree(clk); hipFree(seed);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs; v_mfma_f32_32x32x16_f16, 2 waves per SIMD, registers only\n", p.name, cus);
    for (int stat = 0; stat < 3; ++stat) {
        run<0>(cus, 15000, stat);
        run<1>(cus, 15000, stat);
        run<2>(cus, 15000, stat);
        run<3>(cus, 15000, stat);
    }
    return 0;
}
