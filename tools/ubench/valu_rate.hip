// dev microbenchmark: sustained issue rate of v_pk_fma_f32 / v_pk_add_f32 / v_fma_f32 / v_fma_f64 on gfx950.
// Build with -fno-slp-vectorize (otherwise the "scalar" mode is packed by the compiler and measures v_pk_fma_f32 again):
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wno-unused-result tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N_IT 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float a, float b)
{
    f32x2 acc[8], x[8];
    for (int i = 0; i < 8; ++i) { acc[i] = f32x2{(float)threadIdx.x, (float)i}; x[i] = f32x2{a + i, b - i}; }
    double dacc[8];
    const float va = a + 1e-6f * threadIdx.x;   // a per-lane multiplier keeps it in a VGPR
    for (int i = 0; i < 8; ++i) dacc[i] = threadIdx.x + i;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_elementwise_fma(x[i], f32x2{a, a}, acc[i]);                 // pk_fma
            if (MODE == 1) { f32x2 s = x[i] + acc[(i + 1) & 7]; acc[i] = __builtin_elementwise_fma(s, f32x2{a, a}, acc[i]); }  // pk_add + pk_fma
            if (MODE == 2) { acc[i].x = __builtin_fmaf(x[i].x, a, acc[i].x); acc[i].y = __builtin_fmaf(x[i].y, a, acc[i].y); }  // 2 x v_fma_f32
            if (MODE == 3) dacc[i] = __builtin_fma(dacc[i], (double)a, (double)b);                         // v_fma_f64
            if (MODE == 4) { dacc[i] = dacc[i] * (double)a; dacc[i] = dacc[i] + (double)b; }               // v_mul_f64 + v_add_f64
            if (MODE == 5) { acc[i].x = __builtin_fmaf(x[i].x, va, acc[i].x); acc[i].y = __builtin_fmaf(x[i].y, va, acc[i].y); }  // 2 x v_fmac_f32 (VOP2, VGPR operands)
            if (MODE == 6) { float s0 = x[i].x + acc[(i + 1) & 7].x, s1 = x[i].y + acc[(i + 1) & 7].y; acc[i].x = __builtin_fmaf(s0, va, acc[i].x); acc[i].y = __builtin_fmaf(s1, va, acc[i].y); }  // 2 x (v_add_f32 + v_fmac_f32)
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i].x + acc[i].y + (float)dacc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
void run(const char *name, double flop_per_it_lane)
{
    float *d;
    const int blocks = 256 * 16, threads = 256;
    hipMalloc(&d, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(d, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<MODE><<<blocks, threads>>>(d, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double lanes = (double)blocks * threads;
    printf("%-24s %.3f ms  %.1f TFLOP/s  (%.2f G wave-instr/s/SIMD-equivalent)\n", name, ms, lanes * N_IT * flop_per_it_lane / ms * 1e-9,
           lanes / 64 * N_IT * 8 / (ms * 1e-3) / 1024 * 1e-9);
    hipFree(d);
}
int main()
{
    run<0>("pk_fma_f32", 8 * 4);
    run<1>("pk_add+pk_fma", 8 * 6);
    run<2>("2x v_fma_f32", 8 * 4);
    run<3>("v_fma_f64", 8 * 2);
    run<4>("v_mul_f64+v_add_f64", 8 * 2);
    run<5>("2x v_fmac_f32 (VOP2)", 8 * 4);
    run<6>("2x (v_add+v_fmac) VOP2", 8 * 6);
    return 0;
}
