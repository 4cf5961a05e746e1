// mfma_f16_rounding.hip -- where does v_mfma_f32_32x32x16_f16 round?  The certified argmax of the projection (csrc/tip_project.hip)
// bounds the error of a sum of NON-NEGATIVE terms by the number of roundings a term can pass through, so it needs to know how the
// matrix core adds the 16 products of one instruction to its accumulator operand.  Every row of A and every column of B is the same
// here, so all outputs are equal; the products are exact powers of two chosen so that different summation structures give
// different float32 results:
//   C = 1, sixteen products of 2^-e each:  one rounding per product leaves 1 for e >= 25;  exact groups of g products rounded
//   once per group add g 2^-e whenever that reaches half an ulp of 1 (2^-24);  one exact sum of all 16 products adds 2^(4-e).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_f16_rounding.hip -o tools/ubench/mfma_f16_rounding
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Case { float a[16], b[16], c; };

__global__ void k_probe(const Case *cs, float *out, int ncase)
{
    const int h = threadIdx.x >> 5;
    for (int t = 0; t < ncase; ++t) {
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)cs[t].a[8 * h + e]; b[e] = (_Float16)cs[t].b[8 * h + e]; }
        f32x16 acc;
        for (int q = 0; q < 16; ++q) acc[q] = cs[t].c;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (threadIdx.x == 0) out[t] = acc[0];
    }
}

int main()
{
    const int NC = 64;
    Case hc[NC];
    char names[NC][96];
    int n = 0;
    auto add = [&](const char *name, float c, auto prod) {      // prod(k) -> exponent e of product 2^-e (or 0 for the value 1, <0: none)
        for (int k = 0; k < 16; ++k) {
            const int e = prod(k);
            if (e < 0) { hc[n].a[k] = 0.f; hc[n].b[k] = 0.f; }
            else if (e == 0) { hc[n].a[k] = 1.f; hc[n].b[k] = 1.f; }
            else { hc[n].a[k] = ldexpf(1.f, -(e / 2)); hc[n].b[k] = ldexpf(1.f, -(e - e / 2)); }
        }
        hc[n].c = c;
        snprintf(names[n], sizeof names[n], "%s", name);
        ++n;
    };
    add("C=1, 16 x 2^-25", 1.f, [](int) { return 25; });
    add("C=1, 16 x 2^-26", 1.f, [](int) { return 26; });
    add("C=1, 16 x 2^-27", 1.f, [](int) { return 27; });
    add("C=1, 16 x 2^-28", 1.f, [](int) { return 28; });
    add("C=1,  8 x 2^-26 (k 0..7)", 1.f, [](int k) { return k < 8 ? 26 : -1; });
    add("C=1,  8 x 2^-26 (k 8..15)", 1.f, [](int k) { return k >= 8 ? 26 : -1; });
    add("C=1,  8 x 2^-26 (even k)", 1.f, [](int k) { return k % 2 == 0 ? 26 : -1; });
    add("C=1,  4 x 2^-25 (k 0..3)", 1.f, [](int k) { return k < 4 ? 25 : -1; });
    add("C=1,  4 x 2^-25 (k 0,4,8,12)", 1.f, [](int k) { return k % 4 == 0 ? 25 : -1; });
    add("C=1,  2 x 2^-25 (k 0,1)", 1.f, [](int k) { return k < 2 ? 25 : -1; });
    add("C=1,  2 x 2^-25 (k 0,8)", 1.f, [](int k) { return k == 0 || k == 8 ? 25 : -1; });
    for (int pos = 0; pos < 16; pos += 5) {
        char nm[96];
        snprintf(nm, sizeof nm, "C=0, product 1 at k=%d, 15 x 2^-27", pos);
        add(nm, 0.f, [pos](int k) { return k == pos ? 0 : 27; });
    }
    add("C=2^-27 x 1 ... C tiny, product 1 at k=0, 15 x 2^-27", ldexpf(1.f, -27), [](int k) { return k == 0 ? 0 : 27; });
    Case *dc;
    float *dout, hout[NC];
    (void)hipMalloc(&dc, sizeof hc);
    (void)hipMalloc(&dout, sizeof hout);
    (void)hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dc, dout, n);
    (void)hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
    for (int t = 0; t < n; ++t) {
        const float base = hc[t].c >= 1.f ? 1.f : 1.f;
        printf("%-56s -> %.9g  = 1 + %g ulp\n", names[t], hout[t], (hout[t] - base) / ldexpf(1.f, -23));
    }
    return 0;
}
