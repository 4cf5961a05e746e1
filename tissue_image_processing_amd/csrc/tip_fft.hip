// tip_fft.hip -- global drift between two frames: skimage.registration.phase_cross_correlation(upsample_factor)
// (reference call sites ti.py:1976-1977, 2029-2030 via update_drift / calculate_refine_drift, and bim.py:522-536).
//
//   F1 = fft2(ref), F2 = fft2(mov); P = F1 * conj(F2); whole-pixel peak = argmax |ifft2(P)|;
//   refinement = matrix-multiply upsampled DFT of conj(P) on a ceil(1.5*upsample)^2 grid around the peak
//   (skimage/registration/_phase_cross_correlation.py:11-76, 196-262; skimage 0.18.3 applies no normalisation).
// Everything in float64 (the reference hands uint16 / float64 frames to scipy.fft -> complex128).  The library returns
// the two integer peaks; the host turns them into the shift with numpy's own arithmetic.  Hand-written radix-2 FFT:
// power-of-two extents up to 4096 (one block per row, the row in LDS), columns through a tiled transpose.
#include "tip_internal.h"

namespace tip {

typedef double2 cplx;

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

template <typename T>
__global__ void __launch_bounds__(256) k_to_complex(const T *__restrict__ in, cplx *__restrict__ out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_double2((double)in[i], 0.0);
}

__global__ void __launch_bounds__(256) k_twiddles(cplx *__restrict__ w, int N)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < N / 2) {
        double s, c;
        sincospi(-2.0 * (double)k / (double)N, &s, &c);
        w[k] = make_double2(c, s);
    }
}

// in-place FFT of every row (length N, power of two, N <= 4096); inverse = conjugate twiddles (no scaling)
__global__ void __launch_bounds__(256) k_fft_rows(cplx *__restrict__ data, const cplx *__restrict__ tw, int N, int logN, int inverse)
{
    extern __shared__ __attribute__((aligned(16))) double2 row[];
    cplx *p = data + (long)blockIdx.x * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const int j = (int)(__brev((unsigned)i) >> (32 - logN));
        row[j] = p[i];
    }
    __syncthreads();
    for (int s = 1; s <= logN; ++s) {
        const int half = 1 << (s - 1);
        const int step = N >> s;  // twiddle stride
        for (int t = threadIdx.x; t < N / 2; t += blockDim.x) {
            const int grp = t / half, pos = t - grp * half;
            const int i0 = grp * (half << 1) + pos, i1 = i0 + half;
            cplx w = tw[pos * step];
            if (inverse) w.y = -w.y;
            const cplx a = row[i0], b = cmul(row[i1], w);
            row[i0] = make_double2(a.x + b.x, a.y + b.y);
            row[i1] = make_double2(a.x - b.x, a.y - b.y);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < N; i += blockDim.x) p[i] = row[i];
}

// ---- arbitrary lengths: Bluestein's chirp-z on top of the power-of-two butterflies --------------------------------------
// X[k] = b[k] * sum_n (x[n] b[n]) conj(b[k-n]),  b[n] = exp(-i pi n^2 / N): a length-N DFT as a circular convolution of
// length M = 2^ceil(log2(2N-1)), done in LDS per row: forward DIF (natural in, bit-reversed out), pointwise product with
// the chirp's transform (stored in the same bit-reversed order), inverse DIT (bit-reversed in, natural out).
__global__ void __launch_bounds__(256) k_chirp(cplx *__restrict__ b, int N)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const long r = ((long)n * n) % (2L * N);          // n^2 mod 2N exactly: the phase only matters modulo 2 pi
    double sn, cs;
    sincospi(-(double)r / (double)N, &sn, &cs);
    b[n] = make_double2(cs, sn);
}

__device__ __forceinline__ void lds_dif(cplx *row, const cplx *__restrict__ tw, int M, int logM)
{
    for (int s = logM; s >= 1; --s) {
        const int half = 1 << (s - 1), step = M >> s;
        for (int t = threadIdx.x; t < M / 2; t += blockDim.x) {
            const int grp = t / half, pos = t - grp * half;
            const int i0 = grp * (half << 1) + pos, i1 = i0 + half;
            const cplx a = row[i0], c = row[i1];
            row[i0] = make_double2(a.x + c.x, a.y + c.y);
            row[i1] = cmul(make_double2(a.x - c.x, a.y - c.y), tw[pos * step]);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void lds_dit_inverse(cplx *row, const cplx *__restrict__ tw, int M, int logM)
{
    for (int s = 1; s <= logM; ++s) {
        const int half = 1 << (s - 1), step = M >> s;
        for (int t = threadIdx.x; t < M / 2; t += blockDim.x) {
            const int grp = t / half, pos = t - grp * half;
            const int i0 = grp * (half << 1) + pos, i1 = i0 + half;
            cplx w = tw[pos * step];
            w.y = -w.y;
            const cplx a = row[i0], c = cmul(row[i1], w);
            row[i0] = make_double2(a.x + c.x, a.y + c.y);
            row[i1] = make_double2(a.x - c.x, a.y - c.y);
        }
        __syncthreads();
    }
}

// transform of the wrapped conjugate chirp, left in DIF (bit-reversed) order
__global__ void __launch_bounds__(256) k_bluestein_filter(cplx *__restrict__ cf, const cplx *__restrict__ b, const cplx *__restrict__ tw,
                                                          int N, int M, int logM)
{
    extern __shared__ __attribute__((aligned(16))) double2 row[];
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        cplx v = make_double2(0.0, 0.0);
        if (m < N) v = make_double2(b[m].x, -b[m].y);
        else if (m > M - N) v = make_double2(b[M - m].x, -b[M - m].y);
        row[m] = v;
    }
    __syncthreads();
    lds_dif(row, tw, M, logM);
    for (int m = threadIdx.x; m < M; m += blockDim.x) cf[m] = row[m];
}

// in-place DFT of every row of length N (any N with 2N-1 <= M); inverse = conjugate in, conjugate out (no scaling)
__global__ void __launch_bounds__(256) k_fft_rows_bluestein(cplx *__restrict__ data, const cplx *__restrict__ tw,
                                                            const cplx *__restrict__ b, const cplx *__restrict__ cf, int N, int M,
                                                            int logM, int inverse)
{
    extern __shared__ __attribute__((aligned(16))) double2 row[];
    cplx *p = data + (long)blockIdx.x * N;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        cplx v = make_double2(0.0, 0.0);
        if (i < N) {
            v = p[i];
            if (inverse) v.y = -v.y;
            v = cmul(v, b[i]);
        }
        row[i] = v;
    }
    __syncthreads();
    lds_dif(row, tw, M, logM);
    for (int i = threadIdx.x; i < M; i += blockDim.x) row[i] = cmul(row[i], cf[i]);
    __syncthreads();
    lds_dit_inverse(row, tw, M, logM);
    const double sc = 1.0 / (double)M;
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        cplx v = cmul(row[k], b[k]);
        v.x *= sc; v.y *= sc;
        if (inverse) v.y = -v.y;
        p[k] = v;
    }
}

__global__ void __launch_bounds__(256) k_transpose_c(const cplx *__restrict__ in, cplx *__restrict__ out, int rows, int cols)
{
    __shared__ double2 t[16][17];
    const int bx = blockIdx.x * 16, by = blockIdx.y * 16;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    if (by + ly < rows && bx + lx < cols) t[ly][lx] = in[(long)(by + ly) * cols + bx + lx];
    __syncthreads();
    if (bx + ly < cols && by + lx < rows) out[(long)(bx + ly) * rows + by + lx] = t[lx][ly];
}

__global__ void __launch_bounds__(256) k_cmul_conj(const cplx *__restrict__ a, const cplx *__restrict__ b, cplx *__restrict__ out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const cplx x = a[i], y = b[i]; out[i] = make_double2(x.x * y.x + x.y * y.y, x.y * y.x - x.x * y.y); }
}

// first maximum of |z| in raster order (np.argmax): pack (|z|^2 bits, ~index) and take the max
__global__ void __launch_bounds__(256) k_absargmax(const cplx *__restrict__ z, long n, unsigned long long *__restrict__ best_v,
                                                   unsigned long long *__restrict__ best_i)
{
    // two passes would be cleaner; a single one with a (value, index) lexicographic atomic is enough here:
    // |z|^2 >= 0, so its IEEE bits order like the value
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long key = 0;
    if (i < n) {
        const double m = hypot(z[i].x, z[i].y);   // np.abs of complex128 is hypot
        key = (unsigned long long)__double_as_longlong(m);
    }
    unsigned long long wmax = key;
    for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(wmax, d, 64); wmax = o > wmax ? o : wmax; }
    if ((threadIdx.x & 63) == 0) atomicMax(best_v, wmax);
    (void)best_i;
}
__global__ void __launch_bounds__(256) k_absargmax2(const cplx *__restrict__ z, long n, const unsigned long long *__restrict__ best_v,
                                                    unsigned long long *__restrict__ best_i)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double m = hypot(z[i].x, z[i].y);
    if ((unsigned long long)__double_as_longlong(m) == *best_v) atomicMin(best_i, (unsigned long long)i);
}

// K[u][k] = exp(-2 pi i (u - off) * fftfreq(N, ups)[k])   (skimage _upsampled_dft kernel)
__global__ void __launch_bounds__(256) k_dft_kernel(cplx *__restrict__ K, int region, int N, double off, double ups)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)region * N) return;
    const int u = (int)(i / N), k = (int)(i - (long)u * N);
    const int kk = k < (N + 1) / 2 ? k : k - N;                 // numpy fftfreq ordering
    const double val = 1.0 / ((double)N * ups);                 // numpy.fft.fftfreq: integer results * (1 / (n * d))
    const double f = (double)kk * val;
    const double arg = ((double)u - off) * f;                    // kernel = (arange - off)[:, None] * fftfreq
    double s, c;
    sincos(-2.0 * 3.141592653589793 * arg, &s, &c);              // np.exp(-1j * 2 * pi * kernel)
    K[i] = make_double2(c, s);
}

// C1[u][j] = sum_k Kx[u][k] * conj(conj(PT[k][j])) ... data = conj(P): C1[u][j] = sum_k Kx[u][k] * conj(P[j][k]);
// PT is P transposed (Nx rows of Ny), so the k loop walks rows of PT and j is contiguous.
__global__ void __launch_bounds__(256) k_updft1(const cplx *__restrict__ Kx, const cplx *__restrict__ PT, cplx *__restrict__ C1, int region,
                                                int Nx, int Ny)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, u = blockIdx.y;
    if (j >= Ny) return;
    double ax = 0.0, ay = 0.0;
    const cplx *kr = Kx + (long)u * Nx;
    for (int k = 0; k < Nx; ++k) {
        const cplx w = kr[k];
        cplx d = PT[(long)k * Ny + j];
        d.y = -d.y;                                             // conj(P)
        ax += w.x * d.x - w.y * d.y;
        ay += w.x * d.y + w.y * d.x;
    }
    C1[(long)u * Ny + j] = make_double2(ax, ay);
}

// out[v][u] = conj( sum_j Ky[v][j] * C1[u][j] )
__global__ void __launch_bounds__(64) k_updft2(const cplx *__restrict__ Ky, const cplx *__restrict__ C1, cplx *__restrict__ out, int region,
                                               int Ny)
{
    const int u = blockIdx.x, v = blockIdx.y;
    double ax = 0.0, ay = 0.0;
    for (int j = threadIdx.x; j < Ny; j += 64) {
        const cplx w = Ky[(long)v * Ny + j], d = C1[(long)u * Ny + j];
        ax += w.x * d.x - w.y * d.y;
        ay += w.x * d.y + w.y * d.x;
    }
    for (int d = 32; d >= 1; d >>= 1) { ax += __shfl_xor(ax, d, 64); ay += __shfl_xor(ay, d, 64); }
    if (threadIdx.x == 0) out[(long)v * region + u] = make_double2(ax, -ay);
}

static int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

// per-length tables: power of two -> twiddles only; any other length -> Bluestein (chirp, filter transform, twiddles of M)
struct RowPlan {
    int N = 0, M = 0, logM = 0;
    bool pow2 = true;
    cplx *tw = nullptr, *chirp = nullptr, *cf = nullptr;
};

static int make_plan(RowPlan &pl, int N, WsGuard &ws)
{
    pl.N = N;
    pl.pow2 = (N & (N - 1)) == 0;
    pl.M = N;
    if (!pl.pow2) { pl.M = 1; while (pl.M < 2 * N - 1) pl.M <<= 1; }
    pl.logM = ilog2(pl.M);
    pl.tw = ws.get<cplx>(pl.M / 2 + 1);
    if (!pl.tw) return TIP_ERR_NOMEM;
    TIP_LAUNCH("twiddles", k_twiddles, dim3(cdiv(pl.M / 2, 256)), dim3(256), 0, pl.tw, pl.M);
    if (!pl.pow2) {
        pl.chirp = ws.get<cplx>(N);
        pl.cf = ws.get<cplx>(pl.M);
        if (!pl.chirp || !pl.cf) return TIP_ERR_NOMEM;
        TIP_LAUNCH("chirp", k_chirp, dim3(cdiv(N, 256)), dim3(256), 0, pl.chirp, N);
        TIP_HIP(hipFuncSetAttribute((const void *)k_bluestein_filter, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        TIP_LAUNCH("bluestein_filter", k_bluestein_filter, dim3(1), dim3(256), (size_t)pl.M * sizeof(cplx), pl.cf, (const cplx *)pl.chirp,
                   (const cplx *)pl.tw, N, pl.M, pl.logM);
    }
    return TIP_OK;
}

static int fft_rows(cplx *data, int nrows, const RowPlan &pl, int inverse)
{
    if (pl.pow2) {
        TIP_HIP(hipFuncSetAttribute((const void *)k_fft_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        TIP_LAUNCH("fft_rows", k_fft_rows, dim3(nrows), dim3(256), (size_t)pl.N * sizeof(cplx), data, (const cplx *)pl.tw, pl.N, pl.logM,
                   inverse);
    } else {
        TIP_HIP(hipFuncSetAttribute((const void *)k_fft_rows_bluestein, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        TIP_LAUNCH("fft_rows_bluestein", k_fft_rows_bluestein, dim3(nrows), dim3(256), (size_t)pl.M * sizeof(cplx), data,
                   (const cplx *)pl.tw, (const cplx *)pl.chirp, (const cplx *)pl.cf, pl.N, pl.M, pl.logM, inverse);
    }
    return TIP_OK;
}

// rows (length Nx), transpose, rows (length Ny) -> result transposed (Nx rows of Ny) in tmp; optionally transposed back into a
static int fft2_inplace(cplx *a, cplx *tmp, int Ny, int Nx, const RowPlan &px, const RowPlan &py, int inverse, bool leave_transposed)
{
    int rc;
    if ((rc = fft_rows(a, Ny, px, inverse))) return rc;
    TIP_LAUNCH("transpose_c", k_transpose_c, dim3(cdiv(Nx, 16), cdiv(Ny, 16)), dim3(256), 0, (const cplx *)a, tmp, Ny, Nx);
    if ((rc = fft_rows(tmp, Nx, py, inverse))) return rc;
    if (!leave_transposed)
        TIP_LAUNCH("transpose_c", k_transpose_c, dim3(cdiv(Ny, 16), cdiv(Nx, 16)), dim3(256), 0, (const cplx *)tmp, a, Nx, Ny);
    return TIP_OK;
}

template <typename T>
static int to_complex(const void *in, cplx *out, long n)
{
    TIP_LAUNCH("to_complex", k_to_complex<T>, dim3(cdiv(n, 256)), dim3(256), 0, (const T *)in, out, n);
    return TIP_OK;
}

// dtype: 0 f32, 1 f64, 3 u16.  out4 (host): coarse peak (row, col) and fine peak (row, col) on the upsampled grid
int phase_correlation_dev(const void *ref, const void *mov, int dtype, int Ny, int Nx, int upsample, int64_t *out4_host)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!ref || !mov || !out4_host) return fail(TIP_ERR_ARG, "phase_correlation: null pointer");
    if (Ny < 2 || Nx < 2 || Ny > 4096 || Nx > 4096)
        return fail(TIP_ERR_UNSUPPORTED, "phase_correlation: extents must lie in [2, 4096] (got %dx%d)", Ny, Nx);
    if (upsample < 1 || upsample > 1000) return fail(TIP_ERR_ARG, "phase_correlation: upsample_factor %d", upsample);
    const long n = (long)Ny * Nx;
    const int region = upsample > 1 ? (int)ceil(upsample * 1.5) : 0;
    WsGuard ws;
    cplx *A = ws.get<cplx>(n), *B = ws.get<cplx>(n), *T1 = ws.get<cplx>(n), *T2 = ws.get<cplx>(n);
    unsigned long long *best = ws.get<unsigned long long>(4);
    if (!A || !B || !T1 || !T2 || !best) return TIP_ERR_NOMEM;
    int rc;
    RowPlan plx, ply;
    if ((rc = make_plan(plx, Nx, ws)) || (rc = make_plan(ply, Ny, ws))) return rc;
    for (int w = 0; w < 2; ++w) {
        const void *src = w == 0 ? ref : mov;
        cplx *dst = w == 0 ? A : B;
        if (dtype == 0) rc = to_complex<float>(src, dst, n);
        else if (dtype == 1) rc = to_complex<double>(src, dst, n);
        else if (dtype == 3) rc = to_complex<uint16_t>(src, dst, n);
        else return fail(TIP_ERR_ARG, "phase_correlation: dtype %d (0 f32, 1 f64, 3 u16)", dtype);
        if (rc) return rc;
    }
    if ((rc = fft2_inplace(A, T1, Ny, Nx, plx, ply, 0, true))) return rc;   // T1 = F1^T
    if ((rc = fft2_inplace(B, T2, Ny, Nx, plx, ply, 0, true))) return rc;   // T2 = F2^T
    cplx *PT = A;                                                            // P^T = F1^T * conj(F2^T)
    TIP_LAUNCH("cmul_conj", k_cmul_conj, dim3(cdiv(n, 256)), dim3(256), 0, (const cplx *)T1, (const cplx *)T2, PT, n);
    // cross-correlation = ifft2(P): inverse transform of P^T (Nx rows of Ny) -> rows Ny-point, transpose, rows Nx-point
    TIP_HIP(hipMemcpyAsync(B, PT, n * sizeof(cplx), hipMemcpyDeviceToDevice, c.stream));
    if ((rc = fft_rows(B, Nx, ply, 1))) return rc;
    TIP_LAUNCH("transpose_c", k_transpose_c, dim3(cdiv(Ny, 16), cdiv(Nx, 16)), dim3(256), 0, (const cplx *)B, T1, Nx, Ny);
    if ((rc = fft_rows(T1, Ny, plx, 1))) return rc;
    TIP_HIP(hipMemsetAsync(best, 0, 8, c.stream));
    TIP_HIP(hipMemsetAsync(best + 1, 0xff, 8, c.stream));
    TIP_LAUNCH("absargmax", k_absargmax, dim3(cdiv(n, 256)), dim3(256), 0, (const cplx *)T1, n, best, best + 1);
    TIP_LAUNCH("absargmax2", k_absargmax2, dim3(cdiv(n, 256)), dim3(256), 0, (const cplx *)T1, n, (const unsigned long long *)best,
               best + 1);
    unsigned long long h[4] = {0, 0, 0, 0};
    TIP_HIP(hipMemcpyAsync(h, best, 16, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    const long peak = (long)h[1];
    const int py = (int)(peak / Nx), px = (int)(peak % Nx);
    out4_host[0] = py; out4_host[1] = px; out4_host[2] = 0; out4_host[3] = 0;
    if (upsample <= 1) return TIP_OK;
    // signed whole-pixel shifts as numpy computes them, rounded onto the upsampled grid
    double sy = py, sx = px;
    if (sy > floor(Ny / 2.0)) sy -= Ny;
    if (sx > floor(Nx / 2.0)) sx -= Nx;
    const double uf = (double)upsample;
    sy = nearbyint(sy * uf) / uf;   // np.round: half to even, like nearbyint in the default rounding mode
    sx = nearbyint(sx * uf) / uf;
    const double dftshift = floor(region / 2.0);
    const double offy = dftshift - sy * uf, offx = dftshift - sx * uf;
    cplx *Kx = ws.get<cplx>((size_t)region * Nx), *Ky = ws.get<cplx>((size_t)region * Ny);
    cplx *C1 = ws.get<cplx>((size_t)region * Ny), *O = ws.get<cplx>((size_t)region * region);
    if (!Kx || !Ky || !C1 || !O) return TIP_ERR_NOMEM;
    TIP_LAUNCH("dft_kernel", k_dft_kernel, dim3(cdiv((long)region * Nx, 256)), dim3(256), 0, Kx, region, Nx, offx, uf);
    TIP_LAUNCH("dft_kernel", k_dft_kernel, dim3(cdiv((long)region * Ny, 256)), dim3(256), 0, Ky, region, Ny, offy, uf);
    TIP_LAUNCH("updft1", k_updft1, dim3(cdiv(Ny, 256), region), dim3(256), 0, (const cplx *)Kx, (const cplx *)PT, C1, region, Nx, Ny);
    TIP_LAUNCH("updft2", k_updft2, dim3(region, region), dim3(64), 0, (const cplx *)Ky, (const cplx *)C1, O, region, Ny);
    TIP_HIP(hipMemsetAsync(best, 0, 8, c.stream));
    TIP_HIP(hipMemsetAsync(best + 1, 0xff, 8, c.stream));
    const long nr = (long)region * region;
    TIP_LAUNCH("absargmax", k_absargmax, dim3(cdiv(nr, 256)), dim3(256), 0, (const cplx *)O, nr, best, best + 1);
    TIP_LAUNCH("absargmax2", k_absargmax2, dim3(cdiv(nr, 256)), dim3(256), 0, (const cplx *)O, nr, (const unsigned long long *)best,
               best + 1);
    TIP_HIP(hipMemcpyAsync(h, best, 16, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    out4_host[2] = (int64_t)(h[1] / region);
    out4_host[3] = (int64_t)(h[1] % region);
    return TIP_OK;
}

}  // namespace tip

using namespace tip;

extern "C" {

int tip_phase_correlation_dev(const void *ref, const void *mov, int dtype, int y, int x, int upsample, int64_t *out4_host)
{
    return phase_correlation_dev(ref, mov, dtype, y, x, upsample, out4_host);
}

int tip_phase_correlation(const void *ref, const void *mov, int dtype, int y, int x, int upsample, int64_t *out4)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!ref || !mov || !out4 || y < 1 || x < 1) return fail(TIP_ERR_ARG, "tip_phase_correlation: bad arguments");
    const size_t es = dtype == 0 ? 4 : (dtype == 1 ? 8 : 2);
    const size_t bytes = (size_t)y * x * es;
    WsGuard ws;
    char *da = ws.get<char>(bytes), *db = ws.get<char>(bytes);
    if (!da || !db) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(da, ref, bytes, hipMemcpyHostToDevice, c.stream));
    TIP_HIP(hipMemcpyAsync(db, mov, bytes, hipMemcpyHostToDevice, c.stream));
    return phase_correlation_dev(da, db, dtype, y, x, upsample, out4);
}

}  // extern "C"
