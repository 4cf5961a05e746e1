// tip_gauss.hip -- separable Gaussian entry points (scipy.ndimage.gaussian_filter(mode='nearest'),
// reference call site bim.py:389 via sp.py:37,55,70,71 and ti.py:142).
#include "tip_corr.h"
#include "tip_slide.h"

namespace tip {

constexpr int LONG_TO = 256;  // outputs per line per block in the long-kernel variant
constexpr int LONG_R = 8;     // outputs per register window

template <typename T>
static int corr_generic(const T *in, T *out, int Z, int Y, int X, int axis, const Taps &t)
{
    LoadPlain<T> ld{in, (long)Y * X, (long)X};
    dim3 grid(cdiv(X, 256), Y, Z), block(256);
    const bool f64 = sizeof(T) == 8;
    if (axis == 0) TIP_LAUNCH(f64 ? "corr_generic_z_f64" : "corr_generic_z", (k_corr_generic<T, 0, LoadPlain<T>>), grid, block, 0, ld, out, Z, Y, X, t);
    else if (axis == 1) TIP_LAUNCH(f64 ? "corr_generic_y_f64" : "corr_generic_y", (k_corr_generic<T, 1, LoadPlain<T>>), grid, block, 0, ld, out, Z, Y, X, t);
    else TIP_LAUNCH(f64 ? "corr_generic_x_f64" : "corr_generic_x", (k_corr_generic<T, 2, LoadPlain<T>>), grid, block, 0, ld, out, Z, Y, X, t);
    return TIP_OK;
}

int corr_long_f32(const float *in, float *out, int Z, int Y, int X, int axis, const Taps &t)
{
    const int r = t.n >> 1;
    if (axis == 1) {
        size_t lds = (size_t)(LONG_TO + 2 * r) * 64 * sizeof(float);
        auto k = k_corr_long_f32<1, LONG_TO, LONG_R>;
        TIP_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        dim3 grid(cdiv(X, 64), cdiv(Y, LONG_TO), Z), block(256);
        TIP_LAUNCH("corr_long_y", k, grid, block, lds, in, out, Z, Y, X, t);
    } else {
        size_t lds = (size_t)(LONG_TO + 2 * r) * 65 * sizeof(float);
        auto k = k_corr_long_f32<2, LONG_TO, LONG_R>;
        TIP_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        dim3 grid(cdiv(Y, 64), cdiv(X, LONG_TO), Z), block(256);
        TIP_LAUNCH("corr_long_x", k, grid, block, lds, in, out, Z, Y, X, t);
    }
    return TIP_OK;
}

// dtype 0 = f32, 1 = f64.  force: 0 auto, 1 generic, 2 long
int correlate1d_dev(const void *in, void *out, int dtype, int Z, int Y, int X, int axis, const Taps &t, int force)
{
    if (Z <= 0 || Y <= 0 || X <= 0) return fail(TIP_ERR_ARG, "correlate1d: empty volume");
    if (Y > 65535 || Z > 65535) return fail(TIP_ERR_ARG, "correlate1d: y and z must be <= 65535");
    if (axis < 0 || axis > 2) return fail(TIP_ERR_ARG, "correlate1d: axis %d", axis);
    if (in == out) return fail(TIP_ERR_ARG, "correlate1d: in-place is not supported");
    const int r = t.n >> 1;
    bool use_long = dtype == 0 && axis != 0 && r >= 12 && r <= 120;
    if (force == 1) use_long = false;
    if (force == 2) {
        if (dtype != 0 || axis == 0 || r > 120) return fail(TIP_ERR_ARG, "long kernel needs f32, axis 1|2, r<=120");
        use_long = true;
    }
    if (use_long) return corr_long_f32((const float *)in, (float *)out, Z, Y, X, axis, t);
    if (dtype == 0) return corr_generic<float>((const float *)in, (float *)out, Z, Y, X, axis, t);
    if (dtype == 1) {
        // radius 12 (sigma 3, the watershed's default blur, bim.py:474) along y / x: register-sliding kernels, each
        // input loaded once per thread instead of 25 times through the caches
        if (r == 12 && axis == 1) {
            TIP_LAUNCH("yslide_r12_f64", (k_ypass_slide<double, 12, 2>), dim3(cdiv(X, 256), cdiv(Y, 50), Z), dim3(256), 0,
                       (const double *)in, (double *)out, Y, X, t);
            return TIP_OK;
        }
        if (r == 12 && axis == 2) {
            TIP_LAUNCH("xslide_r12_f64", (k_xpass_slide<double, 12>), dim3(cdiv(cdiv(X, 8), 256), Y, Z), dim3(256), 0,
                       (const double *)in, (double *)out, Y, X, t);
            return TIP_OK;
        }
        return corr_generic<double>((const double *)in, (double *)out, Z, Y, X, axis, t);
    }
    return fail(TIP_ERR_ARG, "correlate1d: dtype %d", dtype);
}

// taps in device memory (any odd, symmetric count): the generic kernel's arithmetic, for radii beyond the tap table
template <typename T, int AXIS>
__global__ void __launch_bounds__(256) k_corr_big(const T *__restrict__ in, T *__restrict__ out, int Z, int Y, int X,
                                                  const double *__restrict__ w, int n)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= X) return;
    const int r = n >> 1;
    const int len = AXIS == 0 ? Z : (AXIS == 1 ? Y : X);
    const int c = AXIS == 0 ? z : (AXIS == 1 ? y : x);
    const long stride = AXIS == 0 ? (long)Y * X : (AXIS == 1 ? (long)X : 1L);
    const T *line = in + ((long)z * Y + y) * X + x - (long)c * stride;      // element 0 of this output's line
    auto at = [&](int i) -> double { return (double)line[(long)clampi(i, 0, len - 1) * stride]; };
    double tmp = at(c) * w[r];
    for (int d = r; d >= 1; --d) tmp += (at(c - d) + at(c + d)) * w[r - d];
    out[((long)z * Y + y) * X + x] = (T)tmp;
}

// in -> out through up to three axis passes; `out` doubles as scratch together with one workspace.
int gaussian3d_dev(const void *in, void *out, int dtype, int Z, int Y, int X, const double *tz, int nz,
                   const double *ty, int ny, const double *tx, int nx)
{
    const size_t es = dtype == 0 ? 4 : 8;
    const size_t bytes = (size_t)Z * Y * X * es;
    const double *tp[3] = {tz, ty, tx};
    const int np_[3] = {nz, ny, nx};
    int passes = 0;
    for (int a = 0; a < 3; a++) passes += np_[a] > 0;
    if (passes == 0) {
        TIP_HIP(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, ctx().stream));
        return TIP_OK;
    }
    WsGuard ws;
    void *tmp = nullptr;
    if (passes > 1) {
        tmp = ws.get<char>(bytes);
        if (!tmp) return TIP_ERR_NOMEM;
    }
    // ping-pong so that the last pass lands in `out`
    const void *cur = in;
    int done = 0;
    for (int a = 0; a < 3; a++) {
        if (np_[a] <= 0) continue;
        const int remaining = passes - done - 1;
        void *dst = (remaining % 2 == 0) ? out : tmp;
        int rc;
        if (np_[a] > 255) {
            // radius > 127 (sigma > 31.8): more taps than the kernel-argument tap table holds; they go to device memory and a
            // plain one-thread-per-output kernel walks them in scipy's order (no reference call site is this wide: slow path)
            const int n = np_[a];
            if (n % 2 == 0 || n > 8191) return fail(TIP_ERR_ARG, "gaussian: %d taps (odd, at most 8191)", n);
            for (int i = 0; i < n / 2; ++i)
                if (tp[a][i] != tp[a][n - 1 - i]) return fail(TIP_ERR_ARG, "gaussian: taps must be symmetric");
            double *wd = ws.get<double>((size_t)n);
            if (!wd) return TIP_ERR_NOMEM;
            TIP_HIP(hipMemcpyAsync(wd, tp[a], (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx().stream));
            TIP_HIP(hipStreamSynchronize(ctx().stream));      // (the host tap array may be a temporary of the caller)
            const dim3 grid(cdiv(X, 256), Y, Z);
            if (dtype == 0) {
                if (a == 0) { TIP_LAUNCH("corr_big", (k_corr_big<float, 0>), grid, dim3(256), 0, (const float *)cur, (float *)dst, Z, Y, X, (const double *)wd, n); }
                else if (a == 1) { TIP_LAUNCH("corr_big", (k_corr_big<float, 1>), grid, dim3(256), 0, (const float *)cur, (float *)dst, Z, Y, X, (const double *)wd, n); }
                else { TIP_LAUNCH("corr_big", (k_corr_big<float, 2>), grid, dim3(256), 0, (const float *)cur, (float *)dst, Z, Y, X, (const double *)wd, n); }
            } else {
                if (a == 0) { TIP_LAUNCH("corr_big", (k_corr_big<double, 0>), grid, dim3(256), 0, (const double *)cur, (double *)dst, Z, Y, X, (const double *)wd, n); }
                else if (a == 1) { TIP_LAUNCH("corr_big", (k_corr_big<double, 1>), grid, dim3(256), 0, (const double *)cur, (double *)dst, Z, Y, X, (const double *)wd, n); }
                else { TIP_LAUNCH("corr_big", (k_corr_big<double, 2>), grid, dim3(256), 0, (const double *)cur, (double *)dst, Z, Y, X, (const double *)wd, n); }
            }
        } else {
            Taps t;
            rc = make_taps(t, tp[a], np_[a]);
            if (rc) return rc;
            rc = correlate1d_dev(cur, dst, dtype, Z, Y, X, a, t, 0);
            if (rc) return rc;
        }
        cur = dst;
        done++;
    }
    return TIP_OK;
}

}  // namespace tip

using namespace tip;

extern "C" {

int tip_correlate1d_dev(const void *in, void *out, int dtype, int z, int y, int x, int axis, const double *taps_host,
                        int n)
{
    if (!in || !out || !taps_host) return fail(TIP_ERR_ARG, "tip_correlate1d_dev: null pointer");
    int force = 0;
    if (axis >= 100) { force = axis / 100; axis %= 100; }  // test hook: 1xx generic, 2xx long kernel
    Taps t;
    int rc = make_taps(t, taps_host, n);
    if (rc) return rc;
    return correlate1d_dev(in, out, dtype, z, y, x, axis, t, force);
}

int tip_gaussian3d_dev_w(const void *in, void *out, int dtype, int z, int y, int x, const double *tz, int nz,
                         const double *ty, int ny, const double *tx, int nx)
{
    if (!in || !out) return fail(TIP_ERR_ARG, "tip_gaussian3d_dev_w: null pointer");
    if (dtype != 0 && dtype != 1) return fail(TIP_ERR_ARG, "dtype must be 0 (f32) or 1 (f64)");
    return gaussian3d_dev(in, out, dtype, z, y, x, tz, nz, ty, ny, tx, nx);
}

int tip_gaussian3d_w(const void *in, void *out, int dtype, int z, int y, int x, const double *tz, int nz,
                     const double *ty, int ny, const double *tx, int nx)
{
    if (!in || !out) return fail(TIP_ERR_ARG, "tip_gaussian3d_w: null pointer");
    if (dtype != 0 && dtype != 1) return fail(TIP_ERR_ARG, "dtype must be 0 (f32) or 1 (f64)");
    if (z <= 0 || y <= 0 || x <= 0) return fail(TIP_ERR_ARG, "empty volume");
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    const size_t bytes = (size_t)z * y * x * (dtype == 0 ? 4 : 8);
    WsGuard ws;
    char *din = ws.get<char>(bytes), *dout = ws.get<char>(bytes);
    if (!din || !dout) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, c.stream));
    int rc = gaussian3d_dev(din, dout, dtype, z, y, x, tz, nz, ty, ny, tx, nx);
    if (rc) return rc;
    TIP_HIP(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

static int taps_for(double sigma, double truncate, double *buf, int *n)
{
    *n = 0;
    if (sigma > 1e-15) {
        int k = libm_taps(sigma, truncate, buf, 255);
        if (k < 0) return fail(TIP_ERR_ARG, "sigma %g too large (radius > 127)", sigma);
        *n = k;
    }
    return TIP_OK;
}

int tip_gaussian3d_f32(const float *in, float *out, int z, int y, int x, double sz, double sy, double sx,
                       double truncate)
{
    double tz[256], ty[256], tx[256];
    int nz, ny, nx, rc;
    if ((rc = taps_for(sz, truncate, tz, &nz)) || (rc = taps_for(sy, truncate, ty, &ny)) ||
        (rc = taps_for(sx, truncate, tx, &nx)))
        return rc;
    return tip_gaussian3d_w(in, out, 0, z, y, x, tz, nz, ty, ny, tx, nx);
}

int tip_gaussian2d_f64(const double *in, double *out, int y, int x, double sy, double sx, double truncate)
{
    double ty[256], tx[256];
    int ny, nx, rc;
    if ((rc = taps_for(sy, truncate, ty, &ny)) || (rc = taps_for(sx, truncate, tx, &nx))) return rc;
    return tip_gaussian3d_w(in, out, 1, 1, y, x, nullptr, 0, ty, ny, tx, nx);
}

}  // extern "C"
