// tip_draw.hip -- the overlay images of the reference's Tissue.draw_* methods (ti.py:584-607, 2585-2645) as dense-array kernels:
// the GUI composites these (3, Y, X) float64 images over the frame it shows.
//
//   draw_cell_types            per-pixel colour from the cell-type byte map (bit tests of is_positive_for_type, ti.py:146-176)
//   draw_all_cell_tracking     per-pixel colour from the track id (id mod 6 picks the colour, id 0 stays black)
//   draw_cell_tracking / draw_marking_points / draw_events
//                              filled discs -- skimage.draw.disk(center, radius, shape): the pixels of the disc's bounding box
//                              (ceil(center - r) .. floor(center + r), clipped to the image) with ((r - r0) / R)^2 + ((c - c0) / R)^2 < 1
//                              in float64, coordinates taken relative to the box as skimage does; a later disc paints over an earlier one
//   draw_neighbors_connections straight lines between cell centroids -- skimage.draw.line's integer Bresenham walk
//
// Host arrays in and out (the callers are the GUI's display paths); the arithmetic runs on the device.
#include "tip_internal.h"

namespace tip {

// positive <=> every bit of `must` set and the cell valid (byte != 255) -- unless `must` is empty, then validity is not asked for
// (ti.py:147-155: the tuple form starts from all-ones) -- and no bit of `lack` set on a valid cell
__global__ void __launch_bounds__(256) k_draw_cell_types(const uint8_t *__restrict__ types, long n, int must, int lack, int must_any,
                                                         double p0, double p1, double p2, double n0, double n1, double n2,
                                                         double *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = types[i];
    const bool valid = t != 255;
    bool pos = (t & must) == must && (valid || !must_any);
    if (valid && (t & lack)) pos = false;
    const bool neg = !pos && valid;
    out[i] = pos ? p0 : (neg ? n0 : 0.0);
    out[n + i] = pos ? p1 : (neg ? n1 : 0.0);
    out[2 * n + i] = pos ? p2 : (neg ? n2 : 0.0);
}

struct Cycle { double c[6][3]; };
__global__ void __launch_bounds__(256) k_draw_tracking(const int32_t *__restrict__ track, long n, Cycle cyc, double *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int id = track[i];
    int k = id % 6;                      // (numpy's %: non-negative for a positive divisor)
    if (k < 0) k += 6;
#pragma unroll
    for (int j = 0; j < 3; ++j) out[j * n + i] = id == 0 ? 0.0 : cyc.c[k][j];
}

// one thread per pixel: the LAST disc that covers it decides the colour (the reference paints them in order)
__global__ void __launch_bounds__(256) k_draw_disks(int Y, int X, int nd, const double *__restrict__ cy, const double *__restrict__ cx, double R,
                                                    const double *__restrict__ rgb, double *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    int hit = -1;
    for (int d = 0; d < nd; ++d) {
        const double r0 = cy[d], c0 = cx[d];
        // bounding box as skimage.draw.ellipse builds it, clipped to the image
        int ur = (int)ceil(r0 - R), uc = (int)ceil(c0 - R), lr = (int)floor(r0 + R), lc = (int)floor(c0 + R);
        ur = max(ur, 0); uc = max(uc, 0); lr = min(lr, Y - 1); lc = min(lc, X - 1);
        if (y < ur || y > lr || x < uc || x > lc) continue;
        const double rr = (double)(y - ur) - (r0 - (double)ur), cc = (double)(x - uc) - (c0 - (double)uc);
        const double a = rr / R, b = (-cc) / R;      // (rotation 0: cos = 1, sin = 0)
        if (a * a + b * b < 1.0) hit = d;
    }
    const long P = (long)Y * X, i = (long)y * X + x;
#pragma unroll
    for (int j = 0; j < 3; ++j) out[j * P + i] = hit >= 0 ? rgb[3 * hit + j] : 0.0;
}

// skimage.draw.line (skimage/draw/_draw.pyx `_line`): Bresenham from (r0, c0) to (r1, c1), both ends included
__global__ void __launch_bounds__(256) k_draw_lines(int Y, int X, int nl, const int32_t *__restrict__ ends, unsigned char *__restrict__ img)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nl) return;
    const int r0 = ends[4 * l], c0 = ends[4 * l + 1], r1 = ends[4 * l + 2], c1 = ends[4 * l + 3];
    int r = r0, c = c0, dr = abs(r1 - r0), dc = abs(c1 - c0);
    int sc = (c1 - c) > 0 ? 1 : -1, sr = (r1 - r) > 0 ? 1 : -1;
    bool steep = false;
    if (dr > dc) {
        steep = true;
        int t = c; c = r; r = t;
        t = dc; dc = dr; dr = t;
        t = sc; sc = sr; sr = t;
    }
    int d = 2 * dr - dc;
    auto put = [&](int rr, int cc) {
        if (rr >= 0 && rr < Y && cc >= 0 && cc < X) img[(long)rr * X + cc] = 1;
    };
    for (int i = 0; i < dc; ++i) {
        if (steep) put(c, r); else put(r, c);
        while (d >= 0) { r += sr; d -= 2 * dc; }
        c += sc;
        d += 2 * dr;
    }
    put(r1, c1);
}

__global__ void __launch_bounds__(256) k_draw_expand(const unsigned char *__restrict__ img, long n, double c0, double c1, double c2,
                                                     double *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = img[i] ? 1.0 : 0.0;
    out[i] = v * c0; out[n + i] = v * c1; out[2 * n + i] = v * c2;
}

}  // namespace tip

using namespace tip;

extern "C" {

int tip_draw_cell_types_u8(const uint8_t *types, long n, int must_mask, int lack_mask, const double *pos_rgb, const double *neg_rgb, double *out3)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!types || !pos_rgb || !neg_rgb || !out3 || n < 1 || must_mask < 0 || must_mask > 255 || lack_mask < 0 || lack_mask > 255)
        return fail(TIP_ERR_ARG, "tip_draw_cell_types_u8: bad arguments");
    WsGuard ws;
    uint8_t *dt = ws.get<uint8_t>((size_t)n);
    double *dout = ws.get<double>((size_t)3 * n);
    if (!dt || !dout) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(dt, types, (size_t)n, hipMemcpyHostToDevice, c.stream));
    TIP_LAUNCH("draw_cell_types", k_draw_cell_types, dim3(cdiv(n, 256)), dim3(256), 0, (const uint8_t *)dt, n, must_mask, lack_mask,
               must_mask != 0 ? 1 : 0, pos_rgb[0], pos_rgb[1], pos_rgb[2], neg_rgb[0], neg_rgb[1], neg_rgb[2], dout);
    TIP_HIP(hipMemcpyAsync(out3, dout, (size_t)3 * n * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_draw_tracking_i32(const int32_t *track, long n, const double *cycle18, double *out3)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!track || !cycle18 || !out3 || n < 1) return fail(TIP_ERR_ARG, "tip_draw_tracking_i32: bad arguments");
    WsGuard ws;
    int32_t *dt = ws.get<int32_t>((size_t)n);
    double *dout = ws.get<double>((size_t)3 * n);
    if (!dt || !dout) return TIP_ERR_NOMEM;
    Cycle cyc;
    for (int k = 0; k < 6; ++k)
        for (int j = 0; j < 3; ++j) cyc.c[k][j] = cycle18[3 * k + j];
    TIP_HIP(hipMemcpyAsync(dt, track, (size_t)n * 4, hipMemcpyHostToDevice, c.stream));
    TIP_LAUNCH("draw_tracking", k_draw_tracking, dim3(cdiv(n, 256)), dim3(256), 0, (const int32_t *)dt, n, cyc, dout);
    TIP_HIP(hipMemcpyAsync(out3, dout, (size_t)3 * n * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_draw_disks_f64(int y, int x, int n, const double *cy, const double *cx, double radius, const double *rgb, double *out3)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!out3 || y < 1 || x < 1 || n < 0 || (n > 0 && (!cy || !cx || !rgb)) || !(radius > 0.0))
        return fail(TIP_ERR_ARG, "tip_draw_disks_f64: bad arguments");
    const long P = (long)y * x;
    WsGuard ws;
    double *dcy = ws.get<double>((size_t)n + 1), *dcx = ws.get<double>((size_t)n + 1), *drgb = ws.get<double>((size_t)3 * n + 1);
    double *dout = ws.get<double>((size_t)3 * P);
    if (!dcy || !dcx || !drgb || !dout) return TIP_ERR_NOMEM;
    if (n > 0) {
        TIP_HIP(hipMemcpyAsync(dcy, cy, (size_t)n * 8, hipMemcpyHostToDevice, c.stream));
        TIP_HIP(hipMemcpyAsync(dcx, cx, (size_t)n * 8, hipMemcpyHostToDevice, c.stream));
        TIP_HIP(hipMemcpyAsync(drgb, rgb, (size_t)3 * n * 8, hipMemcpyHostToDevice, c.stream));
    }
    TIP_LAUNCH("draw_disks", k_draw_disks, dim3(cdiv(x, 256), y), dim3(256), 0, y, x, n, (const double *)dcy, (const double *)dcx, radius,
               (const double *)drgb, dout);
    TIP_HIP(hipMemcpyAsync(out3, dout, (size_t)3 * P * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_draw_lines_f64(int y, int x, int n, const int32_t *ends, const double *rgb, double *out3)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!out3 || !rgb || y < 1 || x < 1 || n < 0 || (n > 0 && !ends)) return fail(TIP_ERR_ARG, "tip_draw_lines_f64: bad arguments");
    const long P = (long)y * x;
    WsGuard ws;
    int32_t *de = ws.get<int32_t>((size_t)4 * n + 4);
    unsigned char *img = ws.get<unsigned char>((size_t)P);
    double *dout = ws.get<double>((size_t)3 * P);
    if (!de || !img || !dout) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemsetAsync(img, 0, (size_t)P, c.stream));
    if (n > 0) {
        TIP_HIP(hipMemcpyAsync(de, ends, (size_t)4 * n * 4, hipMemcpyHostToDevice, c.stream));
        TIP_LAUNCH("draw_lines", k_draw_lines, dim3(cdiv(n, 256)), dim3(256), 0, y, x, n, (const int32_t *)de, img);
    }
    TIP_LAUNCH("draw_expand", k_draw_expand, dim3(cdiv(P, 256)), dim3(256), 0, (const unsigned char *)img, P, rgb[0], rgb[1], rgb[2], dout);
    TIP_HIP(hipMemcpyAsync(out3, dout, (size_t)3 * P * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

}  // extern "C"
