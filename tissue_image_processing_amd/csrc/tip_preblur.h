// tip_preblur.h -- the projection's four short passes in ONE kernel (sp.py:26-37 and the z pass of sp.py:55):
//
//     uint16 stack --(offset, clip)--> z pass sigma 0.5 --> y pass sigma 1 --> x pass sigma 1 --> z pass sigma 0.5 --> float32
//
// i.e. blur_image(ch, (0.5, 1, 1)) followed by the z pass of blur_image(., (0.5, 30, 30)).  As four kernels these move
// 5.2 GB through HBM for 0.75 GB of compulsory traffic (read the uint16 stack, write one float32 volume); here a block
// walks a (PB_Y x PB_X)-pixel tile (+ 4-pixel halo) through all planes:
//   * every thread owns a few pixels of the padded tile and slides their raw values through a 5-plane register window
//     (first z pass, exactly k_zpass_r2_x4's arithmetic), writing the plane to LDS;
//   * y pass and x pass run on that LDS plane (same tap order as k_ypass_slide / k_xpass_slide, each rounded to float32);
//   * the x-pass outputs of the last 5 planes stay in registers and the second z pass streams the result to HBM.
// Every pass accumulates in float64 in scipy's order and rounds to float32 exactly where the separate kernels do, so the
// output is bit-identical to them (tests compare the two paths).  The intermediate volume (the (0.5, 1, 1)-blurred
// channel itself) is not produced: callers that need it (bin_size > 1) use the separate kernels.
#pragma once
#include "tip_slide.h"

namespace tip {

constexpr int PB_Y = 32, PB_X = 128, PB_H = 4, PB_T = 1024, PB_O = 4;   // 1024 threads: 6 padded-tile pixels and 4 outputs each
// (measured: 512 threads with 8 outputs each 0.87 ms, two 512-thread blocks per CU on 16-row tiles 0.58 ms, this 0.57 ms;
//  the four separate kernels 0.78 ms)
struct ShortTaps { double w[8]; };    // the first radius + 1 taps of a Taps (two full Taps would not fit the 4 KB kernel-argument segment)
constexpr int PB_WY = PB_Y + 2 * PB_H, PB_WX = PB_X + 2 * PB_H;       // padded tile: 40 x 136
constexpr int PB_OWN = (PB_WY * PB_WX + PB_T - 1) / PB_T;             // padded-tile pixels per thread (6)
constexpr int PB_YI = PB_WX * (PB_Y / 4);                             // y-pass work items (column, 4-row segment): 1088
static_assert(PB_YI >= PB_T && (PB_YI - PB_T) * 4 <= PB_T, "y-pass work split");

__global__ void __launch_bounds__(PB_T) k_preblur_fused(const uint16_t *__restrict__ src, int airy, const float *__restrict__ clip_p95,
                                                        const int *__restrict__ clip_has, float *__restrict__ out, int Z, int Y, int X,
                                                        ShortTaps k05, ShortTaps k1)
{
    __shared__ __attribute__((aligned(16))) float p1[PB_WY][PB_WX];    // plane after the first z pass (padded tile)
    __shared__ __attribute__((aligned(16))) float yb[PB_Y][PB_WX];     // after the y pass
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * PB_X, y0 = blockIdx.y * PB_Y;
    const long P = (long)Y * X;
    const bool has = *clip_has != 0;
    const float cp = *clip_p95;
    // this thread's pixels of the padded tile (edges replicated: clamped coordinates)
    int off[PB_OWN];     // (a plane has fewer than 2^31 pixels)
#pragma unroll
    for (int k = 0; k < PB_OWN; ++k) {
        const int i = t + k * PB_T;
        const int r = i / PB_WX, c = i - r * PB_WX;
        off[k] = i < PB_WY * PB_WX ? clampi(y0 - PB_H + r, 0, Y - 1) * X + clampi(x0 - PB_H + c, 0, X - 1) : -1;
    }
    auto raw = [&](int z, int k) -> float {
        float v = (float)src[(long)z * P + (long)off[k]];
        if (airy) { v -= 10000.f; if (v < 0.f) v = 0.f; }
        if (has && v > cp) v = cp;
        return v;
    };
    // Both z windows are RINGS indexed by plane % 5 and the plane loop is unrolled five-fold, so that every ring index is a
    // compile-time constant: no window shifting (the shifts were 142 of the loop's ~700 instructions, and the kernel is
    // bound by instruction issue).
    // first z pass: raw values of planes z-2 .. z+2 around the plane being produced, 'nearest' at both ends
    float w[PB_OWN][5];      // (float32: windows in double cost registers and with them occupancy -- measured 1.8x slower)
#pragma unroll
    for (int k = 0; k < PB_OWN; ++k) {
        if (off[k] < 0) continue;
        const float a = raw(0, k);
        w[k][3] = w[k][4] = w[k][0] = a;          // planes -2, -1, 0
        w[k][1] = raw(min(1, Z - 1), k);
        w[k][2] = raw(min(2, Z - 1), k);
    }
    // second z pass: x-pass output planes zo-2 .. zo+2 of this thread's PB_O output pixels (row oy, columns ox ..), same rule
    const int oy = t / (PB_X / PB_O), ox = (t % (PB_X / PB_O)) * PB_O;
    float ring[5][PB_O], last[PB_O];
    for (int zp0 = 0; zp0 < Z + 2; zp0 += 5) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {             // zp % 5 == u
            const int zp = zp0 + u;               // zp < Z: plane zp goes through the in-plane passes; output plane zp - 2 follows
            if (zp >= Z + 2) break;               // (uniform)
            if (zp < Z) {
#pragma unroll
                for (int k = 0; k < PB_OWN; ++k) {
                    if (off[k] < 0) continue;
                    const int i = t + k * PB_T;
                    double tmp = (double)w[k][u] * k05.w[2];
                    tmp += ((double)w[k][(u + 3) % 5] + (double)w[k][(u + 2) % 5]) * k05.w[0];
                    tmp += ((double)w[k][(u + 4) % 5] + (double)w[k][(u + 1) % 5]) * k05.w[1];
                    (&p1[0][0])[i] = (float)tmp;
                    w[k][(u + 3) % 5] = raw(min(zp + 3, Z - 1), k);      // plane zp + 3 takes the slot of plane zp - 2
                }
            }
            __syncthreads();
            if (zp < Z) {
                // y pass: work item = (column, 4-row segment); 136 columns x (PB_Y / 4) segments = 1088 items for 1024
                // threads: the first 1024 go one per thread, the last 64 are split into single outputs over 256 threads
                // (a second whole item for one wave made every other wave wait for it at the barrier)
                {
                    const int yc = t % PB_WX, yr = (t / PB_WX) * 4;
                    double win[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i) win[i] = (double)p1[yr + i][yc];
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        double tmp = win[o + 4] * k1.w[4];
#pragma unroll
                        for (int d = 4; d >= 1; --d) tmp += (win[o + 4 - d] + win[o + 4 + d]) * k1.w[4 - d];
                        yb[yr + o][yc] = (float)tmp;
                    }
                }
                if (t < (PB_YI - PB_T) * 4) {
                    const int item = PB_T + (t >> 2), o = t & 3;
                    const int yc = item % PB_WX, yr = (item / PB_WX) * 4 + o;
                    double win[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) win[i] = (double)p1[yr + i][yc];
                    double tmp = win[4] * k1.w[4];
#pragma unroll
                    for (int d = 4; d >= 1; --d) tmp += (win[4 - d] + win[4 + d]) * k1.w[4 - d];
                    yb[yr][yc] = (float)tmp;
                }
            }
            __syncthreads();
            if (zp < Z) {                                                           // x pass: PB_O outputs from PB_O + 8 inputs
                float v[PB_O + 8];
#pragma unroll
                for (int i = 0; i < (PB_O + 8) / 4; ++i) {
                    const float4 f = *reinterpret_cast<const float4 *>(&yb[oy][ox + 4 * i]);
                    v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
                }
#pragma unroll
                for (int k = 0; k < PB_O; ++k) {
                    double tmp = (double)v[k + 4] * k1.w[4];
#pragma unroll
                    for (int d = 4; d >= 1; --d) tmp += ((double)v[k + 4 - d] + (double)v[k + 4 + d]) * k1.w[4 - d];
                    last[k] = (float)tmp;
                }
            }   // (zp >= Z: `last` keeps the last plane: the window's far end is replicated)
#pragma unroll
            for (int k = 0; k < PB_O; ++k) {
                ring[u][k] = last[k];
                if (zp == 0) ring[3][k] = ring[4][k] = last[k];                 // planes -2, -1
            }
            const int zo = zp - 2;
            if (zo >= 0 && y0 + oy < Y && x0 + ox < X) {
                float o[PB_O];
#pragma unroll
                for (int k = 0; k < PB_O; ++k) {                                // window: planes zo-2 .. zo+2 = slots u+1 .. u+4, u
                    double tmp = (double)ring[(u + 3) % 5][k] * k05.w[2];
                    tmp += ((double)ring[(u + 1) % 5][k] + (double)ring[u][k]) * k05.w[0];
                    tmp += ((double)ring[(u + 2) % 5][k] + (double)ring[(u + 4) % 5][k]) * k05.w[1];
                    o[k] = (float)tmp;
                }
                float *dst = out + (long)zo * P + (long)(y0 + oy) * X + x0 + ox;
                if (x0 + ox + PB_O <= X && (X & 3) == 0) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < PB_O; ++k)
                        if (x0 + ox + k < X) dst[k] = o[k];
                }
            }
        }
    }
}

}  // namespace tip
