// tip_watershed.hip -- skimage.segmentation.watershed(image, markers=None, connectivity=1, watershed_line=True)
// (reference call sites bim.py:475 and pl.py:194) as a data-parallel flood.
//
// The serial algorithm (skimage/segmentation/_watershed_cy.pyx, restated for the tests under oracle/) pops pixels from
// a (value, age) heap.  What decides a pixel's fate is only WHICH OF ITS NEIGHBOURS WERE LABELLED BEFORE IT POPS:
//   * a pixel pops at time T = (value, index) -- or, if every lower neighbour is a watershed line, right after the
//     first neighbour that gets labelled later ("pulled", it then inherits that neighbour's pop time and label);
//   * when it pops, it becomes a line if the neighbours labelled before it carry >= 2 different labels, else it
//     takes their label.
// Mode A ("rounds") evaluates exactly that rule for every undecided pixel in parallel and only commits a pixel when
// the states it read certify the outcome (a not-yet-decided neighbour that could still pop earlier makes the pixel
// wait; a bounded flood of the "pocket" of earlier-keyed undecided pixels proves that nothing can reach it first).
// Decisions are monotone, so stale reads are merely conservative, and tiles iterate to a local fixed point in LDS.
// If the tile rounds, the per-component endgame and the wide pass all stall (plateaus larger than any certificate), the
// rest is one serial dependency chain and is finished by the host stage flood_keyed_finish (tip_ws_serial.hip) with the
// same pop-time rule, which keeps the result identical to the serial flood for any image whose non-marker pixels carry
// DISTINCT values -- that is the guarantee of mode A.  Mode A orders equal values by raster index, the serial heap by push
// age.  Ties are looked for where they can matter locally -- between non-marker pixels that are 4-neighbours or share a
// 4-neighbour (diagonals, distance two: a pulled pixel between them sees one before the other) -- `flags` bit0 reports them,
// and unless the caller chose the fast policy (TIP_WS_TIES=fast) such an image is flooded by the exact serial replay
// flood_exact instead (bit2).  Equal values between pixels further apart can still matter through a CHAIN of pulled pixels
// (a pocket enclosed by lines floods at once); float landscapes do not produce such exact ties away from plateaus, integer
// landscapes trip the local detector everywhere, and no such case is known -- but it is outside the guarantee.
// Mode B handles two-valued images (pl.py:194 floods a {0,255} boundary image) EXACTLY: the pop order of the equal-keyed
// markers follows from the array heap's mechanics (tip_heaporder.hip), everything after it is a FIFO, i.e. a
// breadth-first search in generations whose pixels carry dense ranks (see the mode B section below).
#include "tip_internal.h"
#include "tip_uf.h"
#include <algorithm>
#include <cstdlib>

namespace tip {

int correlate1d_dev(const void *in, void *out, int dtype, int Z, int Y, int X, int axis, const Taps &t, int force);
int marker_pop_order(const uint8_t *c, long M, uint32_t *order);   // tip_heaporder.hip
int flood_exact(const double *img, const int32_t *markers, int32_t *labels, int Y, int X);   // tip_ws_serial.hip
long flood_keyed_finish(const double *img, uint64_t *st, int Y, int X);

// ---- helpers ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long enc_f64(double d)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double dec_f64(unsigned long long e)
{
    unsigned long long b = (e >> 63) ? (e & 0x7fffffffffffffffULL) : ~e;
    return __longlong_as_double((long long)b);
}

__device__ __forceinline__ unsigned long long pack_st(int lab, int tref)
{
    return ((unsigned long long)(unsigned)tref << 32) | (unsigned)lab;
}
__device__ __forceinline__ int st_lab(unsigned long long s) { return (int)(unsigned)(s & 0xffffffffULL); }
__device__ __forceinline__ int st_tref(unsigned long long s) { return (int)(unsigned)(s >> 32); }

struct WsInfo {           // device-resident scalars
    unsigned long long emin, emax;   // encoded min / max of the image
    unsigned long long n_other;      // pixels that are neither min nor max
    int ties;                        // equal-valued non-marker neighbours exist
    int n_markers;
    int changed, undecided;          // per-iteration counters (mode A) / frontier, pending (mode B)
    int unfinished, pad_;            // endgame: components whose replay hit the step limit
    int changed_part[64];            // tile / component kernels spread their `changed` adds over 64 words: thousands of
                                     // same-address atomics per launch serialise in L2 (host adds them up)
    unsigned long long dbg_rounds, dbg_tiles, dbg_evals;  // diagnostics (TIP_WS_DEBUG=1)
    unsigned long long dbg_idle, dbg_certs;               // tile instances that decided nothing / that ran a certificate round
    // endgame results (own words: the tile launches that follow the endgame in the same submission must not clobber them)
    int end_part[64];                // serial commits, spread like changed_part
    int end_oversize, end_unfinished;   // cells of components larger than END_CAP / components whose replay hit the step limit
    int ncomp, ncells;               // endgame: components of undecided pixels and their cells (k_end_offsets)
    int und_total, front_total;      // k_ws_tile_totals: undecided pixels / those of them that touch a labelled pixel
};

// Both reductions: 4 independent loads per thread and trip, one atomic per BLOCK (thousands of same-address 64-bit
// atomics serialise in L2 and used to cost more than the 32 MB read itself).
constexpr int WS_RED_BLOCKS = 512;

__global__ void __launch_bounds__(256) k_ws_minmax(const double *__restrict__ v, long n, WsInfo *info)
{
    __shared__ unsigned long long slo[4], shi[4];
    unsigned long long lo = ~0ULL, hi = 0ULL;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 4 * stride) {
        unsigned long long e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = i + u * stride < n ? enc_f64(v[i + u * stride]) : enc_f64(v[i]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lo = e[u] < lo ? e[u] : lo;
            hi = e[u] > hi ? e[u] : hi;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long l2 = __shfl_xor(lo, d, 64), h2 = __shfl_xor(hi, d, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            lo = slo[w] < lo ? slo[w] : lo;
            hi = shi[w] > hi ? shi[w] : hi;
        }
        atomicMin(&info->emin, lo);
        atomicMax(&info->emax, hi);
    }
}

__global__ void __launch_bounds__(256) k_ws_count_other(const double *__restrict__ v, long n, WsInfo *info)
{
    __shared__ unsigned long long sc[4];
    const unsigned long long emin = info->emin, emax = info->emax;
    unsigned long long c = 0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 4 * stride) {
        unsigned long long e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = i + u * stride < n ? enc_f64(v[i + u * stride]) : emin;
#pragma unroll
        for (int u = 0; u < 4; ++u) c += (e[u] != emin && e[u] != emax);
    }
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        c = sc[0] + sc[1] + sc[2] + sc[3];
        if (c) atomicAdd(&info->n_other, c);
    }
}

// ---- markers: label(local_minima(image)) ---------------------------------------------------------------------------
// Equal-valued neighbours belong to one plateau.  On a two-valued image (the U-Net tail's boundary map) the plateau of the
// MAXIMUM is one giant network that can never be a minimum (it touches the other value somewhere): its pixels stay out of the
// union-find -- hundreds of thousands of unions onto one root were 0.65 ms of contention -- and k_ws_lower_flags marks every one
// of them as "has a lower neighbour".  Whether the image is two-valued is read from the device-side scalars of the two
// reductions that ran just before (no host round trip).
struct SameF64 {
    const double *v;
    const WsInfo *info;
    __device__ __forceinline__ bool two_valued_max(int i) const
    {
        return info->n_other == 0 && info->emin != info->emax && enc_f64(v[i]) == info->emax;
    }
    __device__ __forceinline__ bool valid(int i) const { return !two_valued_max(i); }
    __device__ __forceinline__ bool same(int i, int j) const { return v[i] == v[j]; }
};

// flag[root] |= 1 when the plateau has a strictly lower 4-neighbour, or equals the global maximum and touches the
// image border (skimage pads with the value no candidate can beat and rejects plateaus that equal it, extrema.py)
__global__ void __launch_bounds__(256) k_ws_lower_flags(const double *__restrict__ v, const int *__restrict__ parent,
                                                        int *__restrict__ flag, int Y, int X, const WsInfo *info)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    const int i = y * X + x;
    const double h = v[i];
    bool bad = false;
    if (y > 0 && v[i - X] < h) bad = true;
    if (x > 0 && v[i - 1] < h) bad = true;
    if (x < X - 1 && v[i + 1] < h) bad = true;
    if (y < Y - 1 && v[i + X] < h) bad = true;
    if ((y == 0 || x == 0 || y == Y - 1 || x == X - 1) && enc_f64(h) == info->emax) bad = true;
    if (info->n_other == 0 && info->emin != info->emax && enc_f64(h) == info->emax) bad = true;   // (see SameF64: left out of the union-find)
    // (one giant plateau -- the boundary network of a two-valued image -- would otherwise take hundreds of thousands of
    // same-address atomics; the flag only ever goes 0 -> 1, so a stale 0 just costs one more atomic)
    if (bad) { const int r = parent[i]; if (flag[r] == 0) atomicOr(&flag[r], 1); }
}

__global__ void __launch_bounds__(256) k_ws_min_roots(const int *__restrict__ parent, const int *__restrict__ flag,
                                                      int *__restrict__ isroot, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) isroot[i] = (parent[i] == (int)i && flag[i] == 0) ? 1 : 0;
}

// st[i] = (marker label, tref = i) or (0, 0); also detects value ties between non-marker neighbours
__global__ void __launch_bounds__(256) k_ws_init_state(const double *__restrict__ v, const int *__restrict__ parent,
                                                       const int *__restrict__ flag, const int *__restrict__ rank,
                                                       unsigned long long *__restrict__ st, int Y, int X, WsInfo *info)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    const int i = y * X + x;
    const int r = parent[i];
    const bool is_min = flag[r] == 0;
    st[i] = is_min ? pack_st(rank[r] + 1, i) : 0ULL;
    if (!is_min) {
        const double h = v[i];
        bool tie = false;
        if (x > 0 && v[i - 1] == h) tie = true;
        if (y > 0 && v[i - X] == h) tie = true;
        // ... and between non-marker pixels that SHARE a neighbour (diagonals, distance two): a lower pixel between them that is
        // enclosed by lines pops right after whichever of the two pops first ("pulled") and is then seen, or not, by the other
        auto nm_eq = [&](int j) { return v[j] == h && flag[parent[j]] != 0; };
        if (y > 0 && x > 0 && nm_eq(i - X - 1)) tie = true;
        if (y > 0 && x + 1 < X && nm_eq(i - X + 1)) tie = true;
        if (y > 1 && nm_eq(i - 2 * X)) tie = true;
        if (x > 1 && nm_eq(i - 2)) tie = true;
        if (tie) info->ties = 1;
    }
}

// ---- mode A: tile-local rounds ------------------------------------------------------------------------------------------
// tile interior / threads per block are template parameters: 16x16 tiles with one wave each measured best (4.5 ms per
// 2048^2 frame; 32x32 / 256 threads 5.0 ms; 64x64 / 1024 threads 5.9 ms): the rounds are latency bound, so what counts
// is how many tiles a CU keeps in flight (LDS per tile)
// Two launch flavours: the everyday one certifies pockets of up to 6 cells inside a 3-pixel halo; when a whole
// launch makes no progress the wide one (12-pixel halo, 48-cell pockets: stuck pockets are thin staircases up to
// ~10 px long on smooth landscapes) is tried before the serial finish (tip_ws_serial.hip).
constexpr int WT_FAST = 16, WTH_FAST = 64, WH_FAST = 3, WK_FAST = 6;
constexpr int WT_WIDE = 32, WTH_WIDE = 256, WH_WIDE = 12, WK_WIDE = 48;
// Everyday tile flavour (index into tile_launch's switch) and the opening (tile launches before / after the early endgame).
// Measured on the 2048^2 headline frame, tile kernel time per frame / launches / single-frame rate:
//   0  16x16, halo 3, interior only, every waiting cell re-evaluated each round   1.65 ms / 20 / 172.7 frames/s  (opening 10,8)
//   5  same with the event-driven work list                                       1.45 ms / 18 / 179.4          (8,6)
//   6  16x16, halo 4, 3-cell evaluated margin, event-driven                        1.49 ms / 14 / 181.6          (6,6)
//  14  16x16, halo 3, 2-cell evaluated margin, event-driven (10 KB: 16 tiles / CU)  1.40 ms / 16 / 185.8          (8,6)
//   4  as 6 without the event-driven list 1.58 ms (6,6); margins 5 / 7: 1.83 / 2.67 ms; 32x32 tiles (64, 128, 256 threads):
//   2.2 - 2.5 ms -- a launch costs in proportion to the cells it evaluates, and fewer resident tiles hide less latency;
//   8x8 tiles (12 / 13): 1.9 - 2.1 ms (more launches, more halo per interior cell).  Resident tiles per CU matter: padding the
//   block's LDS so that 10 / 8 / 6 tiles fit instead of 13 (TIP_WS_LDS_PAD) gives 1.70 / 1.83 / 2.19 ms.
constexpr int WS_TILE_DEFAULT = 14, WS_OPEN_A = 8, WS_OPEN_B = 6;
constexpr int LINE_LAB = -1;
constexpr int WS_LDS_PAD = 0;
constexpr int WS_CERT_FROM = 0;         // first tile launch (index within the frame) that may use pocket certificates
constexpr int WST_STUCK = 0x40000000;   // tile_wst: the tile's last run decided nothing (low bits: undecided cells left in its window)
// tile-local marker "undecided and already on the work list": label 0 with a non-zero reference field (never leaves LDS)
constexpr unsigned long long ST_LISTED = 1ULL << 32;

struct T2 { double v; int i; };
__device__ __forceinline__ bool t_lt(const T2 &a, const T2 &b) { return a.v < b.v || (a.v == b.v && a.i < b.i); }

// one window cell: value slot and packed state side by side, so the flood rule fetches a neighbour with ONE 16-byte LDS read
struct __attribute__((aligned(16))) WCell { double v; unsigned long long st; };

struct TileView {
    const WCell *cell;                 // LDS window.  .st: packed state (label | pop-time reference pixel << 32);
                                       // .v: undecided cell: its image value (= key value); labelled cell: its pop-time
                                       // VALUE (its own value, or the puller's pop-time value for a pulled pixel)
    unsigned short *vis;               // LDS: this thread's pocket list
    int budget;                        // pocket flood budget (cells)
    int WL;                            // window edge (tile + 2 * halo)
    int g00, X;                        // global linear index of window cell 0 (may be negative), image row length
    // global linear index of window cell c (meaningless for cells outside the image: those are LINE and never compared)
    __device__ __forceinline__ int gi(int c) const { const int cy = c / WL; return g00 + cy * X + (c - cy * WL); }
    __device__ __forceinline__ T2 key(int c) const { return T2{cell[c].v, gi(c)}; }
};

// Is undecided cell q (key < t) certain not to be labelled before time t?  Flood the pocket of undecided cells with
// key < t around q (breadth first, the per-thread list in LDS is queue and visited set at once); the pocket is closed
// iff nothing labelled before t touches it.  Running out of budget or window is "cannot certify" (the pixel waits).
// Out of line to keep the everyday rule small -- so everything it needs travels BY VALUE in registers, with the LDS
// arrays as address-space-3 pointers: a TileView reference would live on the (global-memory) stack and every field
// access in the flood would be a scratch load (measured: a certificate round cost 400k cycles that way).
typedef __attribute__((address_space(3))) const WCell *lds_ccell;
typedef __attribute__((address_space(3))) unsigned short *lds_u16;

__device__ __forceinline__ bool ws_cert(lds_ccell cell, lds_u16 vis, int budget, int WL, int g00, int X, int q, int asker,
                                     double tvv, int tii)
{
    const T2 t{tvv, tii};
    int nv = 1, head = 0;
    vis[0] = (unsigned short)q;
    while (head < nv) {
        const int c = vis[head++];
        const int cy = c / WL, cx = c - cy * WL;
        if (cy == 0 || cy == WL - 1 || cx == 0 || cx == WL - 1) return false;  // neighbours outside the window
        const int gc = g00 + cy * X + cx;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const int m = k == 0 ? c - WL : (k == 1 ? c - 1 : (k == 2 ? c + 1 : c + WL));
            if (m == asker) continue;
            const int gm = k == 0 ? gc - X : (k == 1 ? gc - 1 : (k == 2 ? gc + 1 : gc + X));
            const double cmv = cell[m].v;
            const unsigned long long sm = cell[m].st;
            const int l = st_lab(sm);
            if (l == LINE_LAB) continue;
            if (l > 0) {
                if (t_lt(T2{cmv, st_tref(sm)}, t)) return false;
            } else if (t_lt(T2{cmv, gm}, t)) {
                bool seen = false;
                for (int j = 0; j < nv; ++j) seen |= vis[j] == (unsigned short)m;
                if (!seen) {
                    if (nv >= budget) return false;
                    vis[nv++] = (unsigned short)m;
                }
            }
        }
    }
    return true;
}

__device__ __forceinline__ bool ws_cert(const TileView &tv, int q, int asker, double tvv, int tii)
{
    return ws_cert((lds_ccell)tv.cell, (lds_u16)tv.vis, tv.budget, tv.WL, tv.g00, tv.X, q, asker, tvv, tii);
}

struct Decision { int lab; int ti; double tv; };  // lab == 0: no decision; (tv, ti) = pop time: value and reference pixel

// The flood rule for one undecided cell, written for few instructions: all LDS loads first, then predicated
// arithmetic; the pocket certificates (rare) are the only calls.  certs == false: any undecided neighbour that could
// pop earlier makes the pixel wait (the common case: that neighbour is simply not processed yet).
__device__ __forceinline__ Decision ws_decide(const TileView &tv, int c, int gc, bool certs)
{
    Decision d{0, 0, 0.0};
    const int WL = tv.WL;
    const int q0 = c - WL, q1 = c - 1, q2 = c + 1, q3 = c + WL;
    const WCell n0 = tv.cell[q0], n1 = tv.cell[q1], n2 = tv.cell[q2], n3 = tv.cell[q3];
    const unsigned long long s0 = n0.st, s1 = n1.st, s2 = n2.st, s3 = n3.st;
    const int l0 = st_lab(s0), l1 = st_lab(s1), l2 = st_lab(s2), l3 = st_lab(s3);
    // (no early-out for "no labelled neighbour": cells on the work list always have one, and the rule below yields
    // "no decision" anyway if they did not)
    const double v0 = n0.v, v1 = n1.v, v2 = n2.v, v3 = n3.v, vc = tv.cell[c].v;
    const int g0 = gc - tv.X, g1 = gc - 1, g2 = gc + 1, g3 = gc + tv.X;
    int s_lab = 0, pull_lab = 0, pull_ti = 0;
    bool conflict = false, has_pull = false;
    double pull_tv = 0.0;
    unsigned early_u = 0, und = 0;   // bit k: undecided neighbour k (that could pop before this cell)
    // straight-line, select-based evaluation of the four neighbours: this code runs with few active lanes and every
    // divergent branch costs scalar exec-mask bookkeeping -- the kernel is bound by scalar/branch issue, not by math
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long sq = k == 0 ? s0 : (k == 1 ? s1 : (k == 2 ? s2 : s3));
        const int l = k == 0 ? l0 : (k == 1 ? l1 : (k == 2 ? l2 : l3));
        const int gq = k == 0 ? g0 : (k == 1 ? g1 : (k == 2 ? g2 : g3));
        const double tq = k == 0 ? v0 : (k == 1 ? v1 : (k == 2 ? v2 : v3));   // a labelled cell's slot holds its pop-time value
        const bool lab = l > 0, undq = l == 0;
        const int ti = lab ? st_tref(sq) : gq;
        const bool before = tq < vc || (tq == vc && ti < gc);
        const bool first = lab & before;                       // labelled before this cell pops
        conflict |= first & (s_lab != 0) & (s_lab != l);
        s_lab = (first & (s_lab == 0)) ? l : s_lab;
        const bool later = lab & !before;                      // a possible puller
        const bool better = later & (!has_pull | (tq < pull_tv) | ((tq == pull_tv) & (ti < pull_ti)));
        has_pull |= later;
        pull_tv = better ? tq : pull_tv;
        pull_ti = better ? ti : pull_ti;
        pull_lab = better ? l : pull_lab;
        und |= undq ? (1u << k) : 0u;
        early_u |= (undq & before) ? (1u << k) : 0u;
    }
    if (!certs) {
        // the everyday round, branch-free: nothing is decided while an undecided neighbour could pop earlier
        unsigned blk = 0;   // undecided neighbours that could still be labelled before the pull
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double vq = k == 0 ? v0 : (k == 1 ? v1 : (k == 2 ? v2 : v3));
            const int gq = k == 0 ? g0 : (k == 1 ? g1 : (k == 2 ? g2 : g3));
            const bool after_pull = (pull_tv < vq) | ((pull_tv == vq) & (pull_ti < gq));
            blk |= ((((und >> k) & 1u) != 0u) & !after_pull) ? 1u : 0u;
        }
        const bool quiet = early_u == 0;
        const bool ok_normal = (s_lab != 0) & quiet;
        const bool ok_pull = (s_lab == 0) & has_pull & quiet & (blk == 0);
        d.lab = ok_normal ? (conflict ? LINE_LAB : s_lab) : (ok_pull ? pull_lab : 0);
        d.ti = ok_normal ? gc : pull_ti;
        d.tv = ok_normal ? vc : pull_tv;
        return d;
    }
    if (early_u) {
#pragma unroll 1
        for (int k = 0; k < 4; ++k)
            if ((early_u >> k) & 1u) {
                const int q = k == 0 ? q0 : (k == 1 ? q1 : (k == 2 ? q2 : q3));
                if (!ws_cert(tv, q, c, vc, gc)) return d;
            }
    }
    if (s_lab != 0) {
        d.lab = conflict ? LINE_LAB : s_lab;
        d.ti = gc; d.tv = vc;
        return d;
    }
    if (!has_pull) return d;
    // stuck pixel: it is pulled by its earliest-labelled neighbour unless another neighbour can still get there first
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        if (!((und >> k) & 1u)) continue;
        const int q = k == 0 ? q0 : (k == 1 ? q1 : (k == 2 ? q2 : q3));
        const double vq = k == 0 ? v0 : (k == 1 ? v1 : (k == 2 ? v2 : v3));
        const int gq = k == 0 ? g0 : (k == 1 ? g1 : (k == 2 ? g2 : g3));
        if (pull_tv < vq || (pull_tv == vq && pull_ti < gq)) continue;   // q cannot pop before the pull
        if (!certs) return d;
        if (!ws_cert(tv, q, c, pull_tv, pull_ti)) return d;
    }
    d.lab = pull_lab; d.ti = pull_ti; d.tv = pull_tv;
    return d;
}

// One block = one 32x32 tile (+ halo) iterated to its local fixed point.  Work list: only undecided cells that touch a
// labelled cell (the frontier) are evaluated each round; a cell that gets labelled wakes its undecided interior
// neighbours.  Cheap rule while the tile progresses, pocket certificates for one round when it stalls.
// WM > 0: the tile also evaluates a margin of WM cells around its interior (redundantly with its neighbours -- a certified
// decision is the same whoever takes it) and stores every decision straight to global memory: dependency chains that
// zig-zag across a tile border no longer cost one launch per crossing.
// EV = 1: event-driven work list.  A cell that has to wait LEAVES the list and comes back when one of its neighbours is
// decided (label or line) -- instead of being re-evaluated every round until its lower neighbours are through.
template <int WT, int WS_THREADS, int WH, int WK, int WM = 0, int EV = 0>
__global__ void __launch_bounds__(WS_THREADS) k_ws_tiles(const double *__restrict__ v, unsigned long long *__restrict__ st, int Y, int X,
                                                  int tilesX, int tilesY, const unsigned char *__restrict__ changed_prev,
                                                  unsigned char *__restrict__ changed_cur, int *__restrict__ tile_und,
                                                  int *__restrict__ tile_front, int *__restrict__ tile_wst, int first, int max_rounds, int dbg,
                                                  int allow_certs, WsInfo *info)
{
    constexpr int WL = WT + 2 * WH;
    constexpr int WE = WT + 2 * WM, E0 = WH - WM, E1 = WL - E0;    // evaluated region: window rows / columns [E0, E1)
    static_assert(WM >= 0 && WM < WH, "the outermost window ring is read-only");
    __shared__ WCell cells[WL * WL];
    __shared__ unsigned short svis[WS_THREADS * WK];
    __shared__ unsigned short slist[2][WE * WE];
    __shared__ int s_n[2], s_any, s_und, s_chg, s_front;
    const int tile = blockIdx.x, ty = tile / tilesX, tx = tile % tilesX;
    // (every block writes its changed_cur word, also when it has nothing to do: no memset between launches)
    if (first == 2 && tile_und[tile] == 0) { if (threadIdx.x == 0) changed_cur[tile] = 0; return; }  // wide pass: every tile that still has undecided pixels
    if (!first) {
        bool act = tile_und[tile] != 0;
        if (act) {
            act = false;
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i) {
                    const int yy = ty + j, xx = tx + i;
                    if (yy >= 0 && yy < tilesY && xx >= 0 && xx < tilesX) act |= changed_prev[yy * tilesX + xx] != 0;
                }
        }
        if (!act) { if (threadIdx.x == 0) changed_cur[tile] = 0; return; }
    }
    const int gy0 = ty * WT - WH, gx0 = tx * WT - WH;
    int wcount = 0;          // undecided cells in the window at load time
    {   // window load: all state loads of the thread in flight together, then all value loads (the value a labelled
        // cell needs is its pop-time value v[tref]); one wave per tile and few tiles per CU: nothing else hides latency
        constexpr int NLOAD = (WL * WL + WS_THREADS - 1) / WS_THREADS;
        unsigned long long ls[NLOAD];
        double lv[NLOAD];
        int lg[NLOAD];
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) {
            const int c = threadIdx.x + u * WS_THREADS;
            const int ly = c / WL, lx = c - ly * WL;
            const int gy = gy0 + ly, gx = gx0 + lx;
            const bool in = c < WL * WL && gy >= 0 && gy < Y && gx >= 0 && gx < X;
            lg[u] = in ? gy * X + gx : -1;
            ls[u] = st[in ? lg[u] : 0];
            if (!in) ls[u] = pack_st(LINE_LAB, 0);
        }
        // A tile that decided nothing last time -- not even with pocket certificates, which cost ~40 plain rounds -- and is
        // woken by a neighbour's news can only get further if its OWN window has changed.  Decisions are final, so the
        // number of undecided cells in the window is an exact change detector: same count, same window, leave at once.
        if (WS_THREADS == 64 && tile_wst != nullptr) {
#pragma unroll
            for (int u = 0; u < NLOAD; ++u) wcount += st_lab(ls[u]) == 0 ? 1 : 0;
            for (int d = 32; d >= 1; d >>= 1) wcount += __shfl_xor(wcount, d, 64);
            if (!first && tile_wst[tile] == (wcount | WST_STUCK)) {
                if (threadIdx.x == 0) changed_cur[tile] = 0;
                return;
            }
        }
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) {
            const int src = lg[u] < 0 ? 0 : (st_lab(ls[u]) > 0 ? st_tref(ls[u]) : lg[u]);
            lv[u] = v[src];
        }
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) {
            const int c = threadIdx.x + u * WS_THREADS;
            if (c < WL * WL) { cells[c].v = lv[u]; cells[c].st = ls[u]; }
        }
    }
    if (threadIdx.x == 0) { s_n[0] = 0; s_n[1] = 0; s_any = 0; s_und = 0; s_chg = 0; s_front = 0; }
    __syncthreads();
    const int g00 = gy0 * X + gx0;
    TileView tv{cells, svis + threadIdx.x * WK, WK, WL, g00, X};
    // initial frontier: undecided cells of the evaluated region next to a labelled cell
    unsigned was_und = 0;   // (WM == 0) bit k: own interior cell k was undecided when the window was loaded
    if (WM == 0) {
#pragma unroll
        for (int k = 0; k < WT * WT / WS_THREADS; ++k) {
            const int p = threadIdx.x + k * WS_THREADS;
            const int c = (p / WT + WH) * WL + (p % WT + WH);
            if (st_lab(cells[c].st) == 0) {
                was_und |= 1u << k;
                if (st_lab(cells[c - WL].st) > 0 || st_lab(cells[c - 1].st) > 0 || st_lab(cells[c + 1].st) > 0 || st_lab(cells[c + WL].st) > 0) {
                    cells[c].st = ST_LISTED;
                    slist[0][atomicAdd(&s_n[0], 1)] = (unsigned short)c;
                }
            }
        }
    } else {
        for (int p = threadIdx.x; p < WE * WE; p += WS_THREADS) {
            const int c = (p / WE + E0) * WL + (p % WE + E0);
            if (st_lab(cells[c].st) == 0 &&
                (st_lab(cells[c - WL].st) > 0 || st_lab(cells[c - 1].st) > 0 || st_lab(cells[c + 1].st) > 0 || st_lab(cells[c + WL].st) > 0)) {
                cells[c].st = ST_LISTED;      // (still label 0 for the threads that scan its neighbours)
                slist[0][atomicAdd(&s_n[0], 1)] = (unsigned short)c;
            }
        }
    }
    __syncthreads();
    int cur = 0, my_evals = 0, my_rounds = 0;
    bool certs = false;
    if (EV) {
        bool certs_done = false;
        for (int round = 0; round < max_rounds; ++round) {
            int n = s_n[cur];
            if (n == 0) {
                // the list ran dry.  A tile that got nowhere at all tries one round with pocket certificates on its frontier
                // (they cost ~40 plain rounds; a tile that moved is re-run next launch anyway, with its neighbours' news)
                if (certs_done || s_chg > 0 || !allow_certs) break;
                __syncthreads();
                for (int p = threadIdx.x; p < WE * WE; p += WS_THREADS) {
                    const int c = (p / WE + E0) * WL + (p % WE + E0);
                    if (cells[c].st == 0ULL &&
                        (st_lab(cells[c - WL].st) > 0 || st_lab(cells[c - 1].st) > 0 || st_lab(cells[c + 1].st) > 0 || st_lab(cells[c + WL].st) > 0)) {
                        cells[c].st = ST_LISTED;
                        slist[cur][atomicAdd(&s_n[cur], 1)] = (unsigned short)c;
                    }
                }
                __syncthreads();
                certs = true; certs_done = true;
                if (dbg && threadIdx.x == 0) atomicAdd(&info->dbg_certs, 1ULL);
                n = s_n[cur];
                if (n == 0) break;
            }
            my_rounds++;
            if (threadIdx.x == 0) s_n[cur ^ 1] = 0;
            __syncthreads();
#pragma unroll 1
            for (int base = 0; base < n; base += WS_THREADS) {
                const int i = base + threadIdx.x;
                int c = -1;
                Decision dec{0, 0, 0.0};
                if (i < n) {
                    c = slist[cur][i];         // (listed cells are undecided: a cell enters the list once per stay)
                    my_evals++;
                    dec = ws_decide(tv, c, g00 + (c / WL) * X + c % WL, certs);
                }
                __syncthreads();  // every read of this chunk is done
                if (c >= 0 && dec.lab == 0) cells[c].st = 0ULL;   // waits: off the list until a neighbour is decided
                __syncthreads();  // (the drops first: a neighbour decided in this very chunk must be able to wake the cell)
                if (c >= 0 && dec.lab != 0) {
                    cells[c].st = pack_st(dec.lab, dec.ti); cells[c].v = dec.tv;
                    if (WM > 0) st[g00 + (c / WL) * X + c % WL] = pack_st(dec.lab, dec.ti);   // (an undecided cell lies inside the image)
                    atomicAdd(&s_chg, 1);
                    const int cy = c / WL, cx = c - cy * WL;   // c is evaluated: a neighbour is too unless c is on that edge of the region
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int q = k == 0 ? c - WL : (k == 1 ? c - 1 : (k == 2 ? c + 1 : c + WL));
                        const bool inside = k == 0 ? cy > E0 : (k == 1 ? cx > E0 : (k == 2 ? cx < E1 - 1 : cy < E1 - 1));
                        if (inside && atomicCAS(&cells[q].st, 0ULL, ST_LISTED) == 0ULL)
                            slist[cur ^ 1][atomicAdd(&s_n[cur ^ 1], 1)] = (unsigned short)q;
                    }
                }
                __syncthreads();
            }
            cur ^= 1;
            certs = false;
        }
    } else {
        for (int round = 0; round < max_rounds; ++round) {
            const int n = s_n[cur];
            if (n == 0) break;
            my_rounds++;
            if (threadIdx.x == 0) { s_n[cur ^ 1] = 0; s_any = 0; }
            __syncthreads();
            // The list is worked off in chunks of one entry per thread, each chunk committed before the next is evaluated
            // (decisions are certified on the states they read, so committing earlier is just a finer round).  One inlined
            // copy of the flood rule instead of four keeps the kernel small -- it is branchy scalar-heavy code and used to
            // overflow the instruction cache -- and almost every round has a single chunk anyway.
    #pragma unroll 1
            for (int base = 0; base < n; base += WS_THREADS) {
                const int i = base + threadIdx.x;
                int c = -1;
                Decision dec{0, 0, 0.0};
                if (i < n) {
                    const int c0 = slist[cur][i];
                    if (st_lab(cells[c0].st) == 0) { my_evals++; c = c0; dec = ws_decide(tv, c, g00 + (c / WL) * X + c % WL, certs); }
                    // else: decided meanwhile (pushed by a neighbour in the round it was decided itself)
                }
                __syncthreads();  // every read of this chunk is done
                if (c >= 0) {
                    if (dec.lab == 0) {  // still waiting: stays on the frontier
                        slist[cur ^ 1][atomicAdd(&s_n[cur ^ 1], 1)] = (unsigned short)c;
                    } else {
                        cells[c].st = pack_st(dec.lab, dec.ti); cells[c].v = dec.tv;
                        if (WM > 0) st[g00 + (c / WL) * X + c % WL] = pack_st(dec.lab, dec.ti);   // (an undecided cell lies inside the image)
                        s_any = 1;
                        atomicAdd(&s_chg, 1);
                        if (dec.lab > 0) {
                            const int cy = c / WL, cx = c - cy * WL;   // c is evaluated: a neighbour is too unless c is on that edge of the region
    #pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int q = k == 0 ? c - WL : (k == 1 ? c - 1 : (k == 2 ? c + 1 : c + WL));
                                const bool inside = k == 0 ? cy > E0 : (k == 1 ? cx > E0 : (k == 2 ? cx < E1 - 1 : cy < E1 - 1));
                                if (inside && atomicCAS(&cells[q].st, 0ULL, ST_LISTED) == 0ULL)
                                    slist[cur ^ 1][atomicAdd(&s_n[cur ^ 1], 1)] = (unsigned short)q;
                            }
                        }
                    }
                }
                __syncthreads();
            }
            cur ^= 1;
            // every wave reads the round's flags before thread 0 may reset them at the top of the next round (blocks of
            // more than one wave: without the barrier the waves could take different branches here)
            const int any = s_any, chg = s_chg;
            if (WS_THREADS > 64) __syncthreads();
            if (any) { certs = false; continue; }
            if (certs) break;   // nothing moved even with pocket certificates: wait for the neighbours
            // local stall.  Pocket certificates cost ~40 plain rounds, and a tile that has just moved is re-run next launch
            // anyway (with its neighbours' news): only a tile that got nowhere at all tries them.
            if (chg > 0 || !allow_certs) break;
            certs = true;
        }
    }
    __syncthreads();
    // und: undecided cells left; front: those of them that touch a labelled cell.  When no tile changed any more and
    // the frontier is empty everywhere, the serial flood's heap would be empty too: the rest stays 0.
    int und = 0, front = 0;
#pragma unroll
    for (int k = 0; k < WT * WT / WS_THREADS; ++k) {
        if (WM == 0 && !((was_und >> k) & 1u)) continue;   // decided before this launch (cells outside the image are LINE)
        const int p = threadIdx.x + k * WS_THREADS;
        const int c = (p / WT + WH) * WL + (p % WT + WH);
        const unsigned long long sc = cells[c].st;
        if (st_lab(sc) == 0) {
            und++;
            front += st_lab(cells[c - WL].st) > 0 || st_lab(cells[c - 1].st) > 0 || st_lab(cells[c + 1].st) > 0 || st_lab(cells[c + WL].st) > 0;
        } else if (WM == 0) {
            st[(ty * WT + p / WT) * X + tx * WT + p % WT] = sc;   // only this tile writes its interior
        }
    }
    if (und) { atomicAdd(&s_und, und); atomicAdd(&s_front, front); }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_und[tile] = s_und;
        tile_front[tile] = s_front;
        if (tile_wst != nullptr) tile_wst[tile] = (wcount - s_chg) | (s_chg == 0 && allow_certs ? WST_STUCK : 0);   // (stuck = certificates tried)
        changed_cur[tile] = s_chg > 0;
        if (s_chg > 0) atomicAdd(&info->changed_part[tile & 63], s_chg);
        if (dbg) {
            atomicAdd(&info->dbg_rounds, (unsigned long long)my_rounds);
            atomicAdd(&info->dbg_tiles, 1ULL);
            if (s_chg == 0) atomicAdd(&info->dbg_idle, 1ULL);
        }
    }
    if (dbg && my_evals) atomicAdd(&info->dbg_evals, (unsigned long long)my_evals);
}

// ---- mode A endgame: what is still undecided when the tile launches stall are stuck pockets and the pixels that
// wait for them.  Connected components of undecided pixels evolve independently (everything around them is final), so
// each one is finished by ONE wave running the serial rule -- commit the component's smallest pop time, repeat -- on an
// LDS copy of the component.  Components larger than END_CAP are left to the wide tile pass / the serial finish.
constexpr int END_CAP = 512;
constexpr int END_GRID = 4096;     // blocks of the endgame launch: they stride over the device-side component count
constexpr int WS_EARLY_BURST = 1;   // the first endgame runs after this many tile bursts (the first has 10 launches), without waiting for a stall
constexpr int WS_END_STEPS = 32;    // serial commits per component and endgame: clears the stuck seeds, the rest is tile work

struct SameU {
    const unsigned long long *st;
    __device__ __forceinline__ bool valid(int i) const { return st_lab(st[i]) == 0; }
    __device__ __forceinline__ bool same(int, int) const { return true; }
};

__global__ void __launch_bounds__(256) k_end_count(const unsigned long long *__restrict__ st, const int *__restrict__ parent,
                                                   int *__restrict__ cnt, int *__restrict__ isroot, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool u = st_lab(st[i]) == 0;
    isroot[i] = (u && parent[i] == (int)i) ? 1 : 0;
    if (u) atomicAdd(&cnt[parent[i]], 1);
}

// every component root reserves its slice of the cell buffer and its slot in the root list with two atomics (a few
// hundred to a few thousand roots per frame: cheaper than two 4 M-element scans; the order of components is irrelevant)
__global__ void __launch_bounds__(256) k_end_offsets(const int *__restrict__ cnt, const int *__restrict__ isroot,
                                                     int *__restrict__ off, int *__restrict__ roots, int *__restrict__ counters, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !isroot[i]) return;
    off[i] = atomicAdd(&counters[1], cnt[i]);
    roots[atomicAdd(&counters[0], 1)] = (int)i;
}

__global__ void __launch_bounds__(256) k_end_scatter(const unsigned long long *__restrict__ st, const int *__restrict__ parent,
                                                     const int *__restrict__ off, int *__restrict__ cursor,
                                                     int *__restrict__ cells, int *__restrict__ slot,
                                                     long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (st_lab(st[i]) != 0) return;
    const int r = parent[i];
    const int k = atomicAdd(&cursor[r], 1);
    cells[off[r] + k] = (int)i;
    slot[i] = k;
}

// One wave replays the serial flood on one component.  Every cell caches its candidate pop time
//     cand = max(own key, earliest pop time among its labelled neighbours)          (none while it has no labelled one)
// as a sortable 128-bit key (encoded value | reference pixel, own pixel) in the REGISTERS of its owner lane (cell k ->
// lane k % 64, slot k / 64).  Pop times only grow, so a commit can only GIVE a candidate to neighbours that had none: a
// step is a register scan + wave-wide minimum, the flood rule for the winner (all lanes redundantly: the LDS reads
// are broadcasts) and at most four candidate updates.
__global__ void __launch_bounds__(64) k_end_resolve(const double *__restrict__ v, unsigned long long *__restrict__ st, int Y, int X,
                                                    const int *__restrict__ roots, const int *__restrict__ cnt,
                                                    const int *__restrict__ off, const int *__restrict__ cells,
                                                    const int *__restrict__ slot, const int *__restrict__ ncomp_d, int max_steps,
                                                    WsInfo *info)
{
    constexpr int EPL = END_CAP / 64;       // cells per lane
    constexpr unsigned long long NONE = ~0ULL;
    __shared__ double cv[END_CAP];          // value of the cell
    __shared__ int cgi[END_CAP];            // global index
    __shared__ int clab[END_CAP], ctr[END_CAP];   // state: label / 0 / LINE and pop-time reference
    // neighbour tables, [direction][cell] so that lanes walking consecutive cells hit consecutive banks
    __shared__ int cnb[4][END_CAP];         // >= 0 local slot, -1 nothing (outside / line), -2 external labelled cell
    __shared__ double ev[4][END_CAP];       // external labelled neighbour: pop-time value
    __shared__ int etr[4][END_CAP], elab[4][END_CAP];
    // (the grid is launched without knowing the number of components on the host: blocks stride over the device-side count)
    const int ncomp = *ncomp_d;
    for (int comp = blockIdx.x; comp < ncomp; comp += gridDim.x) {
    const int r = roots[comp];
    const int m = cnt[r];
    if (m > END_CAP) { if (threadIdx.x == 0) atomicAdd(&info->end_oversize, m); continue; }
    const int base = off[r];
    const int lane = threadIdx.x;
    // candidate keys: hi = encoded pop-time value (NONE: not a candidate), lo = reference pixel << 32 | own pixel
    // (hi == NONE: lo == 0 "no labelled neighbour yet", lo == 1 "committed")
    unsigned long long ch[EPL], cl[EPL];
#pragma unroll
    for (int u = 0; u < EPL; ++u) {
        ch[u] = NONE; cl[u] = 1;
        const int k = lane + 64 * u;
        if (k >= m) continue;
        const int gi = cells[base + k];
        const int y = gi / X, x = gi - y * X;
        const double kv = v[gi];
        cv[k] = kv; cgi[k] = gi; clab[k] = 0; ctr[k] = 0;
        const int nb[4] = {y > 0 ? gi - X : -1, x > 0 ? gi - 1 : -1, x < X - 1 ? gi + 1 : -1, y < Y - 1 ? gi + X : -1};
        bool has = false; double tv = 0.0; int ti = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int code = -1;
            if (nb[j] >= 0) {
                const unsigned long long sq = st[nb[j]];
                const int l = st_lab(sq);
                if (l == 0) code = slot[nb[j]];
                else if (l > 0) {
                    const int tr = st_tref(sq);
                    const double qv = v[tr];
                    code = -2; elab[j][k] = l; etr[j][k] = tr; ev[j][k] = qv;
                    if (!has || qv < tv || (qv == tv && tr < ti)) { tv = qv; ti = tr; has = true; }
                }
            }
            cnb[j][k] = code;
        }
        // pop time = max(own key, earliest labelled neighbour)
        double pv = kv; int pi = gi;
        if (has && (tv > pv || (tv == pv && ti > pi))) { pv = tv; pi = ti; }
        cl[u] = 0;
        if (has) { ch[u] = enc_f64(pv + 0.0); cl[u] = ((unsigned long long)(unsigned)pi << 32) | (unsigned)gi; }
    }
    __syncthreads();
    int committed = 0;
    for (int step = 0; step < max_steps; ++step) {
        // lane-local best, then wave minimum
        unsigned long long bh = ch[0], bl = cl[0];
        int bu = 0;
#pragma unroll
        for (int u = 1; u < EPL; ++u)
            if (ch[u] < bh || (ch[u] == bh && cl[u] < bl)) { bh = ch[u]; bl = cl[u]; bu = u; }
        const unsigned long long mh = bh, ml = bl;
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long oh = __shfl_xor(bh, d, 64), ol = __shfl_xor(bl, d, 64);
            if (oh < bh || (oh == bh && ol < bl)) { bh = oh; bl = ol; }
        }
        if (bh == NONE) break;  // nothing reachable is left (wave-uniform)
        const int wl = __ffsll((unsigned long long)__ballot(mh == bh && ml == bl)) - 1;   // unique: lo holds the own pixel
        const int k = wl + 64 * __shfl(bu, wl, 64);
        const int bi = (int)(unsigned)(bl >> 32);    // the winner pops at (value bh, reference pixel bi)
        // the flood rule for the winner (same on every lane)
        const double kv = cv[k]; const int ki = cgi[k];
        int s_lab = 0, pull_lab = 0, pull_tr = 0; bool conflict = false, has_pull = false; double pt = 0.0; int pti = 0;
        int codes[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int code = cnb[j][k];
            codes[j] = code;
            double qv; int qi, ql;
            if (code == -2) { qv = ev[j][k]; qi = etr[j][k]; ql = elab[j][k]; }
            else if (code >= 0 && clab[code] > 0) { const int tr = ctr[code]; qi = tr; qv = tr == cgi[code] ? cv[code] : v[tr]; ql = clab[code]; }
            else continue;
            if (qv < kv || (qv == kv && qi < ki)) {
                if (s_lab == 0) s_lab = ql; else if (s_lab != ql) conflict = true;
            } else if (!has_pull || qv < pt || (qv == pt && qi < pti)) { has_pull = true; pt = qv; pti = qi; pull_lab = ql; pull_tr = qi; }
        }
        const int new_lab = s_lab != 0 ? (conflict ? LINE_LAB : s_lab) : pull_lab;
        const int new_tr = s_lab != 0 ? ki : pull_tr;
        // a label (not a line) gives its still candidate-less neighbours a pop time: max(their key, (bh, bi))
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int code = codes[j];
            if (code < 0 || new_lab <= 0) continue;        // wave-uniform
            const unsigned long long qh = enc_f64(cv[code] + 0.0);
            const int qg = cgi[code];
            const bool later = bh > qh || (bh == qh && bi > qg);
            const unsigned long long nh = later ? bh : qh;
            const unsigned long long nl = ((unsigned long long)(unsigned)(later ? bi : qg) << 32) | (unsigned)qg;
            const int ol = code & 63, ou = code >> 6;
#pragma unroll
            for (int u = 0; u < EPL; ++u)
                if (lane == ol && u == ou && ch[u] == NONE && cl[u] == 0) { ch[u] = nh; cl[u] = nl; }
        }
#pragma unroll
        for (int u = 0; u < EPL; ++u)
            if (lane == wl && u == bu) { ch[u] = NONE; cl[u] = 1; }
        if (lane == 0) { clab[k] = new_lab; ctr[k] = new_tr; }
        committed++;
        __syncthreads();
    }
    for (int k = lane; k < m; k += 64)
        if (clab[k] != 0) st[cgi[k]] = pack_st(clab[k], ctr[k]);
    if (lane == 0 && committed) atomicAdd(&info->end_part[comp & 63], committed);
    if (lane == 0 && committed == max_steps) atomicAdd(&info->end_unfinished, 1);   // (may have been finished exactly: harmless)
    __syncthreads();   // the LDS copy is reused by the block's next component
    }
}

// ---- mode B: two-valued image (pl.py:194 floods a {0, 255} boundary image) --------------------------------------------
// Every low-valued pixel is a marker with the same heap key, and every other pixel has the same value, so the serial
// flood is (a) the markers popping in the order the array heap's mechanics give equal keys -- tip_heaporder.hip -- and
// (b) a FIFO: entries of the single remaining level pop in push order.  Push order = (pop rank of the pusher, neighbour
// slot up / left / right / down), so the flood is a breadth-first search in generations whose pixels carry a dense RANK:
// generation g+1's ranks come from sorting (rank of the gen-g pusher) * 4 + slot, done with a flag scatter + scan over
// the 4 n_g possible keys.  When a pixel pops it becomes a line iff the neighbours labelled before it (earlier
// generations, or the same generation with a smaller rank) carry two different labels, else it takes its pusher's label.
//   st[p]   low 32: label / 0 undecided / LINE;  high 32: rank + 1 of a marker or of a candidate (0: not reached yet)
//   cand[p] min over pushes of (key << 32 | pusher's label); ~0: never pushed
constexpr unsigned long long MB_NONE = ~0ULL;

__global__ void __launch_bounds__(256) k_mb_marker_flags(const unsigned long long *__restrict__ st, int *__restrict__ isroot, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) isroot[i] = st_lab(st[i]) > 0 ? 1 : 0;
}

// c[raster rank of the marker pixel] = number of its 4-neighbours inside the image that are not markers
__global__ void __launch_bounds__(256) k_mb_push_counts(const unsigned long long *__restrict__ st, const int *__restrict__ mrank,
                                                        unsigned char *__restrict__ c, int Y, int X)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    const int i = y * X + x;
    if (st_lab(st[i]) <= 0) return;
    int k = 0;
    if (y > 0 && st_lab(st[i - X]) == 0) ++k;
    if (x > 0 && st_lab(st[i - 1]) == 0) ++k;
    if (x < X - 1 && st_lab(st[i + 1]) == 0) ++k;
    if (y < Y - 1 && st_lab(st[i + X]) == 0) ++k;
    c[mrank[i]] = (unsigned char)k;
}

// E[order[t]] = t: pop rank of every marker from the pop sequence
__global__ void __launch_bounds__(256) k_mb_invert(const unsigned *__restrict__ order, unsigned *__restrict__ E, long M)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) E[order[t]] = (unsigned)t;
}

__global__ void __launch_bounds__(256) k_mb_init(unsigned long long *__restrict__ st, const int *__restrict__ mrank,
                                                 const unsigned *__restrict__ E, unsigned long long *__restrict__ cand, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = st_lab(st[i]);
    st[i] = l > 0 ? pack_st(l, (int)(E[mrank[i]] + 1u)) : 0ULL;
    cand[i] = MB_NONE;
}

// a labelled pixel p of rank r pushes its undecided neighbours: key = r * 4 + slot, slot = position of the neighbour in
// skimage's push order (up, left, right, down).  The first push of a pixel appends it to the next generation's list.
// Block-aggregated append: the items of a 256-thread block are collected in LDS and the block reserves its slice of
// the global list with ONE atomic (hundreds of thousands of same-address atomics on the list counter serialise in L2:
// one per lane cost 1.9 ms per frame, one per block costs nothing measurable).
struct BlockList {
    int *items;     // LDS, capacity 4 * 256
    int *count;     // LDS
    int *base;      // LDS
};
__device__ __forceinline__ void bl_init(const BlockList &b)
{
    if (threadIdx.x == 0) *b.count = 0;
    __syncthreads();
}
__device__ __forceinline__ void bl_push(const BlockList &b, int value) { b.items[atomicAdd(b.count, 1)] = value; }
__device__ __forceinline__ void bl_flush(const BlockList &b, int *__restrict__ list, int *__restrict__ counter)
{
    __syncthreads();
    const int n = *b.count;
    if (n == 0) return;
    if (threadIdx.x == 0) *b.base = atomicAdd(counter, n);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) list[*b.base + i] = b.items[i];
}

// a labelled pixel p of rank r pushes its undecided neighbours: key = r * 4 + slot, slot = position of the neighbour in
// skimage's push order (up, left, right, down).  The first push of a pixel appends it to the next generation's list.
__device__ __forceinline__ void mb_push_from(unsigned long long *__restrict__ st, unsigned long long *__restrict__ cand,
                                             const BlockList &bl, int p, int Y, int X)
{
    const unsigned long long s = st[p];
    const int l = st_lab(s);
    if (l <= 0) return;
    const unsigned long long r4 = (unsigned long long)(unsigned)(st_tref(s) - 1) * 4ULL;
    const int y = p / X, x = p - y * X;
    const int nb[4] = {y > 0 ? p - X : -1, x > 0 ? p - 1 : -1, x < X - 1 ? p + 1 : -1, y < Y - 1 ? p + X : -1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int u = nb[k];
        if (u < 0 || st[u] != 0ULL) continue;
        const unsigned long long val = ((r4 + (unsigned long long)k) << 32) | (unsigned)l;
        if (atomicMin(&cand[u], val) == MB_NONE) bl_push(bl, u);
    }
}

__global__ void __launch_bounds__(256) k_mb_push_markers(unsigned long long *__restrict__ st, unsigned long long *__restrict__ cand,
                                                         int *__restrict__ next, int *__restrict__ counter, int Y, int X)
{
    __shared__ int s_items[4 * 256], s_count, s_base;
    const BlockList bl{s_items, &s_count, &s_base};
    bl_init(bl);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x < X) mb_push_from(st, cand, bl, y * X + x, Y, X);
    bl_flush(bl, next, counter);
}

// fate of one pixel of the generation (rank r): true when decided.  When the pixel pops, the neighbours labelled before it
// are those of earlier generations plus the same-generation neighbours of smaller rank that took a label.  A pending
// same-generation neighbour q of smaller rank will end as a line (ignored) or with its pusher's label, which is already
// known (cand[q]): if that label equals the one label this pixel sees, q cannot change the outcome and is not waited for --
// a pixel only waits for smaller-ranked neighbours that would bring a DIFFERENT label, i.e. across a collision front, where
// the chains are two pixels long instead of running along the whole front.
// COH: every load / store goes to the L2 (agent scope), for the one-workgroup kernel that runs whole generations back to back: a cache
// line it read in an earlier generation may be stale in the CU's vector cache once atomics have changed it in the L2.
__device__ __forceinline__ unsigned long long mb_ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int mb_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mb_st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mb_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool COH = false>
__device__ __forceinline__ bool mb_try_resolve(unsigned long long *st, const unsigned long long *cand, int p, int myr, int Y, int X)
{
    volatile unsigned long long *vst = st;
    const int y = p / X, x = p - y * X;
    const int nb[4] = {y > 0 ? p - X : -1, x > 0 ? p - 1 : -1, x < X - 1 ? p + 1 : -1, y < Y - 1 ? p + X : -1};
    int l0 = 0;
    bool diff = false;
    unsigned wait_mask = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (nb[k] < 0) continue;
        const unsigned long long s = COH ? mb_ld(st + nb[k]) : vst[nb[k]];
        const int l = st_lab(s);
        if (l > 0) {
            if (l0 == 0) l0 = l;
            else if (l != l0) diff = true;
        } else if (l == 0) {
            const int r = st_tref(s);
            if (r != 0 && r < myr) wait_mask |= 1u << k;
        }
    }
    if (!diff && wait_mask) {       // (two labels already: a line whatever the pending neighbours become)
        bool pending = false;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((wait_mask >> k) & 1u) pending |= (int)(unsigned)((COH ? mb_ld(cand + nb[k]) : cand[nb[k]]) & 0xffffffffULL) != l0;
        if (pending) return false;
    }
    const unsigned long long out = pack_st(diff ? LINE_LAB : (int)(unsigned)((COH ? mb_ld(cand + p) : cand[p]) & 0xffffffffULL), myr);
    if (COH) mb_st(st + p, out); else vst[p] = out;
    return true;
}

// ---- the generation loop without a host round trip per generation ------------------------------------------------------------------
// The sizes of a generation live on the device (MbState); every kernel reads them there and walks its list with a grid-stride loop,
// so the host queues several generations' launches back to back and looks at the state once per batch (13 generations on a U-Net
// tail frame: two looks instead of thirteen synchronisations).  A generation after the last one is a handful of empty launches.
struct MbState {
    int ncur;        // pixels of the generation that pushes (its ranked list); generation 0: the markers push
    int nnext;       // pixels pushed so far by this generation (append counter of the unordered list)
    int keyspace;    // rank keys of the generation being ranked: 4 x ncur (generation 0: 4 x markers)
    int gen;         // generations completed
    int pcount[4];   // waiting-list counters of the resolve passes
    int flip;        // which of the two ranked-list buffers holds the generation that pushes (0: listA)
    int small_gens;  // generations finished by the one-workgroup kernel (diagnostic)
    int gsize[30];   // pixels of generation 1, 2, ... (diagnostic, TIP_WS_DEBUG)
};

__global__ void k_mb_state_init(MbState *S, int keyspace0)
{
    S->ncur = 0; S->nnext = 0; S->keyspace = keyspace0; S->gen = 0; S->flip = 0; S->small_gens = 0;
    for (int q = 0; q < 4; ++q) S->pcount[q] = 0;
    for (int q = 0; q < 30; ++q) S->gsize[q] = 0;
}

__global__ void __launch_bounds__(256) k_mb_push_list_dn(unsigned long long *__restrict__ st, unsigned long long *__restrict__ cand,
                                                         const int *__restrict__ listA, const int *__restrict__ listB, MbState *S,
                                                         int *__restrict__ next, int Y, int X)
{
    __shared__ int s_items[4 * 256], s_count, s_base;
    const BlockList bl{s_items, &s_count, &s_base};
    const int nlist = S->ncur;
    const int *__restrict__ list = S->flip ? listB : listA;
    for (int j0 = blockIdx.x * blockDim.x; j0 < nlist; j0 += gridDim.x * blockDim.x) {      // (block-uniform trip count: barriers inside)
        bl_init(bl);
        const int i = j0 + threadIdx.x;
        if (i < nlist) mb_push_from(st, cand, bl, list[i], Y, X);
        bl_flush(bl, next, &S->nnext);
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_mb_flag_keys_dn(const unsigned long long *__restrict__ cand, const int *__restrict__ next,
                                                         const MbState *S, int *__restrict__ flag)
{
    const int nnext = S->nnext;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nnext; i += gridDim.x * blockDim.x) flag[(unsigned)(cand[next[i]] >> 32)] = 1;
}

// exclusive scan of the key flags, length on the device: a fixed grid walks the 2048-element chunks; a block first needs the sum of all
// chunks in front of its own, which it accumulates as it goes (its chunks are gridDim.x apart)
constexpr int MBS_ITEMS = 8, MBS_CHUNK = 256 * MBS_ITEMS;
__global__ void __launch_bounds__(256) k_mb_scan_chunks(const int *__restrict__ in, int *__restrict__ out, const MbState *S, int *__restrict__ csum)
{
    __shared__ int wsum[4];
    const int n = S->nnext > 0 ? S->keyspace : 0;
    for (int c0 = blockIdx.x; (long)c0 * MBS_CHUNK < n; c0 += gridDim.x) {
        const long base = (long)c0 * MBS_CHUNK + (long)threadIdx.x * MBS_ITEMS;
        int v[MBS_ITEMS], sum = 0;
#pragma unroll
        for (int i = 0; i < MBS_ITEMS; ++i) {
            v[i] = base + i < n ? in[base + i] : 0;
            sum += v[i];
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        int run = woff + inc - sum;
#pragma unroll
        for (int i = 0; i < MBS_ITEMS; ++i) {
            if (base + i < n) out[base + i] = run;
            run += v[i];
        }
        if (threadIdx.x == 255) csum[c0] = run;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_mb_scan_add(int *__restrict__ out, const MbState *S, const int *__restrict__ csum)
{
    __shared__ int wsum[4];
    __shared__ int s_off;
    const int n = S->nnext > 0 ? S->keyspace : 0;
    int off = 0, done_to = 0;                    // sum of csum[0 .. done_to)
    for (int c0 = blockIdx.x; (long)c0 * MBS_CHUNK < n; c0 += gridDim.x) {
        int part = 0;
        for (int j = done_to + threadIdx.x; j < c0; j += 256) part += csum[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) s_off = off + wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
        off = s_off;
        done_to = c0;
        const long base = (long)c0 * MBS_CHUNK + (long)threadIdx.x * MBS_ITEMS;
#pragma unroll
        for (int i = 0; i < MBS_ITEMS; ++i)
            if (base + i < n) out[base + i] += off;
        __syncthreads();
    }
}

// ranks of the generation; the key flags of the NEXT generation's key space (4 x this generation's pixels) are cleared on the way
__global__ void __launch_bounds__(256) k_mb_assign_ranks_dn(unsigned long long *__restrict__ st, const unsigned long long *__restrict__ cand,
                                                            const int *__restrict__ next, const MbState *S, const int *__restrict__ drank,
                                                            int *__restrict__ listA, int *__restrict__ listB, int *__restrict__ kflag_next)
{
    const int nnext = S->nnext;
    int *__restrict__ list = S->flip ? listA : listB;        // the generation being ranked goes to the OTHER buffer
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nnext; i += gridDim.x * blockDim.x) {
        const int u = next[i];
        const int r = drank[(unsigned)(cand[u] >> 32)];
        st[u] = pack_st(0, r + 1);
        list[r] = u;
    }
    const long nk = 4L * nnext;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += (long)gridDim.x * blockDim.x) kflag_next[i] = 0;
}

// resolve pass p (0: the whole generation; else the waiting list of pass p - 1), device counts, grid-stride
__global__ void __launch_bounds__(256) k_mb_resolve_dn(unsigned long long *__restrict__ st, const unsigned long long *__restrict__ cand,
                                                       const int *__restrict__ listA, const int *__restrict__ listB, MbState *S, int pass,
                                                       const int *__restrict__ src, int Y, int X, int *__restrict__ pend)
{
    __shared__ int s_items[4 * 256], s_count, s_base;
    const BlockList bl{s_items, &s_count, &s_base};
    const int n = pass == 0 ? S->nnext : S->pcount[pass - 1];
    const int *__restrict__ list = S->flip ? listA : listB;
    for (int j0 = blockIdx.x * blockDim.x; j0 < n; j0 += gridDim.x * blockDim.x) {
        bl_init(bl);
        const int j = j0 + threadIdx.x;
        if (j < n) {
            const int i = pass == 0 ? j : src[j];
            const int p = list[i];
            bool waiting = true;
            for (int attempt = 0; attempt < 2 && waiting; ++attempt) waiting = !mb_try_resolve(st, cand, p, i + 1, Y, X);
            if (waiting) bl_push(bl, i);
        }
        bl_flush(bl, pend, &S->pcount[pass]);
        __syncthreads();
    }
}

// the one-block tail of a generation, then the state moves on to the next generation
constexpr int MBT_THREADS = 256;
__global__ void __launch_bounds__(MBT_THREADS) k_mb_resolve_tail_dn(unsigned long long *__restrict__ st, const unsigned long long *__restrict__ cand,
                                                             const int *__restrict__ listA, const int *__restrict__ listB,
                                                             const int *__restrict__ pend, MbState *S, int last_pass, int Y, int X, WsInfo *info)
{
    const int n = S->pcount[last_pass];
    const int *__restrict__ list = S->flip ? listA : listB;
    volatile unsigned long long *vst = st;
    int left = n;
    for (int sweep = 0; sweep <= n && left > 0; ++sweep) {
        int mine = 0;
        for (int j = threadIdx.x; j < n; j += MBT_THREADS) {
            const int i = pend[j], p = list[i];
            if (st_lab(vst[p]) != 0) continue;
            if (!mb_try_resolve(st, cand, p, i + 1, Y, X)) mine = 1;
        }
        __threadfence_block();
        left = __syncthreads_count(mine);
    }
    if (threadIdx.x == 0) {
        if (left) info->unfinished = 1;       // only if the generation is inconsistent (never seen)
        const int nn = S->nnext;
        S->ncur = nn;
        S->keyspace = 4 * nn;
        S->nnext = 0;
        if (nn > 0 && S->gen < 30) S->gsize[S->gen] = nn;
        S->gen += nn > 0 ? 1 : 0;
        S->flip ^= 1;
        for (int q = 0; q < 4; ++q) S->pcount[q] = 0;
    }
}

// Small generations, as many as follow each other, in ONE workgroup: the late generations of a frame are a few hundred to a few thousand
// pixels (the flood's fronts meeting inside the boundary bands), and a generation of the grid-wide path is nine launches whatever its size.
// Here a generation is: push (append counter in LDS), the key flags as BITS in LDS (4 x ncur of them), ranks from a scan of the words'
// population counts, the ranked list, and resolve sweeps until nothing waits -- barriers instead of launches.  The kernel leaves as soon
// as a generation is larger than `small` again (state and key flags as the grid-wide kernels expect them), or when the flood is over.
constexpr int MB_SMALL_DEFAULT = 8192, MB_BATCH_DEFAULT = 4;
// 256 threads and 8 KB of LDS, like every kernel of this flood: a workgroup of that size (<= 64 registers a lane) finds room on a CU
// BESIDE the two waves per SIMD of another frame's convolution kernel (220 registers each of 512); the 1024-thread workgroups these
// two kernels had first waited for a convolution workgroup to retire -- 0.2 ms per launch, 5 ms of latency per frame in the kernel
// trace of the headline.  (Latency only: an A/B on one box shows the same frames/s either way, the frame's worker thread has that slack.)
constexpr int MBG_THREADS = 256, MBG_WORDS = 1024;            // key bits: 4 x ncur <= 32 x MBG_WORDS
constexpr int MB_SMALL_MAX = MBG_WORDS * 32 / 4;
__global__ void __launch_bounds__(MBG_THREADS) k_mb_small_gens(unsigned long long *st, unsigned long long *cand, int *listA, int *listB,
                                                               int *unordered, MbState *S, int small, int *kflag, int Y, int X, WsInfo *info)
{
    __shared__ unsigned bits[MBG_WORDS];
    __shared__ int pre[MBG_WORDS];
    __shared__ int wsum[MBG_THREADS / 64];
    __shared__ int s_nnext;
    int ncur = S->ncur, flip = S->flip, gen = S->gen;
    if (ncur <= 0 || ncur > small) return;                    // (uniform)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int unfinished = 0, done = 0;
    for (;;) {
        int *cur = flip ? listB : listA, *nxt = flip ? listA : listB;
        const int nwords = (4 * ncur + 31) >> 5;
        for (int w = t; w < nwords; w += MBG_THREADS) bits[w] = 0u;
        if (t == 0) s_nnext = 0;
        __syncthreads();
        // push: key = rank of the pusher * 4 + slot (up, left, right, down); the first push of a pixel appends it
        for (int i = t; i < ncur; i += MBG_THREADS) {
            const int p = mb_ld(cur + i);
            const unsigned long long sp = mb_ld(st + p);
            const int l = st_lab(sp);
            if (l <= 0) continue;
            const unsigned long long r4 = (unsigned long long)(unsigned)(st_tref(sp) - 1) * 4ULL;
            const int y = p / X, x = p - y * X;
            const int nb[4] = {y > 0 ? p - X : -1, x > 0 ? p - 1 : -1, x < X - 1 ? p + 1 : -1, y < Y - 1 ? p + X : -1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = nb[k];
                if (u < 0 || mb_ld(st + u) != 0ULL) continue;
                const unsigned long long val = ((r4 + (unsigned long long)k) << 32) | (unsigned)l;
                if (atomicMin(&cand[u], val) == MB_NONE) mb_st(unordered + atomicAdd(&s_nnext, 1), u);
            }
        }
        __threadfence();
        __syncthreads();
        const int nnext = s_nnext;
        if (nnext > 0) {
            for (int i = t; i < nnext; i += MBG_THREADS) {
                const unsigned key = (unsigned)(mb_ld(cand + mb_ld(unordered + i)) >> 32);
                atomicOr(&bits[key >> 5], 1u << (key & 31u));
            }
            __syncthreads();
            int carry = 0;
            for (int base = 0; base < nwords; base += MBG_THREADS) {
                const int w = base + t;
                const int c = w < nwords ? __popc(bits[w]) : 0;
                int inc = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(inc, d, 64);
                    if (lane >= d) inc += o;
                }
                if (lane == 63) wsum[wave] = inc;
                __syncthreads();
                int woff = 0, total = 0;
#pragma unroll
                for (int q = 0; q < MBG_THREADS / 64; ++q) {
                    const int v = wsum[q];
                    woff += q < wave ? v : 0;
                    total += v;
                }
                if (w < nwords) pre[w] = carry + woff + inc - c;
                carry += total;
                __syncthreads();
            }
            for (int i = t; i < nnext; i += MBG_THREADS) {
                const int u = mb_ld(unordered + i);
                const unsigned key = (unsigned)(mb_ld(cand + u) >> 32);
                const int r = pre[key >> 5] + __popc(bits[key >> 5] & ((1u << (key & 31u)) - 1u));
                mb_st(st + u, pack_st(0, r + 1));
                mb_st(nxt + r, u);
            }
            __threadfence();
            __syncthreads();
            int left = nnext;
            for (int sweep = 0; sweep <= nnext && left > 0; ++sweep) {
                int mine = 0;
                for (int i = t; i < nnext; i += MBG_THREADS) {
                    const int p = mb_ld(nxt + i);
                    if (st_lab(mb_ld(st + p)) != 0) continue;
                    if (!mb_try_resolve<true>(st, cand, p, i + 1, Y, X)) mine = 1;
                }
                __threadfence();
                left = __syncthreads_count(mine);
            }
            if (left) unfinished = 1;
            if (t == 0 && gen < 30) S->gsize[gen] = nnext;
            ++gen;
            ++done;
        }
        flip ^= 1;
        ncur = nnext;
        if (ncur == 0 || ncur > small) break;
    }
    for (int i = t; i < 4 * ncur; i += MBG_THREADS) kflag[i] = 0;       // the grid-wide path ranks the next generation: its key flags start clean
    if (t == 0) {
        if (unfinished) info->unfinished = 1;
        S->ncur = ncur; S->keyspace = 4 * ncur; S->nnext = 0; S->gen = gen; S->flip = flip; S->small_gens += done;
        for (int q = 0; q < 4; ++q) S->pcount[q] = 0;
    }
}

__global__ void __launch_bounds__(256) k_ws_emit(const unsigned long long *__restrict__ st, int32_t *__restrict__ out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = st_lab(st[i]);
    out[i] = l > 0 ? l : 0;
}

__global__ void k_ws_info_init(WsInfo *info)
{
    info->emin = ~0ULL; info->emax = 0ULL; info->n_other = 0; info->ties = 0; info->n_markers = 0;
    info->changed = 0; info->undecided = 0; info->unfinished = 0;
    for (int q = 0; q < 64; ++q) info->changed_part[q] = 0;
    info->dbg_rounds = 0; info->dbg_tiles = 0; info->dbg_evals = 0; info->dbg_idle = 0; info->dbg_certs = 0;
    info->end_oversize = 0; info->end_unfinished = 0; info->ncomp = 0; info->ncells = 0; info->und_total = 0; info->front_total = 0;
    for (int q = 0; q < 64; ++q) info->end_part[q] = 0;
}
__global__ void k_ws_changed_reset(WsInfo *info)
{
    info->changed = 0;
    for (int q = 0; q < 64; ++q) info->changed_part[q] = 0;
}
__global__ void k_ws_end_reset(WsInfo *info)
{
    info->end_oversize = 0; info->end_unfinished = 0; info->ncomp = 0; info->ncells = 0;
    for (int q = 0; q < 64; ++q) info->end_part[q] = 0;
}
// totals over the tiles' bookkeeping (a tile that sat a launch out keeps its last count, which is still true: only the
// tile itself decides its interior -- after an endgame every tile is woken and recounts)
__global__ void __launch_bounds__(256) k_ws_tile_totals(const int *__restrict__ tile_und, const int *__restrict__ tile_front, int ntiles,
                                                        WsInfo *info)
{
    __shared__ int su, sf;
    if (threadIdx.x == 0) { su = 0; sf = 0; }
    __syncthreads();
    int u = 0, f = 0;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gridDim.x * blockDim.x) {
        const int a = tile_und[t];
        u += a;
        if (a) f += tile_front[t];
    }
    for (int d = 32; d >= 1; d >>= 1) { u += __shfl_xor(u, d, 64); f += __shfl_xor(f, d, 64); }
    if ((threadIdx.x & 63) == 0 && (u | f)) { atomicAdd(&su, u); atomicAdd(&sf, f); }
    __syncthreads();
    if (threadIdx.x == 0 && (su | sf)) { atomicAdd(&info->und_total, su); atomicAdd(&info->front_total, sf); }
}
__global__ void k_ws_iter_reset(WsInfo *info)
{
    info->und_total = 0; info->front_total = 0;
    info->changed = 0; info->undecided = 0; info->unfinished = 0;
    for (int q = 0; q < 64; ++q) info->changed_part[q] = 0;
    info->dbg_rounds = 0; info->dbg_tiles = 0; info->dbg_evals = 0; info->dbg_idle = 0; info->dbg_certs = 0;
}

int watershed_dev(const double *img, int32_t *labels, int Y, int X, int wsl, int32_t *flags_host)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!img || !labels) return fail(TIP_ERR_ARG, "watershed: null pointer");
    if (!wsl) return fail(TIP_ERR_UNSUPPORTED, "watershed: only watershed_line=True (the reference's call sites)");
    if (Y < 1 || X < 1 || Y > 65535 || (long)Y * X > 2147483647L) return fail(TIP_ERR_ARG, "watershed: bad shape %dx%d", Y, X);
    const long n = (long)Y * X;
    WsGuard ws;
    WsInfo *info = ws.get<WsInfo>(1);
    int *parent = ws.get<int>(n), *flag = ws.get<int>(n), *isroot = ws.get<int>(n), *rank = ws.get<int>(n);
    unsigned long long *st = ws.get<unsigned long long>(n);
    if (!info || !parent || !flag || !isroot || !rank || !st) return TIP_ERR_NOMEM;
    hipStream_t s = c.stream;
    TIP_LAUNCH("ws_info_init", k_ws_info_init, dim3(1), dim3(1), 0, info);
    TIP_LAUNCH("ws_minmax", k_ws_minmax, dim3(min(WS_RED_BLOCKS, cdiv(n, 256))), dim3(256), 0, img, n, info);
    TIP_LAUNCH("ws_count_other", k_ws_count_other, dim3(min(WS_RED_BLOCKS, cdiv(n, 256))), dim3(256), 0, img, n, info);
    // markers
    SameF64 same{img, info};
    int rc = uf_components(same, parent, Y, X);
    if (rc) return rc;
    TIP_HIP(hipMemsetAsync(flag, 0, n * sizeof(int), s));
    TIP_LAUNCH("ws_lower_flags", k_ws_lower_flags, dim3(cdiv(X, 256), Y), dim3(256), 0, img, (const int *)parent, flag, Y, X,
               (const WsInfo *)info);
    TIP_LAUNCH("ws_min_roots", k_ws_min_roots, dim3(cdiv(n, 256)), dim3(256), 0, (const int *)parent, (const int *)flag, isroot, n);
    if ((rc = exclusive_scan_i32(isroot, rank, n, &info->n_markers))) return rc;
    TIP_LAUNCH("ws_init_state", k_ws_init_state, dim3(cdiv(X, 256), Y), dim3(256), 0, img, (const int *)parent, (const int *)flag,
               (const int *)rank, st, Y, X, info);
    WsInfo h;
    TIP_HIP(hipMemcpyAsync(&h, info, sizeof h, hipMemcpyDeviceToHost, s));
    TIP_HIP(hipStreamSynchronize(s));
    for (int q = 0; q < 64; ++q) h.changed += h.changed_part[q];
    c.last_ws_labels = h.n_markers;
    c.last_ws_other = (long)h.n_other;
    int flags = h.ties ? TIP_WS_FLAG_TIES : 0;
    const bool two_valued = h.n_other == 0 && h.emin != h.emax;
    const Tuning &tune = tuning();
    if (h.n_markers > 0 && h.ties && !two_valued && tune.ws_ties != 0) {
        // value ties that are not the two-valued case: the serial (value, age) heap replay (tip_ws_serial.hip) on the
        // markers found above; one download of image + markers, one upload of the labels
        flags |= TIP_WS_FLAG_SERIAL_EXACT;
        TIP_LAUNCH("ws_emit", k_ws_emit, dim3(cdiv(n, 256)), dim3(256), 0, (const unsigned long long *)st, labels, n);
        std::vector<double> himg((size_t)n);
        std::vector<int32_t> hmark((size_t)n), hlab((size_t)n);
        TIP_HIP(hipMemcpyAsync(himg.data(), img, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        TIP_HIP(hipMemcpyAsync(hmark.data(), labels, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        TIP_HIP(hipStreamSynchronize(s));
        if ((rc = flood_exact(himg.data(), hmark.data(), hlab.data(), Y, X))) return rc;
        TIP_HIP(hipMemcpyAsync(labels, hlab.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        TIP_HIP(hipStreamSynchronize(s));   // the host vectors go out of scope
        if (flags_host) *flags_host = flags;
        return TIP_OK;
    }
    if (h.n_markers > 0 && two_valued) {
        flags |= TIP_WS_FLAG_TWO_VALUED;  // mode B
        // (a) pop order of the equal-keyed markers: per-marker push counts -> host recurrence (tip_heaporder.hip) -> ranks
        int *mrank = rank, *total_d = &info->n_markers;    // (n_markers was copied out above; reused as the scan's total)
        TIP_LAUNCH("mb_marker_flags", k_mb_marker_flags, dim3(cdiv(n, 256)), dim3(256), 0, (const unsigned long long *)st, isroot, n);
        if ((rc = exclusive_scan_i32(isroot, mrank, n, total_d))) return rc;
        int M = 0;
        TIP_HIP(hipMemcpyAsync(&M, total_d, sizeof(int), hipMemcpyDeviceToHost, s));
        TIP_HIP(hipStreamSynchronize(s));
        unsigned char *c_d = ws.get<unsigned char>((size_t)M);
        unsigned *E_d = ws.get<unsigned>((size_t)M), *order_d = ws.get<unsigned>((size_t)M);
        unsigned long long *cand = ws.get<unsigned long long>(n);
        int *lists = ws.get<int>((size_t)2 * n), *counter = ws.get<int>(1), *pcount = ws.get<int>(16);
        int *pendA = isroot, *pendB = flag;   // waiting lists of the resolve passes (isroot / flag are free here)
        // rank keys of a generation live in [0, 4 * size of the previous one): the markers first, later at most every
        // other pixel
        const size_t keycap = (size_t)4 * (size_t)std::max<long>(M, n - M) + 4;
        int *kflag = ws.get<int>(keycap), *drank = ws.get<int>(keycap);
        if (!c_d || !E_d || !order_d || !cand || !lists || !counter || !pcount || !kflag || !drank) return TIP_ERR_NOMEM;
        TIP_LAUNCH("mb_push_counts", k_mb_push_counts, dim3(cdiv(X, 256), Y), dim3(256), 0, (const unsigned long long *)st,
                   (const int *)mrank, c_d, Y, X);
        {
            // counts down, pop sequence up, through this thread's pinned staging buffer (asynchronous copies, no per-frame allocation)
            const size_t order_off = ((size_t)M + 63) & ~(size_t)63;
            unsigned char *pin = (unsigned char *)pinned_scratch(order_off + (size_t)M * 4);
            if (!pin) return TIP_ERR_NOMEM;
            uint32_t *horder = reinterpret_cast<uint32_t *>(pin + order_off);
            TIP_HIP(hipMemcpyAsync(pin, c_d, (size_t)M, hipMemcpyDeviceToHost, s));
            TIP_HIP(hipStreamSynchronize(s));
            if ((rc = marker_pop_order(pin, M, horder))) return rc;
            TIP_HIP(hipMemcpyAsync(order_d, horder, (size_t)M * 4, hipMemcpyHostToDevice, s));      // (the buffer is next touched after this frame's later synchronisations)
        }
        TIP_LAUNCH("mb_invert", k_mb_invert, dim3(cdiv(M, 256)), dim3(256), 0, (const unsigned *)order_d, E_d, (long)M);
        TIP_LAUNCH("mb_init", k_mb_init, dim3(cdiv(n, 256)), dim3(256), 0, st, (const int *)mrank, (const unsigned *)E_d, cand, n);
        // (b) generations: sizes on the device (MbState), launches queued MB_BATCH generations at a time, one look at the state per batch
        int *unordered = parent;                            // append buffer of a generation before it is ranked (parent is free here)
        MbState *S = ws.get<MbState>(1);
        const long nm = n - M;                              // non-marker pixels: the most a generation (and all of them together) can hold
        const long keycap_later = 4L * nm + 4;
        int *csum = ws.get<int>((size_t)(std::max<long>(4L * M, keycap_later) / MBS_CHUNK + 2));
        if (!S || !csum) return TIP_ERR_NOMEM;
        TIP_LAUNCH("mb_state_init", k_mb_state_init, dim3(1), dim3(1), 0, S, (int)std::min<long>(4L * M, 0x7fffffffL));
        TIP_HIP(hipMemsetAsync(kflag, 0, (size_t)(4L * M) * sizeof(int), s));
        constexpr int MB_PASSES = 2;                        // (pixels wait only across collision fronts: the second pass is already nearly empty, the tail takes what it leaves)
        const int mb_small = tune.mb_small < 0 ? MB_SMALL_DEFAULT : std::min(tune.mb_small, MB_SMALL_MAX);
        const int mb_batch = tune.mb_batch > 0 ? std::min(tune.mb_batch, 64) : MB_BATCH_DEFAULT;
        const int lgrid = (int)std::max<long>(1, std::min<long>(cdiv(nm, 256), 1024));      // fixed grids, grid-stride loops over device counts
        int *listA = lists, *listB = lists + n;             // ranked lists of the pushing / the pushed generation; MbState::flip says which is which
        MbState hS;
        for (int gen = 0;;) {
            for (int b = 0; b < mb_batch; ++b, ++gen) {
                if (gen == 0)
                    TIP_LAUNCH("mb_push_markers", k_mb_push_markers, dim3(cdiv(X, 256), Y), dim3(256), 0, st, cand, unordered, &S->nnext, Y, X);
                else
                    TIP_LAUNCH("mb_push_list", k_mb_push_list_dn, dim3(lgrid), dim3(256), 0, st, cand, (const int *)listA, (const int *)listB, S,
                               unordered, Y, X);
                TIP_LAUNCH("mb_flag_keys", k_mb_flag_keys_dn, dim3(lgrid), dim3(256), 0, (const unsigned long long *)cand, (const int *)unordered,
                           (const MbState *)S, kflag);
                const long keys = gen == 0 ? 4L * M : keycap_later;
                const int sgrid = (int)std::max<long>(1, std::min<long>(cdiv(keys, MBS_CHUNK), 1024));
                TIP_LAUNCH("mb_scan_chunks", k_mb_scan_chunks, dim3(sgrid), dim3(256), 0, (const int *)kflag, drank, (const MbState *)S, csum);
                TIP_LAUNCH("mb_scan_add", k_mb_scan_add, dim3(sgrid), dim3(256), 0, drank, (const MbState *)S, (const int *)csum);
                TIP_LAUNCH("mb_assign_ranks", k_mb_assign_ranks_dn, dim3(lgrid), dim3(256), 0, st, (const unsigned long long *)cand,
                           (const int *)unordered, (const MbState *)S, (const int *)drank, listA, listB, kflag);
                // fate of the generation: parallel passes that ping-pong the list of waiting pixels, then the one-block tail, which
                // also moves the state on to the next generation
                for (int pass = 0; pass < MB_PASSES; ++pass) {
                    int *dst = pass & 1 ? pendB : pendA;
                    const int *src = pass == 0 ? nullptr : (pass & 1 ? pendA : pendB);
                    TIP_LAUNCH("mb_resolve", k_mb_resolve_dn, dim3(pass == 0 ? lgrid : std::max(1, lgrid >> (2 * pass))), dim3(256), 0, st,
                               (const unsigned long long *)cand, (const int *)listA, (const int *)listB, S, pass, src, Y, X, dst);
                }
                TIP_LAUNCH("mb_resolve_tail", k_mb_resolve_tail_dn, dim3(1), dim3(MBT_THREADS), 0, st, (const unsigned long long *)cand,
                           (const int *)listA, (const int *)listB, (const int *)((MB_PASSES - 1) & 1 ? pendB : pendA), S, MB_PASSES - 1, Y, X, info);
                // whatever small generations follow (usually all that are left) run in one workgroup
                if (mb_small > 0)
                    TIP_LAUNCH("mb_small_gens", k_mb_small_gens, dim3(1), dim3(MBG_THREADS), 0, st, cand, listA, listB, unordered, S, mb_small, kflag, Y, X, info);
            }
            TIP_HIP(hipMemcpyAsync(&hS, S, sizeof hS, hipMemcpyDeviceToHost, s));
            TIP_HIP(hipMemcpyAsync(&h, info, sizeof h, hipMemcpyDeviceToHost, s));      // (the same look: did every generation resolve?)
            TIP_HIP(hipStreamSynchronize(s));
            if (hS.ncur == 0) break;                        // the last generation pushed nothing: the flood is complete
            if (gen > 4 * (Y + X) + 64) return fail(TIP_ERR_HIP, "watershed: the two-valued flood does not terminate");
        }
        if (tune.ws_debug) {
            fprintf(stderr, "[tip] two-valued flood: %d generations (%d in the one-workgroup kernel), sizes", hS.gen, hS.small_gens);
            for (int q = 0; q < 30 && q < hS.gen; ++q) fprintf(stderr, " %d", hS.gsize[q]);
            fprintf(stderr, "\n");
        }
        if (h.unfinished != 0) return fail(TIP_ERR_HIP, "watershed: a generation of the two-valued flood did not resolve");
    } else if (h.n_markers > 0) {
        // everyday tile flavour (tuning hook TIP_WS_TILE): interior edge, halo, evaluated margin
        const int variant = tune.ws_tile >= 0 ? tune.ws_tile : WS_TILE_DEFAULT;
        const int open_a = tune.ws_open_a >= 0 ? tune.ws_open_a : WS_OPEN_A, open_b = tune.ws_open_b >= 0 ? tune.ws_open_b : WS_OPEN_B;
        if (variant < 0 || variant > 15 || open_a < 1 || open_b < 1 || open_a > 64 || open_b > 64)
            return fail(TIP_ERR_ARG, "watershed: bad TIP_WS_TILE / TIP_WS_OPEN");
        const int WTv = variant == 12 || variant == 13 ? 8 : (variant == 3 || (variant >= 8 && variant <= 11) ? 32 : WT_FAST);
        const int tilesX = cdiv(X, WTv), tilesY = cdiv(Y, WTv), ntiles = tilesX * tilesY;
        const int wtilesX = cdiv(X, WT_WIDE), wtilesY = cdiv(Y, WT_WIDE), wntiles = wtilesX * wtilesY;
        unsigned char *wchg = ws.get<unsigned char>((size_t)2 * wntiles);
        int *wtile_und = ws.get<int>((size_t)2 * wntiles);
        if (!wchg || !wtile_und) return TIP_ERR_NOMEM;
        unsigned char *chg = ws.get<unsigned char>((size_t)2 * ntiles);
        int *tile_und = ws.get<int>((size_t)2 * ntiles);  // [0, ntiles) undecided cells per tile, [ntiles, 2 ntiles) its frontier
        int *tile_wst = ws.get<int>((size_t)ntiles);
        if (!chg || !tile_und || !tile_wst) return TIP_ERR_NOMEM;
        TIP_HIP(hipMemsetAsync(chg, 0, (size_t)2 * ntiles, s));
        TIP_HIP(hipMemsetAsync(tile_wst, 0, (size_t)ntiles * sizeof(int), s));
        const int cert_from = tune.ws_cert_from >= 0 ? tune.ws_cert_from : WS_CERT_FROM;
        int *wst_arg = tune.ws_no_skip ? nullptr : tile_wst;     // test hook: re-run stuck tiles on every wake-up
        // extra (unused) dynamic LDS per tile block: fewer resident tiles per CU, room for other frames' kernels (tuning hook)
        const size_t lds_pad = tune.ws_lds_pad > 0 ? (size_t)tune.ws_lds_pad : WS_LDS_PAD;
        int iter = 0, crawl = 0;
        long finished_serially = -1;
        bool wide = false, wide_after_endgame = false;
        int endgames = 0;
        int burst_no = 0;
        bool early_done = false;
        int post_end_burst = -1;   // index of the first burst after the early endgame
        // test hooks (tip_set_tuning): TIP_WS_DEBUG prints per-burst counters, TIP_WS_NO_ENDGAME / TIP_WS_NO_WIDE exercise
        // the stall machinery
        const int dbg = tune.ws_debug;
        const bool no_endgame = tune.ws_no_endgame != 0, no_wide = tune.ws_no_wide != 0;
        int *cursor = nullptr, *cellsbuf = nullptr, *slot = nullptr, *roots = nullptr;   // endgame workspaces
        // one launch of the everyday tiles (activity words ping-pong by launch parity; every block writes its word)
        auto tile_launch = [&](int it) -> int {
            unsigned char *prev = chg + (size_t)(it & 1) * ntiles, *cur = chg + (size_t)((it + 1) & 1) * ntiles;
#define WS_TILE_ARGS img, st, Y, X, tilesX, tilesY, (const unsigned char *)prev, cur, tile_und, tile_und + ntiles, wst_arg, it == 0 ? 1 : 0, 4096, dbg, it >= cert_from ? 1 : 0, info
            switch (variant) {
            case 0: TIP_LAUNCH("ws_tiles", (k_ws_tiles<WT_FAST, WTH_FAST, WH_FAST, WK_FAST>), dim3(ntiles), dim3(WTH_FAST), lds_pad, WS_TILE_ARGS); break;
            case 1: TIP_LAUNCH("ws_tiles", (k_ws_tiles<16, 64, 6, 6, 5>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            case 2: TIP_LAUNCH("ws_tiles", (k_ws_tiles<16, 64, 8, 6, 7>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            case 3: TIP_LAUNCH("ws_tiles", (k_ws_tiles<32, 256, 8, 6, 7>), dim3(ntiles), dim3(256), lds_pad, WS_TILE_ARGS); break;
            case 4: TIP_LAUNCH("ws_tiles", (k_ws_tiles<16, 64, 4, 6, 3>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            case 5: TIP_LAUNCH("ws_tiles", (k_ws_tiles<16, 64, 3, 6, 0, 1>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            case 6: TIP_LAUNCH("ws_tiles", (k_ws_tiles<16, 64, 4, 6, 3, 1>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            case 7: TIP_LAUNCH("ws_tiles", (k_ws_tiles<16, 64, 6, 6, 5, 1>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            case 8: TIP_LAUNCH("ws_tiles", (k_ws_tiles<32, 256, 8, 6, 7, 1>), dim3(ntiles), dim3(256), lds_pad, WS_TILE_ARGS); break;
            case 9: TIP_LAUNCH("ws_tiles", (k_ws_tiles<32, 64, 3, 6, 0, 1>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            case 10: TIP_LAUNCH("ws_tiles", (k_ws_tiles<32, 128, 3, 6, 0, 1>), dim3(ntiles), dim3(128), lds_pad, WS_TILE_ARGS); break;
            case 11: TIP_LAUNCH("ws_tiles", (k_ws_tiles<32, 64, 4, 6, 3, 1>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            case 12: TIP_LAUNCH("ws_tiles", (k_ws_tiles<8, 64, 3, 6, 0, 1>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            case 13: TIP_LAUNCH("ws_tiles", (k_ws_tiles<8, 64, 4, 6, 3, 1>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            case 14: TIP_LAUNCH("ws_tiles", (k_ws_tiles<16, 64, 3, 6, 2, 1>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            default: TIP_LAUNCH("ws_tiles", (k_ws_tiles<16, 64, 3, 6, 1, 1>), dim3(ntiles), dim3(64), lds_pad, WS_TILE_ARGS); break;
            }
#undef WS_TILE_ARGS
            return TIP_OK;
        };
        // the endgame, submitted without a host round trip: components of undecided pixels, their cell lists, and one wave
        // per component replaying the serial rule (the grid strides over the device-side component count); results in
        // info->end_*; every tile is woken afterwards
        auto endgame_submit = [&]() -> int {
            SameU su{st};
            int rc2 = uf_components(su, parent, Y, X);
            if (rc2) return rc2;
            TIP_HIP(hipMemsetAsync(flag, 0, n * sizeof(int), s));      // cnt
            TIP_LAUNCH("ws_end_count", k_end_count, dim3(cdiv(n, 256)), dim3(256), 0, (const unsigned long long *)st,
                       (const int *)parent, flag, isroot, n);
            if (!cursor) {   // taken from the pool once per call
                cursor = ws.get<int>(n); cellsbuf = ws.get<int>(n); slot = ws.get<int>(n); roots = ws.get<int>(n);
            }
            if (!cursor || !cellsbuf || !slot || !roots) return TIP_ERR_NOMEM;
            TIP_LAUNCH("ws_end_reset", k_ws_end_reset, dim3(1), dim3(1), 0, info);
            TIP_LAUNCH("ws_end_offsets", k_end_offsets, dim3(cdiv(n, 256)), dim3(256), 0, (const int *)flag, (const int *)isroot,
                       rank /* start of every component's cells in cellsbuf */, roots, &info->ncomp, n);
            TIP_HIP(hipMemsetAsync(cursor, 0, n * sizeof(int), s));
            TIP_LAUNCH("ws_end_scatter", k_end_scatter, dim3(cdiv(n, 256)), dim3(256), 0, (const unsigned long long *)st,
                       (const int *)parent, (const int *)rank, cursor, cellsbuf, slot, n);
            TIP_LAUNCH("ws_end_resolve", k_end_resolve, dim3(END_GRID), dim3(64), 0, img, st, Y, X, (const int *)roots,
                       (const int *)flag, (const int *)rank, (const int *)cellsbuf, (const int *)slot, (const int *)&info->ncomp,
                       WS_END_STEPS, info);
            TIP_HIP(hipMemsetAsync(chg, 1, (size_t)2 * ntiles, s));
            return TIP_OK;
        };
        for (;; ++iter) {
            TIP_LAUNCH("ws_iter_reset", k_ws_iter_reset, dim3(1), dim3(1), 0, info);
            // The opening is ONE submission with one host check at its end (a host round trip idles the GPU for ~50 us):
            // 10 tile launches (the bulk), the early endgame -- what is left then are a few thousand pixels in long
            // dependency chains that would cost one latency-bound launch per tile border crossed, replayed serially per
            // component instead -- and the 8 launches its dependents need (measured on 2048^2 frames: endgame after 10
            // launches with 32 serial steps per component 2.9 ms per frame; after 6 launches 4.0 ms, after 12 3.05 ms,
            // 512 steps 3.7 ms, no early endgame 3.6 ms).  Later: tile launches in bursts of 2 with a check per burst.
            const bool opening = burst_no == 0 && !no_endgame && !dbg;
            if (opening) {
                for (int rep = 0; rep < open_a; ++rep) if ((rc = tile_launch(iter + rep))) return rc;
                if ((rc = endgame_submit())) return rc;
                TIP_LAUNCH("ws_changed_reset", k_ws_changed_reset, dim3(1), dim3(1), 0, info);
                for (int rep = open_a; rep < open_a + open_b; ++rep) if ((rc = tile_launch(iter + rep))) return rc;
                iter += open_a + open_b - 1;
                burst_no = 2;
                early_done = true;
                endgames++;
            } else {
                // (debug / no-endgame runs open with the plain bursts: 10 launches, then 8 after the early endgame)
                const int burst = wide ? 1 : (burst_no == 0 ? 10 : (burst_no == post_end_burst ? 8 : 2));
                burst_no++;
                for (int rep = 0; rep < burst; ++rep) {
                    if (rep) ++iter;
                    if (!wide) {
                        if ((rc = tile_launch(iter))) return rc;
                    } else {   // wide pass over every 32x32 tile (own bookkeeping arrays); the everyday tiles recount afterwards
                        TIP_LAUNCH("ws_tiles_wide", (k_ws_tiles<WT_WIDE, WTH_WIDE, WH_WIDE, WK_WIDE>), dim3(wntiles), dim3(WTH_WIDE), 0, img,
                                   st, Y, X, wtilesX, wtilesY, (const unsigned char *)wchg, wchg + wntiles, wtile_und, wtile_und + wntiles, (int *)nullptr, 1, 4096, dbg, 1, info);
                        TIP_HIP(hipMemsetAsync(chg + (size_t)((iter + 1) & 1) * ntiles, 1, ntiles, s));
                    }
                }
            }
            TIP_LAUNCH("ws_tile_totals", k_ws_tile_totals, dim3(16), dim3(256), 0, (const int *)tile_und, (const int *)(tile_und + ntiles),
                       ntiles, info);
            TIP_HIP(hipMemcpyAsync(&h, info, sizeof h, hipMemcpyDeviceToHost, s));
            TIP_HIP(hipStreamSynchronize(s));
            for (int q = 0; q < 64; ++q) h.changed += h.changed_part[q];
            if (dbg)
                fprintf(stderr, "ws iter %d %s: tiles %llu (idle %llu, certificate rounds %llu) rounds %llu evals %llu changed %d undecided %d\n",
                        iter, wide ? "wide" : "fast", h.dbg_tiles, h.dbg_idle, h.dbg_certs, h.dbg_rounds, h.dbg_evals, h.changed, h.und_total);
            if (opening) {
                if (h.und_total == 0) break;
                if (h.end_oversize == 0 && h.end_unfinished == 0) break;   // every component was replayed to its end: the rest is unreachable
                if (h.changed > 0) continue;                                // the endgame's dependents are still moving
                // else: quiescent already -- the stall handling below
            }
            const bool early_endgame = !early_done && burst_no >= WS_EARLY_BURST && !wide && !no_endgame;
            // crawl detector: a plateau of equal values floods in raster order under mode A's static keys -- one serial chain
            // that the tiles follow at ~16 pixels per launch.  When several bursts in a row decide less than 1/64 of what is
            // left, the rest goes to the serial finish below instead of thousands of launches.
            crawl = (h.changed > 0 && h.und_total > 2048 && (long)h.changed * 64 < (long)h.und_total) ? crawl + 1 : 0;
            if (h.changed > 0 && !early_endgame && crawl < 6) { wide = false; continue; }
            const bool quiescent = h.changed == 0;
            // (after a wide pass the fine tiles' counts are stale -- too large, never too small)
            const long und_total = h.und_total, front_total = h.front_total;
            if (und_total == 0) break;
            if (crawl >= 6) goto serial_finish;
            // quiescent and no undecided pixel touches a labelled one: what is left is enclosed by lines and stays 0.
            // (After a wide pass the fine tiles' counts are stale, so this shortcut only applies to the fine rounds.)
            if (front_total == 0 && !wide && quiescent) break;
            if (!wide_after_endgame && !no_endgame) {
                // serial rule on every connected component of undecided pixels that fits one wave's LDS copy
                if (!early_done) post_end_burst = burst_no;
                early_done = true;
                if ((rc = endgame_submit())) return rc;
                TIP_HIP(hipMemcpyAsync(&h, info, sizeof h, hipMemcpyDeviceToHost, s));
                TIP_HIP(hipStreamSynchronize(s));
                int end_changed = 0;
                for (int q = 0; q < 64; ++q) end_changed += h.end_part[q];
                if (dbg)
                    fprintf(stderr, "ws endgame: %d components, committed %d, oversize cells %d, unfinished %d\n", h.ncomp, end_changed,
                            h.end_oversize, h.end_unfinished);
                endgames++;
                if (h.end_oversize == 0 && h.end_unfinished == 0) break;   // every component was replayed to its end: the rest is unreachable
                if (end_changed > 0 || !quiescent) {   // oversize components remain: back to the tile rounds for them
                    wide = false;
                    continue;
                }
                // only oversize components remain: wide pass, then the serial finish
                wide_after_endgame = true;
                wide = true;
                continue;
            }
            if (!wide && !no_wide) { wide = true; continue; }  // no progress: one wide launch over every tile
            wide = false;
        serial_finish:
            // still nothing: what is left is a serial dependency chain (plateaus larger than any certificate).  One download,
            // the host stage finishes the flood with the same pop-time rule, one upload -- instead of one committed pixel per
            // host round trip (which took minutes on a noisy integer image).
            {
                std::vector<double> himg((size_t)n);
                std::vector<uint64_t> hst((size_t)n);
                TIP_HIP(hipMemcpyAsync(himg.data(), img, (size_t)n * 8, hipMemcpyDeviceToHost, s));
                TIP_HIP(hipMemcpyAsync(hst.data(), st, (size_t)n * 8, hipMemcpyDeviceToHost, s));
                TIP_HIP(hipStreamSynchronize(s));
                finished_serially = flood_keyed_finish(himg.data(), hst.data(), Y, X);
                TIP_HIP(hipMemcpyAsync(st, hst.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
                TIP_HIP(hipStreamSynchronize(s));
            }
            if (dbg) fprintf(stderr, "ws serial finish: %ld pixels\n", finished_serially);
            break;
        }
        if (finished_serially >= 0)
            flags |= TIP_WS_FLAG_SERIAL_FINISH | (int)(std::min<long>(finished_serially, 0x7fffff) << TIP_WS_FLAG_COUNT_SHIFT);
    }
    TIP_LAUNCH("ws_emit", k_ws_emit, dim3(cdiv(n, 256)), dim3(256), 0, (const unsigned long long *)st, labels, n);
    if (flags_host) *flags_host = flags;
    return TIP_OK;
}

}  // namespace tip

using namespace tip;

extern "C" {

int tip_last_watershed_labels(void) { return ctx().last_ws_labels; }

int tip_watershed_f64_dev(const double *img, int32_t *labels, int y, int x, int wsl, int32_t *flags_host)
{
    return watershed_dev(img, labels, y, x, wsl, flags_host);
}

int tip_watershed_f64(const double *img, int32_t *labels, int y, int x, int wsl, int32_t *flags)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!img || !labels || y < 1 || x < 1) return fail(TIP_ERR_ARG, "tip_watershed_f64: bad arguments");
    const size_t P = (size_t)y * x;
    WsGuard ws;
    double *di = ws.get<double>(P);
    int32_t *dl = ws.get<int32_t>(P);
    if (!di || !dl) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(di, img, P * 8, hipMemcpyHostToDevice, c.stream));
    int rc = watershed_dev(di, dl, y, x, wsl, flags);
    if (rc) return rc;
    TIP_HIP(hipMemcpyAsync(labels, dl, P * 4, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_watershed_segmentation_f64_dev(const double *img, int32_t *labels, int y, int x, double imgthresh, const double *taps,
                                       int ntaps, int block, int32_t *flags_host)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!img || !labels || y < 1 || x < 1) return fail(TIP_ERR_ARG, "tip_watershed_segmentation_f64_dev: bad arguments");
    const size_t P = (size_t)y * x;
    WsGuard ws;
    double *a = ws.get<double>(P), *b = ws.get<double>(P);
    if (!a || !b) return TIP_ERR_NOMEM;
    int rc = tip_local_threshold_f64_dev(img, a, y, x, imgthresh, block);
    if (rc) return rc;
    const double *blurred = a;
    if (taps && ntaps > 0) {
        Taps t;
        if ((rc = make_taps(t, taps, ntaps))) return rc;
        if ((rc = correlate1d_dev(a, b, 1, 1, y, x, 1, t, 0))) return rc;
        if ((rc = correlate1d_dev(b, a, 1, 1, y, x, 2, t, 0))) return rc;
        blurred = a;
    }
    return watershed_dev(blurred, labels, y, x, 1, flags_host);
}

}  // extern "C"
