// tip_props.hip -- per-cell reductions over an int32 label map.
//
//   regionprops: skimage.measure.regionprops_table(labels, [label, area, perimeter, centroid, bbox]) (ti.py:891)
//                + intensity sums for intensity_mean (ti.py:2353)
//   neighbor pairs: the relation Tissue.find_neighbors evaluates with labels[dilated == i] (ti.py:1822-1835)
#include "tip_internal.h"

namespace tip {

struct PropAcc {
    unsigned long long area, sy, sx, p0, p1, p2;
    int ymin, ymax, xmin, xmax;
    double isum;
};

struct PropOut {
    unsigned long long *area, *sumy, *sumx, *pc;  // pc[3*l+k]
    int *bbox;                                    // bbox[4*l+{0..3}] = min_row, min_col, max_row+1, max_col+1
    double *isum;                                 // nullable
};

__device__ __forceinline__ void flush(const PropOut &o, int l, const PropAcc &a, bool has_i)
{
    const int k = l - 1;
    atomicAdd(&o.area[k], a.area);
    atomicAdd(&o.sumy[k], a.sy);
    atomicAdd(&o.sumx[k], a.sx);
    if (a.p0) atomicAdd(&o.pc[3 * k], a.p0);
    if (a.p1) atomicAdd(&o.pc[3 * k + 1], a.p1);
    if (a.p2) atomicAdd(&o.pc[3 * k + 2], a.p2);
    atomicMin(&o.bbox[4 * k], a.ymin);
    atomicMin(&o.bbox[4 * k + 1], a.xmin);
    atomicMax(&o.bbox[4 * k + 2], a.ymax + 1);
    atomicMax(&o.bbox[4 * k + 3], a.xmax + 1);
    if (has_i) atomicAdd(&o.isum[k], a.isum);
}

// Label tile in LDS: PT x PT pixels plus a 2-pixel halo (0 outside the image), loaded with every global load of the
// thread in flight.  Both kernels below were bound by the latency of serial global loads, not by bytes.
constexpr int PT = 64, PH = 2, PL = PT + 2 * PH;

__device__ __forceinline__ void load_label_tile(int *tile, const int32_t *__restrict__ lab, int Y, int X, int ty0, int tx0)
{
    constexpr int N = (PL * PL + 255) / 256;
    int v[N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int c = threadIdx.x + u * 256;
        const int ly = c / PL, lx = c - ly * PL;
        const int gy = ty0 - PH + ly, gx = tx0 - PH + lx;
        v[u] = (c < PL * PL && gy >= 0 && gy < Y && gx >= 0 && gx < X) ? lab[(long)gy * X + gx] : 0;
    }
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int c = threadIdx.x + u * 256;
        if (c < PL * PL) tile[c] = v[u];
    }
}

// border pixel of region l inside the LDS tile (cells outside the image hold 0 and labels are > 0, so the image edge
// counts as outside): image - binary_erosion(image, cross, border_value=0) of skimage.measure.perimeter
__device__ __forceinline__ int tile_border(const int *t, int c, int l)
{
    return t[c] == l && (t[c - PL] != l || t[c + PL] != l || t[c - 1] != l || t[c + 1] != l);
}

// each thread walks PROP_RUN consecutive pixels of a COLUMN of the tile (lanes = columns: conflict-free LDS reads,
// coalesced intensity loads) and adds one set of partial sums per label run to a small per-block table in LDS (a
// 64x64 tile touches a handful of labels); the block then issues ONE set of global atomics per label.  The global
// atomics were the bound: ~3000 per tile before, ~100 now.
constexpr int PROP_RUN = 16;
constexpr int PSLOTS = 64;

struct PropTile {
    int key[PSLOTS];
    unsigned long long area[PSLOTS], sy[PSLOTS], sx[PSLOTS], p0[PSLOTS], p1[PSLOTS], p2[PSLOTS];
    int ymin[PSLOTS], ymax[PSLOTS], xmin[PSLOTS], xmax[PSLOTS];
    double isum[PSLOTS];
};

__device__ __forceinline__ void flush_tile(PropTile &t, const PropOut &o, int l, const PropAcc &a, bool has_i)
{
    int sl = (unsigned)l % PSLOTS, probes = 0;
    for (;; sl = (sl + 1) % PSLOTS) {
        const int k = t.key[sl];
        if (k == l) break;
        if (k == 0) {
            const int old = atomicCAS(&t.key[sl], 0, l);
            if (old == 0 || old == l) break;
        }
        if (++probes == PSLOTS) { flush(o, l, a, has_i); return; }   // more labels than slots in this tile: go global
    }
    atomicAdd(&t.area[sl], a.area);
    atomicAdd(&t.sy[sl], a.sy);
    atomicAdd(&t.sx[sl], a.sx);
    if (a.p0) atomicAdd(&t.p0[sl], a.p0);
    if (a.p1) atomicAdd(&t.p1[sl], a.p1);
    if (a.p2) atomicAdd(&t.p2[sl], a.p2);
    atomicMin(&t.ymin[sl], a.ymin);
    atomicMin(&t.xmin[sl], a.xmin);
    atomicMax(&t.ymax[sl], a.ymax);
    atomicMax(&t.xmax[sl], a.xmax);
    if (has_i) atomicAdd(&t.isum[sl], a.isum);
}

__global__ void __launch_bounds__(256) k_regionprops(const int32_t *__restrict__ lab, const double *__restrict__ inten, int Y,
                                                     int X, int nlab, PropOut o)
{
    __shared__ int tile[PL * PL];
    __shared__ PropTile pt;
    const int tx0 = blockIdx.x * PT, ty0 = blockIdx.y * PT;
    load_label_tile(tile, lab, Y, X, ty0, tx0);
    if (threadIdx.x < PSLOTS) {
        const int k = threadIdx.x;
        pt.key[k] = 0;
        pt.area[k] = pt.sy[k] = pt.sx[k] = pt.p0[k] = pt.p1[k] = pt.p2[k] = 0;
        pt.ymin[k] = 0x7fffffff; pt.xmin[k] = 0x7fffffff; pt.ymax[k] = -1; pt.xmax[k] = -1;
        pt.isum[k] = 0.0;
    }
    __syncthreads();
    const int lx = threadIdx.x & (PT - 1), seg = threadIdx.x / PT;   // 64 columns x 4 segments of 16 rows
    const int x = tx0 + lx;
    const bool has_i = inten != nullptr;
    if (x < X) {
        int cur = 0;
        PropAcc a;
        for (int r = 0; r < PROP_RUN; ++r) {
            const int ly = seg * PROP_RUN + r, y = ty0 + ly;
            if (y >= Y) break;
            const int c = (ly + PH) * PL + lx + PH;
            const int l = tile[c];
            if (l != cur) {
                if (cur > 0 && cur <= nlab) flush_tile(pt, o, cur, a, has_i);
                cur = l;
                a.area = a.sy = a.sx = a.p0 = a.p1 = a.p2 = 0;
                a.ymin = a.ymax = y;
                a.xmin = a.xmax = x;
                a.isum = 0.0;
            }
            if (l <= 0 || l > nlab) continue;
            a.area += 1;
            a.sy += (unsigned long long)y;
            a.sx += (unsigned long long)x;
            a.ymax = y;
            if (has_i) a.isum += inten[(long)y * X + x];
            if (tile_border(tile, c, l)) {
                int code = 1;
                code += 2 * (tile_border(tile, c - PL, l) + tile_border(tile, c + PL, l) + tile_border(tile, c - 1, l) +
                             tile_border(tile, c + 1, l));
                code += 10 * (tile_border(tile, c - PL - 1, l) + tile_border(tile, c - PL + 1, l) +
                              tile_border(tile, c + PL - 1, l) + tile_border(tile, c + PL + 1, l));
                if (code == 5 || code == 7 || code == 15 || code == 17 || code == 25 || code == 27) a.p0++;
                else if (code == 21 || code == 33) a.p1++;
                else if (code == 13 || code == 23) a.p2++;
            }
        }
        if (cur > 0 && cur <= nlab) flush_tile(pt, o, cur, a, has_i);
    }
    __syncthreads();
    if (threadIdx.x < PSLOTS && pt.key[threadIdx.x] != 0) {
        const int k = threadIdx.x;
        PropAcc a;
        a.area = pt.area[k]; a.sy = pt.sy[k]; a.sx = pt.sx[k]; a.p0 = pt.p0[k]; a.p1 = pt.p1[k]; a.p2 = pt.p2[k];
        a.ymin = pt.ymin[k]; a.ymax = pt.ymax[k]; a.xmin = pt.xmin[k]; a.xmax = pt.xmax[k]; a.isum = pt.isum[k];
        if (a.area) flush(o, pt.key[k], a, has_i);
    }
}

__global__ void __launch_bounds__(256) k_props_init(PropOut o, int nlab, int Y, int X)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nlab) return;
    o.area[k] = o.sumy[k] = o.sumx[k] = 0;
    o.pc[3 * k] = o.pc[3 * k + 1] = o.pc[3 * k + 2] = 0;
    o.bbox[4 * k] = Y; o.bbox[4 * k + 1] = X; o.bbox[4 * k + 2] = 0; o.bbox[4 * k + 3] = 0;
    if (o.isum) o.isum[k] = 0.0;
}

__global__ void __launch_bounds__(256) k_props_finish(PropOut o, int nlab, int64_t *__restrict__ bbox64)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 4 * nlab) return;
    bbox64[k] = o.bbox[k];
}

int regionprops_dev(const int32_t *labels, const double *intensity, int Y, int X, int n, int64_t *area, int64_t *bbox4,
                    int64_t *sumy, int64_t *sumx, int64_t *pc3, double *isum)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || !area || !bbox4 || !sumy || !sumx || !pc3) return fail(TIP_ERR_ARG, "regionprops: null pointer");
    if ((intensity == nullptr) != (isum == nullptr)) return fail(TIP_ERR_ARG, "regionprops: intensity and isum go together");
    if (Y < 1 || X < 1 || Y > 65535 || n < 0) return fail(TIP_ERR_ARG, "regionprops: bad shape");
    if (n == 0) return TIP_OK;
    WsGuard ws;
    int *bbox32 = ws.get<int>((size_t)4 * n);
    if (!bbox32) return TIP_ERR_NOMEM;
    PropOut o{(unsigned long long *)area, (unsigned long long *)sumy, (unsigned long long *)sumx,
              (unsigned long long *)pc3, bbox32, isum};
    TIP_LAUNCH("props_init", k_props_init, dim3(cdiv(n, 256)), dim3(256), 0, o, n, Y, X);
    TIP_LAUNCH("regionprops", k_regionprops, dim3(cdiv(X, PT), cdiv(Y, PT)), dim3(256), 0, labels, intensity, Y, X, n, o);
    TIP_LAUNCH("props_finish", k_props_finish, dim3(cdiv(4L * n, 256)), dim3(256), 0, o, n, bbox4);
    return TIP_OK;
}

// ---- neighbour pairs: hash set of (hi<<32 | lo) keys -------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

// insert the ordered pair key into the global hash set: 1 for the call that created the entry, 0 if it was there,
// 2 when every slot was probed without finding the key or a free slot (more distinct pairs than slots: the table has
// >= 2 * cap of them, so the caller's capacity is exceeded anyway and the host reports TIP_ERR_OVERFLOW)
__device__ __forceinline__ int np_insert(unsigned long long key, unsigned long long *__restrict__ table, unsigned long long tmask)
{
    unsigned long long h = mix64(key) & tmask;
    for (unsigned long long probes = 0; probes <= tmask; ++probes) {
        const unsigned long long cur = table[h];
        if (cur == key) return 0;
        if (cur == 0ULL) {
            const unsigned long long old = atomicCAS(&table[h], 0ULL, key);
            if (old == 0ULL) return 1;
            if (old == key) return 0;
        }
        h = (h + 1) & tmask;
    }
    return 2;
}

constexpr int NPSLOTS = 256, NPNEW = 512;

// 5x5 maximum as a horizontal then a vertical 5-maximum on the LDS tile; every thread owns 16 pixels of one column
__global__ void __launch_bounds__(256) k_neighbor_pairs(const int32_t *__restrict__ lab, int Y, int X,
                                                        unsigned long long *__restrict__ table, unsigned long long tmask,
                                                        int32_t *__restrict__ pairs, long long cap,
                                                        unsigned long long *__restrict__ count)
{
    __shared__ int tile[PL * PL];
    __shared__ int hmax[PL * PT];   // horizontal maxima of rows -2 .. PT+1, interior columns only
    __shared__ unsigned long long bset[NPSLOTS], s_new[NPNEW], s_base;
    __shared__ int s_nnew;
    for (int i = threadIdx.x; i < NPSLOTS; i += 256) bset[i] = 0ULL;
    if (threadIdx.x == 0) s_nnew = 0;
    const int tx0 = blockIdx.x * PT, ty0 = blockIdx.y * PT;
    load_label_tile(tile, lab, Y, X, ty0, tx0);   // zero padding takes part (mode='constant'); labels are >= 0 here
    __syncthreads();
    for (int c = threadIdx.x; c < PL * PT; c += 256) {
        const int ly = c / PT, lx = c - ly * PT;
        const int *t = tile + ly * PL + lx + PH;
        hmax[c] = max(max(max(t[-2], t[-1]), max(t[0], t[1])), t[2]);
    }
    __syncthreads();
    const int lx = threadIdx.x & (PT - 1), seg = threadIdx.x / PT;
    const bool in_x = tx0 + lx < X;
    int last_m = -1, last_l = -1;
    for (int r = 0; in_x && r < PT / 4; ++r) {
        const int ly = seg * (PT / 4) + r;
        if (ty0 + ly >= Y) break;
        const int l = tile[(ly + PH) * PL + lx + PH];
        if (l <= 0) continue;
        const int *h = hmax + ly * PT + lx;       // rows ly-2 .. ly+2 of the image = rows ly .. ly+4 of hmax
        const int m = max(max(max(h[0], h[PT]), max(h[2 * PT], h[3 * PT])), h[4 * PT]);
        if (m == l) continue;
        if (m == last_m && l == last_l) continue;   // walking down a cell border the same pair repeats for many rows
        last_m = m; last_l = l;
        // block-level set in LDS first: a tile sees a few dozen distinct pairs, the global table only the first of each
        const unsigned long long key = ((unsigned long long)(unsigned)m << 32) | (unsigned)l;
        int sl = (int)(mix64(key) % NPSLOTS), probes = 0;
        bool fresh = true;
        for (;; sl = (sl + 1) % NPSLOTS) {
            const unsigned long long k = bset[sl];
            if (k == key) { fresh = false; break; }
            if (k == 0ULL) {
                const unsigned long long old = atomicCAS(&bset[sl], 0ULL, key);
                if (old == 0ULL) break;
                if (old == key) { fresh = false; break; }
            }
            if (++probes == NPSLOTS) break;   // set full: treat as new, the global table dedupes
        }
        const int ins = fresh ? np_insert(key, table, tmask) : 0;
        if (ins == 2) atomicAdd(count, (unsigned long long)cap + 1ULL);   // table full: make the host see the overflow
        if (ins == 1) {
            // this thread created the global entry: queue the pair; the block reserves list slots with ONE atomic
            // (a global counter bumped once per pair serialises: 25k same-address atomics cost 0.2 ms)
            const int q = atomicAdd(&s_nnew, 1);
            if (q < NPNEW) s_new[q] = key;
            else {
                const unsigned long long slot = atomicAdd(count, 1ULL);
                if ((long long)slot < cap) { pairs[2 * slot] = m; pairs[2 * slot + 1] = l; }
            }
        }
    }
    __syncthreads();
    const int nnew = min(s_nnew, NPNEW);
    if (threadIdx.x == 0 && nnew > 0) s_base = atomicAdd(count, (unsigned long long)nnew);
    __syncthreads();
    for (int q = threadIdx.x; q < nnew; q += 256) {
        const unsigned long long slot = s_base + q, key = s_new[q];
        if ((long long)slot < cap) { pairs[2 * slot] = (int)(unsigned)(key >> 32); pairs[2 * slot + 1] = (int)(unsigned)key; }
    }
}

int neighbor_pairs_dev(const int32_t *labels, int Y, int X, int32_t *pairs_dev, int64_t cap, int64_t *n_pairs_host)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || !pairs_dev || !n_pairs_host || cap < 1) return fail(TIP_ERR_ARG, "neighbor_pairs: bad arguments");
    if (Y < 1 || X < 1 || Y > 65535) return fail(TIP_ERR_ARG, "neighbor_pairs: bad shape");
    unsigned long long tsize = 1024;
    while (tsize < (unsigned long long)cap * 2) tsize <<= 1;
    WsGuard ws;
    unsigned long long *table = ws.get<unsigned long long>(tsize), *count = ws.get<unsigned long long>(1);
    if (!table || !count) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemsetAsync(table, 0, tsize * 8, c.stream));
    TIP_HIP(hipMemsetAsync(count, 0, 8, c.stream));
    TIP_LAUNCH("neighbor_pairs", k_neighbor_pairs, dim3(cdiv(X, PT), cdiv(Y, PT)), dim3(256), 0, labels, Y, X, table, tsize - 1,
               pairs_dev, (long long)cap, count);
    unsigned long long h = 0;
    TIP_HIP(hipMemcpyAsync(&h, count, 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    *n_pairs_host = (int64_t)h;
    if ((int64_t)h > cap) return fail(TIP_ERR_OVERFLOW, "neighbor_pairs: %lld pairs exceed capacity %lld", (long long)h, (long long)cap);
    return TIP_OK;
}

// ---- contact lengths (ti.py:1844-1872, 4073-4094): histogram of (cross max, cross min) label pairs -----------------
// The reference counts, per cell and neighbour, the pixels of the cell's bounding box (+2) whose 4-neighbour maximum of
// the labels is the larger and whose 4-neighbour minimum of the labels (zeros replaced by max+1) is the smaller of the two
// labels (scipy maximum_filter / minimum_filter with the cross footprint, mode='constant').  Such a pixel touches both
// cells, so it always lies inside that box: the per-pair number is a property of the label map -- one pass, one hash
// histogram keyed (hi << 32 | lo).
__device__ __forceinline__ long np_slot(unsigned long long key, unsigned long long *__restrict__ table, unsigned long long tmask)
{
    unsigned long long h = mix64(key) & tmask;
    for (unsigned long long probes = 0; probes <= tmask; ++probes) {
        const unsigned long long cur = table[h];
        if (cur == key) return (long)h;
        if (cur == 0ULL) {
            const unsigned long long old = atomicCAS(&table[h], 0ULL, key);
            if (old == 0ULL || old == key) return (long)h;
        }
        h = (h + 1) & tmask;
    }
    return -1;
}

__global__ void __launch_bounds__(256) k_contact_pairs(const int32_t *__restrict__ lab, int Y, int X, int big,
                                                       unsigned long long *__restrict__ table, unsigned long long *__restrict__ counts,
                                                       unsigned long long tmask, int *__restrict__ overflow)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= X) return;
    const long i = (long)y * X + x;
    const int nb[4] = {y > 0 ? lab[i - X] : -1, x > 0 ? lab[i - 1] : -1, x < X - 1 ? lab[i + 1] : -1, y < Y - 1 ? lab[i + X] : -1};
    int mx = 0, mn = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int v = nb[k];                      // -1: outside the image = the filters' constant 0
        mx = max(mx, v < 0 ? 0 : v);
        mn = min(mn, v < 0 ? 0 : (v == 0 ? big : v));
    }
    if (mx == mn || mn == 0 || mx == 0) return;   // never asked for: a pair is (larger label, smaller label >= 1)
    const long slot = np_slot(((unsigned long long)(unsigned)mx << 32) | (unsigned)mn, table, tmask);
    if (slot < 0) { atomicOr(overflow, 1); return; }
    atomicAdd(&counts[slot], 1ULL);
}

__global__ void __launch_bounds__(256) k_contact_emit(const unsigned long long *__restrict__ table, const unsigned long long *__restrict__ counts,
                                                      unsigned long long tsize, int32_t *__restrict__ pairs, int64_t *__restrict__ cnt,
                                                      long long cap, unsigned long long *__restrict__ n_out)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tsize) return;
    const unsigned long long key = table[i];
    if (key == 0ULL) return;
    const unsigned long long slot = atomicAdd(n_out, 1ULL);
    if ((long long)slot < cap) {
        pairs[2 * slot] = (int)(unsigned)(key >> 32);
        pairs[2 * slot + 1] = (int)(unsigned)key;
        cnt[slot] = (int64_t)counts[i];
    }
}

int contact_pairs_dev(const int32_t *labels, int Y, int X, int big, int32_t *pairs_dev, int64_t *cnt_dev, int64_t cap,
                      int64_t *n_pairs_host)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || !pairs_dev || !cnt_dev || !n_pairs_host || cap < 1) return fail(TIP_ERR_ARG, "contact_pairs: bad arguments");
    if (Y < 1 || X < 1 || Y > 65535) return fail(TIP_ERR_ARG, "contact_pairs: bad shape");
    unsigned long long tsize = 1024;
    while (tsize < (unsigned long long)cap * 2) tsize <<= 1;
    WsGuard ws;
    unsigned long long *table = ws.get<unsigned long long>(tsize), *counts = ws.get<unsigned long long>(tsize),
                       *n_d = ws.get<unsigned long long>(1);
    int *ovf = ws.get<int>(1);
    if (!table || !counts || !n_d || !ovf) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemsetAsync(table, 0, tsize * 8, c.stream));
    TIP_HIP(hipMemsetAsync(counts, 0, tsize * 8, c.stream));
    TIP_HIP(hipMemsetAsync(n_d, 0, 8, c.stream));
    TIP_HIP(hipMemsetAsync(ovf, 0, 4, c.stream));
    TIP_LAUNCH("contact_pairs", k_contact_pairs, dim3(cdiv(X, 256), Y), dim3(256), 0, labels, Y, X, big, table, counts, tsize - 1, ovf);
    TIP_LAUNCH("contact_emit", k_contact_emit, dim3(cdiv((long)tsize, 256)), dim3(256), 0, (const unsigned long long *)table,
               (const unsigned long long *)counts, tsize, pairs_dev, cnt_dev, (long long)cap, n_d);
    unsigned long long h = 0;
    int hov = 0;
    TIP_HIP(hipMemcpyAsync(&h, n_d, 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(&hov, ovf, 4, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    *n_pairs_host = (int64_t)h;
    if (hov || (int64_t)h > cap)
        return fail(TIP_ERR_OVERFLOW, "contact_pairs: more than %lld distinct label pairs", (long long)cap);
    return TIP_OK;
}

}  // namespace tip

using namespace tip;

extern "C" {

int tip_regionprops_i32_dev(const int32_t *labels, const double *intensity, int y, int x, int n, int64_t *area,
                            int64_t *bbox4, int64_t *sumy, int64_t *sumx, int64_t *pc3, double *isum)
{
    return regionprops_dev(labels, intensity, y, x, n, area, bbox4, sumy, sumx, pc3, isum);
}

int tip_regionprops_i32(const int32_t *labels, const double *intensity, int y, int x, int n, int64_t *area, int64_t *bbox4,
                        int64_t *sumy, int64_t *sumx, int64_t *pc3, double *isum)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || y < 1 || x < 1 || n < 0) return fail(TIP_ERR_ARG, "tip_regionprops_i32: bad arguments");
    if (n == 0) return TIP_OK;
    const size_t P = (size_t)y * x;
    WsGuard ws;
    int32_t *dl = ws.get<int32_t>(P);
    double *di = intensity ? ws.get<double>(P) : nullptr;
    int64_t *da = ws.get<int64_t>(n), *db = ws.get<int64_t>((size_t)4 * n), *dsy = ws.get<int64_t>(n),
            *dsx = ws.get<int64_t>(n), *dp = ws.get<int64_t>((size_t)3 * n);
    double *dis = intensity ? ws.get<double>(n) : nullptr;
    if (!dl || !da || !db || !dsy || !dsx || !dp || (intensity && (!di || !dis))) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(dl, labels, P * 4, hipMemcpyHostToDevice, c.stream));
    if (intensity) TIP_HIP(hipMemcpyAsync(di, intensity, P * 8, hipMemcpyHostToDevice, c.stream));
    int rc = regionprops_dev(dl, di, y, x, n, da, db, dsy, dsx, dp, dis);
    if (rc) return rc;
    TIP_HIP(hipMemcpyAsync(area, da, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(bbox4, db, (size_t)n * 32, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(sumy, dsy, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(sumx, dsx, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(pc3, dp, (size_t)n * 24, hipMemcpyDeviceToHost, c.stream));
    if (isum) TIP_HIP(hipMemcpyAsync(isum, dis, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_neighbor_pairs_i32_dev(const int32_t *labels, int y, int x, int32_t *pairs_dev, int64_t cap, int64_t *n_pairs_host)
{
    return neighbor_pairs_dev(labels, y, x, pairs_dev, cap, n_pairs_host);
}

int tip_neighbor_pairs_i32(const int32_t *labels, int y, int x, int32_t *pairs, int64_t cap, int64_t *n_pairs)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || !pairs || !n_pairs || y < 1 || x < 1 || cap < 1) return fail(TIP_ERR_ARG, "tip_neighbor_pairs_i32: bad arguments");
    const size_t P = (size_t)y * x;
    WsGuard ws;
    int32_t *dl = ws.get<int32_t>(P), *dp = ws.get<int32_t>((size_t)2 * cap);
    if (!dl || !dp) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(dl, labels, P * 4, hipMemcpyHostToDevice, c.stream));
    int rc = neighbor_pairs_dev(dl, y, x, dp, cap, n_pairs);
    if (rc) return rc;
    TIP_HIP(hipMemcpyAsync(pairs, dp, (size_t)(*n_pairs) * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_contact_pairs_i32(const int32_t *labels, int y, int x, int big, int32_t *pairs, int64_t *counts, int64_t cap, int64_t *n_pairs)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!labels || !pairs || !counts || !n_pairs || y < 1 || x < 1 || cap < 1) return fail(TIP_ERR_ARG, "tip_contact_pairs_i32: bad arguments");
    const size_t P = (size_t)y * x;
    WsGuard ws;
    int32_t *dl = ws.get<int32_t>(P), *dp = ws.get<int32_t>((size_t)2 * cap);
    int64_t *dc = ws.get<int64_t>((size_t)cap);
    if (!dl || !dp || !dc) return TIP_ERR_NOMEM;
    TIP_HIP(hipMemcpyAsync(dl, labels, P * 4, hipMemcpyHostToDevice, c.stream));
    int rc = contact_pairs_dev(dl, y, x, big, dp, dc, cap, n_pairs);
    if (rc) return rc;
    TIP_HIP(hipMemcpyAsync(pairs, dp, (size_t)(*n_pairs) * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipMemcpyAsync(counts, dc, (size_t)(*n_pairs) * 8, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

}  // extern "C"
