/*
 * tip_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the third-party arithmetic the reference's hot path
 * relies on (scipy.ndimage 1.7.1 / scikit-image 0.18.3 semantics at the
 * reference's call sites).  It is pinned against golden
 * vectors produced by running the reference itself (tools/make_goldens.py ->
 * the .npz files under tests/golden).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path never does.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off matters: scipy's x86-64 wheels have no FMA contraction, and
 * bit parity with them requires separately rounded multiply and add.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* correlate1d: scipy/ndimage/src/ni_filters.c NI_Correlate1D semantics.       */
/* Reference call sites: bim.py:389 gaussian_filter(mode='nearest') via        */
/* sp.py:37,55,70,71 ; ti.py:142 (blur sigma 7).                               */
/* A line is copied into a double buffer extended by the border mode; the      */
/* symmetric branch sums  x[c]*w[c] + sum_{j=size1..1} (x[c-j]+x[c+j])*w[c-j]  */
/* in double, farthest tap first, and the result is cast to the array dtype.   */
/* ------------------------------------------------------------------------- */
enum { ORC_NEAREST = 0, ORC_REFLECT = 1, ORC_CONSTANT = 2, ORC_MIRROR = 3, ORC_WRAP = 4 };

static void extend_line(double *buf, long len, long s1, long s2, int mode, double cval)
{
    /* buf[s1 .. s1+len) holds the line; fill buf[0..s1) and buf[s1+len .. s1+len+s2) */
    double *first = buf + s1, *last = first + len;
    long i;
    switch (mode) {
    case ORC_NEAREST:
        for (i = 0; i < s1; i++) buf[i] = first[0];
        for (i = 0; i < s2; i++) last[i] = last[-1];
        break;
    case ORC_CONSTANT:
        for (i = 0; i < s1; i++) buf[i] = cval;
        for (i = 0; i < s2; i++) last[i] = cval;
        break;
    case ORC_REFLECT: /* d c b a | a b c d | d c b a */
        for (i = 0; i < s1; i++) {
            long k = i % (2 * len);
            long src = k < len ? k : 2 * len - 1 - k;
            first[-1 - i] = first[src];
        }
        for (i = 0; i < s2; i++) {
            long k = i % (2 * len);
            long src = k < len ? len - 1 - k : k - len;
            last[i] = first[src];
        }
        break;
    case ORC_MIRROR: /* d c b | a b c d | c b a */
        if (len == 1) {
            for (i = 0; i < s1; i++) buf[i] = first[0];
            for (i = 0; i < s2; i++) last[i] = first[0];
        } else {
            long p = 2 * len - 2;
            for (i = 0; i < s1; i++) {
                long k = (i + 1) % p;
                long src = k < len ? k : p - k;
                first[-1 - i] = first[src];
            }
            for (i = 0; i < s2; i++) {
                long k = (i + 1) % p;
                long src = k < len ? len - 1 - k : k - len + 1;
                last[i] = first[src];
            }
        }
        break;
    case ORC_WRAP:
        for (i = 0; i < s1; i++) first[-1 - i] = first[((len - 1 - i) % len + len) % len];
        for (i = 0; i < s2; i++) last[i] = first[i % len];
        break;
    }
}

/* dtype: 0 = float32, 1 = float64.  dims[3] (use 1 for unused leading dims). */
ORC_API int orc_correlate1d(const void *in, void *out, int dtype, const long *dims, int axis,
                            const double *w, long n, int mode, double cval)
{
    long s1 = n / 2, s2 = n - s1 - 1;
    long len = dims[axis];
    long stride = 1, outer = 1, inner = 1;
    int a;
    for (a = axis + 1; a < 3; a++) inner *= dims[a];
    for (a = 0; a < axis; a++) outer *= dims[a];
    stride = inner;
    /* symmetry test as scipy: odd size and |w[i]-w[n-1-i]| <= DBL_EPSILON */
    int symmetric = 0;
    if (n & 1) {
        symmetric = 1;
        for (long i = 1; i <= n / 2; i++)
            if (fabs(w[i + s1] - w[s1 - i]) > DBL_EPSILON) { symmetric = 0; break; }
        if (!symmetric) {
            symmetric = -1;
            for (long i = 1; i <= n / 2; i++)
                if (fabs(w[s1 + i] + w[s1 - i]) > DBL_EPSILON) { symmetric = 0; break; }
        }
    }
    double *buf = (double *)malloc(sizeof(double) * (size_t)(len + s1 + s2));
    if (!buf) return -1;
    const double *fw = w + s1;
    for (long o = 0; o < outer; o++) {
        for (long q = 0; q < inner; q++) {
            long base = o * len * inner + q;
            if (dtype == 0) {
                const float *p = (const float *)in + base;
                for (long l = 0; l < len; l++) buf[s1 + l] = (double)p[l * stride];
            } else {
                const double *p = (const double *)in + base;
                for (long l = 0; l < len; l++) buf[s1 + l] = p[l * stride];
            }
            extend_line(buf, len, s1, s2, mode, cval);
            for (long l = 0; l < len; l++) {
                const double *il = buf + s1 + l;
                double tmp;
                if (symmetric > 0) {
                    tmp = il[0] * fw[0];
                    for (long j = -s1; j < 0; j++) tmp += (il[j] + il[-j]) * fw[j];
                } else if (symmetric < 0) {
                    tmp = il[0] * fw[0];
                    for (long j = -s1; j < 0; j++) tmp += (il[j] - il[-j]) * fw[j];
                } else {
                    tmp = il[s2] * fw[s2];
                    for (long j = -s1; j < s2; j++) tmp += il[j] * fw[j];
                }
                if (dtype == 0) ((float *)out)[base + l * stride] = (float)tmp;
                else ((double *)out)[base + l * stride] = tmp;
            }
        }
    }
    free(buf);
    return 0;
}

/* numpy's pairwise float64 sum (numpy/core/src/umath/loops_utils.h.src) so that the Gaussian
 * normalisation  phi / phi.sum()  (scipy/ndimage/filters.py:_gaussian_kernel1d) is bit-identical. */
static double np_pairwise_sum(const double *a, long n)
{
    if (n < 8) {
        double r = 0.0; /* numpy starts from -0.0 for an exact identity; +0.0 gives the same sum here */
        for (long i = 0; i < n; i++) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

/* Gaussian weights as scipy builds them: radius=int(truncate*sigma+0.5), exp(-0.5/sigma^2 * x^2)/sum.
 * Returns the tap count (2*radius+1); w must hold that many doubles. */
ORC_API long orc_gaussian_weights(double sigma, double truncate, double *w, long cap)
{
    long radius = (long)(truncate * sigma + 0.5);
    long n = 2 * radius + 1;
    if (n > cap) return -n;
    double s2 = sigma * sigma;
    for (long i = 0; i < n; i++) {
        double x = (double)(i - radius);
        w[i] = exp(-0.5 / s2 * (x * x));
    }
    double sum = np_pairwise_sum(w, n);
    for (long i = 0; i < n; i++) w[i] = w[i] / sum;
    return n;
}

/* ------------------------------------------------------------------------- */
/* min / max rank filters: scipy.ndimage.maximum_filter / minimum_filter       */
/* (ti.py:1822,2081,2969,4079-4084) and skimage.morphology.erosion/dilation    */
/* (= grey min/max with a flat footprint, mode='reflect'; pl.py:170-193), and  */
/* threshold_local(method='generic', max callback) = max filter, reflect       */
/* (bim.py:468-472).  Window for output i spans i-k/2 .. i-k/2+k-1 (origin 0). */
/* mode: ORC_CONSTANT (cval 0) or ORC_REFLECT.                                 */
/* ------------------------------------------------------------------------- */
static inline long reflect_idx(long i, long n)
{
    if (n == 1) return 0;
    long p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

#define DEF_MINMAX(NAME, T)                                                                         \
    ORC_API int NAME(const T *in, T *out, long ny, long nx, long ky, long kx, const uint8_t *fp,     \
                     int mode, int is_max)                                                          \
    {                                                                                               \
        long oy = ky / 2, ox = kx / 2;                                                              \
        for (long y = 0; y < ny; y++)                                                               \
            for (long x = 0; x < nx; x++) {                                                         \
                int have = 0;                                                                       \
                T best = 0;                                                                         \
                for (long j = 0; j < ky; j++)                                                       \
                    for (long i = 0; i < kx; i++) {                                                 \
                        if (fp && !fp[j * kx + i]) continue;                                        \
                        long yy = y - oy + j, xx = x - ox + i;                                      \
                        T v;                                                                        \
                        if (mode == ORC_CONSTANT) {                                                 \
                            v = (yy < 0 || yy >= ny || xx < 0 || xx >= nx) ? (T)0 : in[yy * nx + xx]; \
                        } else {                                                                    \
                            v = in[reflect_idx(yy, ny) * nx + reflect_idx(xx, nx)];                 \
                        }                                                                           \
                        if (!have) { best = v; have = 1; }                                          \
                        else if (is_max ? (v > best) : (v < best)) best = v;                        \
                    }                                                                               \
                out[y * nx + x] = best;                                                             \
            }                                                                                       \
        return 0;                                                                                   \
    }
DEF_MINMAX(orc_minmax2d_f64, double)
DEF_MINMAX(orc_minmax2d_i32, int32_t)

/* ------------------------------------------------------------------------- */
/* 4-connected component labelling, skimage.measure.label(connectivity=1)     */
/* semantics (ti.py:2922, 3470): pixels are connected when adjacent AND equal; */
/* `bg`-valued pixels get 0; labels are 1..n in raster order of each           */
/* component's first pixel.  scipy.ndimage.label on a boolean image is the     */
/* same thing with the input binarised (watershed markers).                    */
/* ------------------------------------------------------------------------- */
static long uf_find(int32_t *p, long i)
{
    long r = i;
    while (p[r] != r) r = p[r];
    while (p[i] != r) { long n = p[i]; p[i] = (int32_t)r; i = n; }
    return r;
}
ORC_API long orc_label4_i32(const int32_t *in, int32_t bg, int32_t *out, long ny, long nx)
{
    long n = ny * nx;
    int32_t *par = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!par) return -1;
    for (long i = 0; i < n; i++) par[i] = (int32_t)i;
    for (long y = 0; y < ny; y++)
        for (long x = 0; x < nx; x++) {
            long i = y * nx + x;
            if (in[i] == bg) continue;
            if (x > 0 && in[i - 1] == in[i]) {
                long a = uf_find(par, i), b = uf_find(par, i - 1);
                if (a != b) { if (a < b) par[b] = (int32_t)a; else par[a] = (int32_t)b; }
            }
            if (y > 0 && in[i - nx] == in[i]) {
                long a = uf_find(par, i), b = uf_find(par, i - nx);
                if (a != b) { if (a < b) par[b] = (int32_t)a; else par[a] = (int32_t)b; }
            }
        }
    /* roots are the raster-first pixel of each component (min index wins every union) */
    long next = 0;
    for (long i = 0; i < n; i++) {
        if (in[i] == bg) { out[i] = 0; continue; }
        long r = uf_find(par, i);
        if (r == i) out[i] = (int32_t)(++next);
        else out[i] = out[r];
    }
    free(par);
    return next;
}

/* ------------------------------------------------------------------------- */
/* local minima, skimage.morphology.local_minima(connectivity=1,              */
/* allow_borders=True) (skimage/morphology/extrema.py:272-432 read as text;    */
/* _extrema_cy is binary-only).  A 4-connected plateau of equal value is a     */
/* minimum iff every 4-neighbour of the plateau is strictly greater.  With     */
/* allow_borders the image is padded with the value that can never beat a      */
/* candidate, except that a plateau EQUAL to the pad value touching the border */
/* is rejected -- the pad value is max(image), so that only hits a plateau at  */
/* the global maximum (i.e. a constant image).  Images with any dim < 1 after  */
/* padding rules: skimage pads first, so 1xN images are evaluated normally.    */
/* ------------------------------------------------------------------------- */
ORC_API int orc_local_minima_f64(const double *img, uint8_t *out, long ny, long nx)
{
    long n = ny * nx;
    double gmax = img[0];
    for (long i = 1; i < n; i++) if (img[i] > gmax) gmax = img[i];
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    uint8_t *seen = (uint8_t *)calloc((size_t)n, 1);
    if (!stack || !seen) return -1;
    memset(out, 0, (size_t)n);
    for (long s = 0; s < n; s++) {
        if (seen[s]) continue;
        double h = img[s];
        long top = 0, cnt = 0;
        int is_min = 1;
        stack[top++] = (int32_t)s;
        seen[s] = 1;
        /* flood the plateau; remember members by re-walking with a second pass */
        long start_cnt = 0;
        (void)start_cnt;
        /* we reuse `out` as a temporary member flag (2) */
        while (top) {
            long i = stack[--top];
            out[i] = 2;
            cnt++;
            long y = i / nx, x = i % nx;
            const long nb[4] = { y > 0 ? i - nx : -1, x > 0 ? i - 1 : -1, x < nx - 1 ? i + 1 : -1, y < ny - 1 ? i + nx : -1 };
            for (int k = 0; k < 4; k++) {
                long j = nb[k];
                if (j < 0) { if (h == gmax) is_min = 0; continue; } /* pad pixel has value gmax */
                if (img[j] == h) {
                    if (!seen[j]) { seen[j] = 1; stack[top++] = (int32_t)j; }
                } else if (img[j] < h) is_min = 0;
            }
        }
        /* second walk to finalise flags of this plateau */
        top = 0;
        stack[top++] = (int32_t)s;
        out[s] = is_min ? 1 : 0;
        while (top) {
            long i = stack[--top];
            long y = i / nx, x = i % nx;
            const long nb[4] = { y > 0 ? i - nx : -1, x > 0 ? i - 1 : -1, x < nx - 1 ? i + 1 : -1, y < ny - 1 ? i + nx : -1 };
            for (int k = 0; k < 4; k++) {
                long j = nb[k];
                if (j >= 0 && out[j] == 2) { out[j] = is_min ? 1 : 0; stack[top++] = (int32_t)j; }
            }
        }
    }
    free(stack);
    free(seen);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* watershed: skimage.segmentation.watershed(image, markers, connectivity=1,   */
/* watershed_line=wsl) (bim.py:475, pl.py:194).  skimage 0.18.3 ships          */
/* _watershed_cy as a binary only; this restates its published algorithm       */
/* (skimage/segmentation/_watershed_cy.pyx + heap_general.pxi upstream):       */
/* a binary min-heap of (value, age, index, source); all marker pixels are     */
/* pushed in raster order with age 0; pop; if wsl: skip if already labelled    */
/* (non-marker), turn into a line (mask off) if its labelled neighbours        */
/* disagree, else take the source's label; push every unlabelled in-mask       */
/* neighbour in the order up, left, right, down with age = ++counter.          */
/* Heap tie behaviour (equal value and age) follows the upstream sift rules.   */
/* Pinned empirically by tests/golden/watershed.npz.                           */
/* ------------------------------------------------------------------------- */
typedef struct { double value; int64_t age; int32_t index; int32_t source; } heapitem;
typedef struct { heapitem *d; long items, space; } heap_t;

static inline int h_smaller(const heapitem *a, const heapitem *b)
{
    if (a->value != b->value) return a->value < b->value;
    return a->age < b->age;
}
static int h_push(heap_t *h, const heapitem *e)
{
    if (h->items == h->space) {
        long ns = h->space * 2;
        heapitem *nd = (heapitem *)realloc(h->d, sizeof(heapitem) * (size_t)ns);
        if (!nd) return -1;
        h->d = nd; h->space = ns;
    }
    long child = h->items;
    h->d[child] = *e;
    h->items++;
    while (child > 0) {
        long parent = (child + 1) / 2 - 1;
        if (h_smaller(&h->d[child], &h->d[parent])) {
            heapitem t = h->d[parent]; h->d[parent] = h->d[child]; h->d[child] = t;
            child = parent;
        } else break;
    }
    return 0;
}
static void h_pop(heap_t *h, heapitem *dest)
{
    *dest = h->d[0];
    h->items--;
    if (h->items == 0) return;
    { heapitem t = h->d[0]; h->d[0] = h->d[h->items]; h->d[h->items] = t; }
    long i = 0, smallest = 0;
    for (;;) {
        long l = i * 2 + 1, r = i * 2 + 2;
        if (l < h->items) {
            if (h_smaller(&h->d[l], &h->d[i])) smallest = l;
            if (r < h->items && h_smaller(&h->d[r], &h->d[smallest])) smallest = r;
        } else break;
        if (smallest == i) break;
        { heapitem t = h->d[i]; h->d[i] = h->d[smallest]; h->d[smallest] = t; }
        i = smallest;
    }
}

/* labels: in = markers (int32, 0 = unlabelled), out = watershed labels.  Works on a 1-pixel padded copy. */
ORC_API int orc_watershed_f64(const double *img, int32_t *labels, long ny, long nx, int wsl)
{
    long py = ny + 2, px = nx + 2, pn = py * px;
    double *pimg = (double *)calloc((size_t)pn, sizeof(double));
    int32_t *pout = (int32_t *)calloc((size_t)pn, sizeof(int32_t));
    uint8_t *mask = (uint8_t *)calloc((size_t)pn, 1);
    heap_t hp;
    hp.space = 1024 > ny * nx ? 1024 : ny * nx;
    hp.items = 0;
    hp.d = (heapitem *)malloc(sizeof(heapitem) * (size_t)hp.space);
    if (!pimg || !pout || !mask || !hp.d) return -1;
    for (long y = 0; y < ny; y++)
        for (long x = 0; x < nx; x++) {
            long p = (y + 1) * px + x + 1;
            pimg[p] = img[y * nx + x];
            pout[p] = labels[y * nx + x];
            mask[p] = 1;
        }
    const long nb[4] = { -px, -1, 1, px };
    int64_t age = 1;
    heapitem e, ne;
    for (long p = 0; p < pn; p++)
        if (pout[p]) {
            e.value = pimg[p]; e.age = 0; e.index = (int32_t)p; e.source = (int32_t)p;
            if (h_push(&hp, &e)) return -1;
        }
    while (hp.items > 0) {
        h_pop(&hp, &e);
        if (wsl) {
            if (pout[e.index] && e.index != e.source) continue;
            /* _diff_neighbors */
            int diff = 0;
            if (!mask[e.index]) diff = 1;
            else {
                int32_t l0 = 0, l1 = 0;
                for (int k = 0; k < 4; k++) {
                    long q = e.index + nb[k];
                    if (mask[q]) {
                        if (!l0) l0 = pout[q];
                        else { l1 = pout[q]; if (l1 && l1 != l0) { diff = 1; break; } }
                    }
                }
            }
            if (diff) { mask[e.index] = 0; continue; }
            pout[e.index] = pout[e.source];
        }
        for (int k = 0; k < 4; k++) {
            long q = e.index + nb[k];
            if (!mask[q]) continue;
            if (pout[q]) continue;
            age += 1;
            ne.value = pimg[q];
            if (!wsl) pout[q] = pout[e.index];
            ne.age = age; ne.index = (int32_t)q; ne.source = e.source;
            if (h_push(&hp, &ne)) return -1;
        }
    }
    for (long y = 0; y < ny; y++)
        for (long x = 0; x < nx; x++) labels[y * nx + x] = pout[(y + 1) * px + x + 1];
    free(pimg); free(pout); free(mask); free(hp.d);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Pop order of M equal-keyed marker entries (key (v, age 0), pushed in raster  */
/* order) when popping marker i is followed by c[i] pushes of larger, unique    */
/* entries -- the marker phase of the flood above on a two-valued image         */
/* (pl.py:194), replayed on the same literal heap.  order[t] = marker popped    */
/* t-th.  Checks the closed-form evaluation in the product's                    */
/* csrc/tip_heaporder.hip.                                                      */
/* ------------------------------------------------------------------------- */
ORC_API int orc_equal_key_pop_order(const uint8_t *c, long m, int32_t *order)
{
    heap_t hp;
    hp.space = m > 1024 ? m : 1024;
    hp.items = 0;
    hp.d = (heapitem *)malloc(sizeof(heapitem) * (size_t)hp.space);
    if (!hp.d) return -1;
    heapitem e, ne;
    int64_t age = 0;
    long t = 0;
    for (long i = 0; i < m; i++) {
        e.value = 0.0; e.age = 0; e.index = (int32_t)i; e.source = (int32_t)i;
        if (h_push(&hp, &e)) return -1;
    }
    while (hp.items > 0) {
        h_pop(&hp, &e);
        if (e.value != 0.0) break;               /* first non-marker entry: the marker phase is over */
        order[t++] = e.index;
        for (int k = 0; k < c[e.index]; k++) {
            ne.value = 255.0; ne.age = ++age; ne.index = -1; ne.source = -1;
            if (h_push(&hp, &ne)) return -1;
        }
    }
    free(hp.d);
    return t == m ? 0 : -2;
}

/* ------------------------------------------------------------------------- */
/* regionprops reductions (ti.py:891): per label area, bbox (min_row,min_col,  */
/* max_row+1,max_col+1), coordinate sums (centroid = sum/area), and the        */
/* skimage.measure.perimeter(neighbourhood=4) code histogram                   */
/* (skimage/measure/_regionprops_utils.py:186-249): border pixel = label pixel */
/* with a 4-neighbour outside the region; code = 1 + 2*#4-nbr border +         */
/* 10*#diag border (of the same region); weights 1 for {5,7,15,17,25,27},      */
/* sqrt2 for {21,33}, (1+sqrt2)/2 for {13,23}.  pc[3*l+{0,1,2}] count those.   */
/* ------------------------------------------------------------------------- */
static inline int is_border(const int32_t *lab, long ny, long nx, long y, long x, int32_t l)
{
    if (y < 0 || y >= ny || x < 0 || x >= nx) return 0;
    if (lab[y * nx + x] != l) return 0;
    if (y == 0 || lab[(y - 1) * nx + x] != l) return 1;
    if (y == ny - 1 || lab[(y + 1) * nx + x] != l) return 1;
    if (x == 0 || lab[y * nx + x - 1] != l) return 1;
    if (x == nx - 1 || lab[y * nx + x + 1] != l) return 1;
    return 0;
}
ORC_API int orc_regionprops_i32(const int32_t *lab, const double *intensity, long ny, long nx, long nlab,
                                int64_t *area, int64_t *bbox, int64_t *sumy, int64_t *sumx, int64_t *pc,
                                double *isum)
{
    for (long l = 0; l < nlab; l++) {
        area[l] = 0; sumy[l] = 0; sumx[l] = 0;
        bbox[4 * l] = ny; bbox[4 * l + 1] = nx; bbox[4 * l + 2] = 0; bbox[4 * l + 3] = 0;
        pc[3 * l] = pc[3 * l + 1] = pc[3 * l + 2] = 0;
        if (isum) isum[l] = 0.0;
    }
    for (long y = 0; y < ny; y++)
        for (long x = 0; x < nx; x++) {
            int32_t l = lab[y * nx + x];
            if (l <= 0 || l > nlab) continue;
            long k = l - 1;
            area[k]++; sumy[k] += y; sumx[k] += x;
            if (isum) isum[k] += intensity[y * nx + x];
            if (y < bbox[4 * k]) bbox[4 * k] = y;
            if (x < bbox[4 * k + 1]) bbox[4 * k + 1] = x;
            if (y + 1 > bbox[4 * k + 2]) bbox[4 * k + 2] = y + 1;
            if (x + 1 > bbox[4 * k + 3]) bbox[4 * k + 3] = x + 1;
            if (is_border(lab, ny, nx, y, x, l)) {
                int code = 1;
                code += 2 * (is_border(lab, ny, nx, y - 1, x, l) + is_border(lab, ny, nx, y + 1, x, l) +
                             is_border(lab, ny, nx, y, x - 1, l) + is_border(lab, ny, nx, y, x + 1, l));
                code += 10 * (is_border(lab, ny, nx, y - 1, x - 1, l) + is_border(lab, ny, nx, y - 1, x + 1, l) +
                              is_border(lab, ny, nx, y + 1, x - 1, l) + is_border(lab, ny, nx, y + 1, x + 1, l));
                if (code == 5 || code == 7 || code == 15 || code == 17 || code == 25 || code == 27) pc[3 * k]++;
                else if (code == 21 || code == 33) pc[3 * k + 1]++;
                else if (code == 13 || code == 23) pc[3 * k + 2]++;
            }
        }
    return 0;
}

/* u16 histogram helper for the exact percentile of integer-valued float data (sp.py:33-36). */
ORC_API int orc_hist_u16(const uint16_t *in, long n, int sub, int64_t *hist /* 65536 */)
{
    memset(hist, 0, sizeof(int64_t) * 65536);
    for (long i = 0; i < n; i++) {
        int v = in[i];
        if (sub) { v -= sub; if (v < 0) v = 0; }
        hist[v]++;
    }
    return 0;
}

/* ---- build_continues_manifold (sp.py:87-165), restated: a square spiral around the first global maximum of score;
 * every visited pixel takes its plane from the planes of its visited neighbours (find_pixel_plane, sp.py:131-165).
 * score: float32 (Z, R, C); out: int64 (R, C).  Returns 0, or -1 when a pixel has no visited neighbour (upstream raises). */
static long man_plane(const float *score, const long long *out, long r, long c, long Z, long R, long C, int *bad)
{
    const long P = R * C;
    long nb[4], n1 = -1, n2 = -1, j;
    nb[0] = (long)out[(r > 0 ? r - 1 : R - 1) * C + c];              /* chosen_z[row - 1]: row 0 wraps to the last row */
    nb[1] = r < R - 1 ? (long)out[(r + 1) * C + c] : -1;
    nb[2] = c > 0 ? (long)out[r * C + c - 1] : -1;
    nb[3] = c < C - 1 ? (long)out[r * C + c + 1] : -1;
    for (j = 0; j < 4; ++j) {                                        /* the first two visited neighbours, in that order */
        if (nb[j] < 0) continue;
        if (n1 < 0) n1 = nb[j];
        else if (n2 < 0) n2 = nb[j];
    }
    if (n1 < 0) { *bad = 1; return 0; }
    {
        long lo, hi, z, best;
        if (n2 < 0 || n1 == n2) { lo = n1 - 1 < 0 ? 0 : n1 - 1; hi = n1 + 2 > Z ? Z : n1 + 2; }
        else if (n1 - n2 == 1 || n2 - n1 == 1) { lo = n1 < n2 ? n1 : n2; hi = lo + 2 > Z ? Z : lo + 2; }
        else return (n1 + n2) / 2;                                   /* float mean stored into an integer array */
        best = lo;
        for (z = lo + 1; z < hi; ++z)
            if (score[z * P + r * C + c] > score[best * P + r * C + c]) best = z;
        return best;
    }
}

ORC_API long orc_build_manifold_f32(const float *score, long Z, long R, long C, long long *out)
{
    const long P = R * C;
    long i, best = 0, sp, sr, sc, d, dmax, row, col;
    int bad = 0;
    for (i = 0; i < P; ++i) out[i] = -1;
    for (i = 1; i < Z * P; ++i)
        if (score[i] > score[best]) best = i;
    sp = best / P; sr = (best % P) / C; sc = best % C;
    out[sr * C + sc] = sp;
    dmax = sc;
    if (sr > dmax) dmax = sr;
    if (C - 1 - sc > dmax) dmax = C - 1 - sc;
    if (R - 1 - sr > dmax) dmax = R - 1 - sr;
#define MAN_SET(rr, cc) out[(rr) * C + (cc)] = man_plane(score, out, (rr), (cc), Z, R, C, &bad)
    for (d = 1; d <= dmax; ++d) {
        col = sc + d;                                                /* right edge, lower half */
        if (col < C)
            for (row = sr; row <= sr + d; ++row)
                if (row < R) MAN_SET(row, col);
        row = sr + d;                                                /* bottom edge, leftwards */
        if (row < R)
            for (col = sc + d - 1; col >= sc - d; --col)
                if (col >= 0 && col < C) MAN_SET(row, col);
        col = sc - d;                                                /* left edge, upwards */
        if (col >= 0)
            for (row = sr + d - 1; row >= sr - d; --row)
                if (row >= 0 && row < R) MAN_SET(row, col);
        row = sr - d;                                                /* top edge, rightwards */
        if (row >= 0)
            for (col = sc - d + 1; col <= sc + d; ++col)
                if (col >= 0 && col < C) MAN_SET(row, col);
        col = sc + d;                                                /* right edge, upper half */
        if (col < C)
            for (row = sr - d + 1; row < sr; ++row)
                if (row >= 0) MAN_SET(row, col);
    }
#undef MAN_SET
    return bad ? -1 : 0;
}
