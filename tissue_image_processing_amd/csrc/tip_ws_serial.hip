// tip_ws_serial.hip -- the two SERIAL host stages of the watershed (product code; nothing here touches the device).
//
// skimage's flood (skimage/segmentation/_watershed_cy.pyx + heap_general.pxi; reference call sites bim.py:475 and
// pl.py:194) is a serial priority flood whose result depends on the pop order of entries with EQUAL keys.  Two cases
// cannot be decomposed into independent device work:
//
//  (1) flood_exact(): landscapes whose pixels tie in value without being two-valued -- the integer frames the GUI hands
//      to watershed_segmentation (gui.py:1841-1845, bim.py:473-475).  Equal-valued non-markers pop in push order (age),
//      and equal-valued MARKERS (all pushed with age 0) pop in an order that is a function of where the whole history of
//      sift-ups and sift-downs has left them in the heap ARRAY.  Bit parity therefore needs the heap
//      itself: this routine runs the flood on a literal array heap with upstream's comparison (value, then age, strict),
//      upstream's sift rules (a pushed entry climbs while strictly smaller than its parent; a pop moves the last array
//      element to the root and sinks it towards the smaller child, the left one among equals) and upstream's push order
//      (up, left, right, down; age = running push count).  The device still does everything around it: local minima,
//      their connected-component labels in raster order, threshold and blur before, cell tables after.
//      Cost: O(n log n) on one host core, ~0.3 us per pixel (about 1 s for a 2048^2 frame) -- the same class as the
//      reference's own Cython loop; `flags` bit 2 tells the caller it ran.
//
//  (2) flood_keyed_finish(): mode A (tip_watershed.hip) orders pixels by the static key (value, raster index).  When its
//      tile rounds, pocket certificates, per-component endgame and wide pass all stall (plateaus of equal value larger
//      than any certificate can close), what is left is one long dependency chain, and a chain is serial work: this
//      routine finishes the flood from the device's partial state with the same pop-time rule the device kernels
//      evaluate (k_end_resolve / the old one-pixel-per-round-trip global-minimum step), in one pass over a heap of
//      candidate pop times.  `flags` bit 3 + the number of pixels it committed (bits 8..) report it.
#include "tip_internal.h"
#include <algorithm>

namespace tip {

namespace {

// sortable image of a double: a < b  <=>  key(a) < key(b), and -0.0 == +0.0 (the + 0.0 folds the zeros together)
inline uint64_t order_key(double d)
{
    d += 0.0;
    uint64_t b;
    memcpy(&b, &d, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}

struct Entry {          // 16 bytes: three entries per 48 bytes of cache instead of two per 48 with upstream's 24-byte struct
    uint64_t v;         // order_key(value)
    uint64_t ai;        // age << 31 | padded pixel index   (age < 2^33: 4 pushes per pixel of a < 2^31 pixel image)
};
constexpr uint64_t IDX_MASK = (1ULL << 31) - 1;
inline bool smaller(const Entry &a, const Entry &b)
{
    if (a.v != b.v) return a.v < b.v;
    return (a.ai >> 31) < (b.ai >> 31);      // age only: entries that tie in value AND age are "not smaller" either way
}

struct ArrayHeap {
    std::vector<Entry> d;
    void push(const Entry &e)
    {
        size_t child = d.size();
        d.push_back(e);
        while (child > 0) {
            const size_t parent = (child + 1) / 2 - 1;
            if (!smaller(d[child], d[parent])) break;
            std::swap(d[child], d[parent]);
            child = parent;
        }
    }
    Entry pop()
    {
        const Entry top = d[0];
        const size_t n = d.size() - 1;
        if (n == 0) { d.pop_back(); return top; }
        d[0] = d[n];
        d.pop_back();
        size_t i = 0;
        for (;;) {
            const size_t l = 2 * i + 1, r = l + 1;
            if (l >= n) break;
            size_t s = i;
            if (smaller(d[l], d[i])) s = l;
            if (r < n && smaller(d[r], d[s])) s = r;
            if (s == i) break;
            std::swap(d[i], d[s]);
            i = s;
        }
        return top;
    }
};

}  // namespace

// img, markers, labels: Y x X row-major.  markers > 0 are the seeds; labels receives the flood (0 = line / unreached).
int flood_exact(const double *img, const int32_t *markers, int32_t *labels, int Y, int X)
{
    const long PX = (long)X + 2, PN = ((long)Y + 2) * PX;
    if (PN > (long)IDX_MASK) return fail(TIP_ERR_ARG, "watershed: image too large for the serial stage");
    std::vector<int32_t> out((size_t)PN, 0);
    std::vector<uint8_t> open((size_t)PN, 0);     // upstream's mask: 1 inside the image until the pixel becomes a line
    std::vector<uint64_t> key((size_t)PN, 0);
    ArrayHeap hp;
    long nmark = 0;
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            const long p = (long)(y + 1) * PX + x + 1;
            const long i = (long)y * X + x;
            out[(size_t)p] = markers[i];
            open[(size_t)p] = 1;
            key[(size_t)p] = order_key(img[i]);
            nmark += markers[i] > 0;
        }
    hp.d.reserve((size_t)std::max<long>(1024, nmark + (long)Y * X / 2));
    for (long p = 0; p < PN; ++p)
        if (out[(size_t)p] > 0) hp.push(Entry{key[(size_t)p], (uint64_t)p});     // age 0, raster order
    const long nb[4] = {-PX, -1, 1, PX};
    uint64_t age = 1;
    while (!hp.d.empty()) {
        const Entry e = hp.pop();
        const long p = (long)(e.ai & IDX_MASK);
        const bool seed = (e.ai >> 31) == 0;
        if (!seed && out[(size_t)p] != 0) continue;          // reached earlier through another neighbour
        // upstream's _diff_neighbors: a pixel that is no longer open, or whose open neighbours carry two labels, is a line
        bool line = !open[(size_t)p];
        int32_t l0 = 0;
        if (!line)
            for (int k = 0; k < 4; ++k) {
                const long q = p + nb[k];
                if (!open[(size_t)q]) continue;
                const int32_t l = out[(size_t)q];
                if (l0 == 0) l0 = l;
                else if (l != 0 && l != l0) { line = true; break; }
            }
        if (line) { open[(size_t)p] = 0; continue; }
        // the entry that pops first for a pixel was pushed by its earliest-labelled neighbour, and with no second label
        // around, every labelled neighbour carries that neighbour's label: upstream's output[source]
        if (!seed) out[(size_t)p] = l0;
        for (int k = 0; k < 4; ++k) {
            const long q = p + nb[k];
            if (!open[(size_t)q] || out[(size_t)q] != 0) continue;
            ++age;
            hp.push(Entry{key[(size_t)q], (age << 31) | (uint64_t)q});
        }
    }
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) labels[(long)y * X + x] = out[(size_t)((long)(y + 1) * PX + x + 1)];
    return TIP_OK;
}

// ---- mode A finisher ------------------------------------------------------------------------------------------------------
// st: the device's packed state per pixel (low 32 bits: label > 0, 0 undecided, -1 line; high 32: pop-time reference pixel).
// A labelled pixel's pop time is (img[ref], ref).  An undecided pixel with a labelled neighbour pops at
//     max( (img[p], p) , earliest pop time among its labelled neighbours )
// and on popping takes the label of the neighbours labelled before its own key (a line when they disagree) or, when there
// are none, the label and pop time of the neighbour that pulled it.  Pop times only grow, so a pixel's candidate is
// fixed when it gets its first labelled neighbour: one heap entry per pixel.  Returns the number of pixels decided.
long flood_keyed_finish(const double *img, uint64_t *st, int Y, int X)
{
    struct Time { uint64_t v; uint32_t ref; };
    auto lab_of = [](uint64_t s) { return (int32_t)(uint32_t)(s & 0xffffffffULL); };
    auto ref_of = [](uint64_t s) { return (uint32_t)(s >> 32); };
    auto before = [](const Time &a, const Time &b) { return a.v < b.v || (a.v == b.v && a.ref < b.ref); };
    struct Cand { uint64_t v; uint32_t ref; uint32_t p; };
    struct Later { bool operator()(const Cand &a, const Cand &b) const {
        if (a.v != b.v) return a.v > b.v;
        if (a.ref != b.ref) return a.ref > b.ref;
        return a.p > b.p;
    } };
    std::vector<Cand> heap;
    const long n = (long)Y * X;
    auto neighbours = [&](long p, long *q) {
        const int y = (int)(p / X), x = (int)(p - (long)y * X);
        q[0] = y > 0 ? p - X : -1; q[1] = x > 0 ? p - 1 : -1; q[2] = x < X - 1 ? p + 1 : -1; q[3] = y < Y - 1 ? p + X : -1;
    };
    auto time_of = [&](uint64_t s) { const uint32_t r = ref_of(s); return Time{order_key(img[r]), r}; };
    // candidate of undecided pixel p from its labelled neighbours (false: it has none)
    auto candidate = [&](long p, Cand &c) {
        long q[4];
        neighbours(p, q);
        bool has = false;
        Time best{0, 0};
        for (int k = 0; k < 4; ++k) {
            if (q[k] < 0 || lab_of(st[q[k]]) <= 0) continue;
            const Time t = time_of(st[q[k]]);
            if (!has || before(t, best)) { best = t; has = true; }
        }
        if (!has) return false;
        const Time own{order_key(img[p]), (uint32_t)p};
        const Time t = before(own, best) ? best : own;
        c = Cand{t.v, t.ref, (uint32_t)p};
        return true;
    };
    for (long p = 0; p < n; ++p) {
        Cand c;
        if (lab_of(st[p]) == 0 && candidate(p, c)) heap.push_back(c);
    }
    std::make_heap(heap.begin(), heap.end(), Later());
    long decided = 0;
    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), Later());
        const Cand c = heap.back();
        heap.pop_back();
        const long p = c.p;
        if (lab_of(st[p]) != 0) continue;
        const Time own{order_key(img[p]), (uint32_t)p};
        long q[4];
        neighbours(p, q);
        int32_t first = 0, pull_lab = 0;
        uint32_t pull_ref = 0;
        bool conflict = false, has_pull = false;
        Time pull{0, 0};
        for (int k = 0; k < 4; ++k) {
            if (q[k] < 0) continue;
            const int32_t l = lab_of(st[q[k]]);
            if (l <= 0) continue;
            const Time t = time_of(st[q[k]]);
            if (before(t, own)) {
                if (first == 0) first = l;
                else if (first != l) conflict = true;
            } else if (!has_pull || before(t, pull)) {
                has_pull = true; pull = t; pull_lab = l; pull_ref = ref_of(st[q[k]]);
            }
        }
        int32_t lab;
        uint32_t ref;
        if (first != 0) { lab = conflict ? -1 : first; ref = (uint32_t)p; }
        else if (has_pull) { lab = pull_lab; ref = pull_ref; }
        else continue;
        st[p] = ((uint64_t)ref << 32) | (uint32_t)lab;
        ++decided;
        if (lab <= 0) continue;
        for (int k = 0; k < 4; ++k) {
            if (q[k] < 0 || lab_of(st[q[k]]) != 0) continue;
            // (a neighbour that already had a labelled neighbour owns an entry that is not later: the duplicate is skipped
            // when it pops)
            Cand nc;
            if (candidate(q[k], nc)) { heap.push_back(nc); std::push_heap(heap.begin(), heap.end(), Later()); }
        }
    }
    return decided;
}

}  // namespace tip

extern "C" {

// Host arrays in, host array out: the serial (value, age) flood itself, exposed for callers that already hold markers
// (and for the tests, which compare it with the reference's goldens without a device).
int tip_watershed_serial_host(const double *img, const int32_t *markers, int32_t *labels, int y, int x)
{
    if (!img || !markers || !labels || y < 1 || x < 1) return TIP_ERR_ARG;
    return tip::flood_exact(img, markers, labels, y, x);
}

}  // extern "C"
