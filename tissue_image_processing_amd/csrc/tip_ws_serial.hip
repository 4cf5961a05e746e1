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
//      Cost: O(n log n) on one host core, ~0.1 us per pixel for integer-valued frames (see "the literal array heap" below;
//      the reference's own Cython loop is ~0.3 us); `flags` bit 2 tells the caller it ran.
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

// ---- the literal array heap ------------------------------------------------------------------------------------------------
// Bit parity needs upstream's heap ARRAY at every moment (header above), not its speed.  What is free: how an entry is stored and how
// a sift is carried out, as long as every element ends in the slot upstream's code would put it.
//   * pop: upstream moves the last element x to the root and sinks it towards the smaller child (the left one among equals) while that
//     child is smaller than x.  The path of smaller children does not depend on x, so the hole is walked down that path to a leaf
//     without looking at x (one comparison and one move per level, no unpredictable exit branch), and x then climbs back while its
//     parent on the path is NOT smaller than x -- it stops under the last element smaller than it, which is the slot the top-down
//     sink ends in: above every path element >= x (equal keys exist only between seeds, and upstream leaves x above them too).
//     x comes from the bottom row, so the climb is a step or two.
//   * push: the entry climbs while strictly smaller than its parent (upstream's rule), moved with a hole instead of swaps.
//   * the children two levels down are prefetched while a level is decided; the cells of the next few pops are prefetched from the
//     heap's top right after a pop (the pops of one level are scattered over the whole frame: three cache misses per pop otherwise).
// Keys: integer-valued landscapes below 2^31 (the uint16 frames this stage exists for) order by ONE 64-bit word,
// value << 33 | age; anything else by (sortable image of the double, age).  Measured on a 2048^2 uint16 landscape: 3.3 x faster than
// the swap-based heap with 24-byte comparisons it replaces (same output, tests/test_ws_serial_host.py).
constexpr int AGE_BITS = 33;                      // age < 2^33: four pushes per pixel of a < 2^31 pixel image
constexpr uint64_t AGE_MASK = (1ULL << AGE_BITS) - 1;

struct KeyInt {                                   // value << 33 | age
    uint64_t k;
    static KeyInt make(uint32_t value, uint64_t age) { return KeyInt{((uint64_t)value << AGE_BITS) | age}; }
    bool smaller(const KeyInt &o) const { return k < o.k; }
    bool seed() const { return (k & AGE_MASK) == 0; }
};
struct KeyF64 {                                   // (order_key(value), age)
    uint64_t v, age;
    static KeyF64 make(uint64_t value, uint64_t age) { return KeyF64{value, age}; }
    bool smaller(const KeyF64 &o) const { return v < o.v || (v == o.v && age < o.age); }   // entries that tie in value AND age are "not smaller" either way
    bool seed() const { return age == 0; }
};

template <typename Key> struct HeapEntry { Key key; uint32_t pixel; };

template <typename Key> struct ArrayHeap {
    typedef HeapEntry<Key> E;
    std::vector<E> store;
    E *d = nullptr;
    size_t n = 0;
    explicit ArrayHeap(size_t cap) { store.resize(cap + 8); d = store.data(); }
    void push(const E &e)
    {
        if (n + 8 > store.size()) { store.resize(store.size() * 2); d = store.data(); }
        size_t c = n++;
        while (c > 0) {
            const size_t parent = (c - 1) >> 1;
            if (!e.key.smaller(d[parent].key)) break;
            d[c] = d[parent];
            c = parent;
        }
        d[c] = e;
    }
    E pop()
    {
        const E top = d[0];
        const size_t m = --n;
        if (m == 0) return top;
        const E x = d[m];
        size_t i = 0;
        for (;;) {
            const size_t l = 2 * i + 1;
            if (l >= m) break;
            __builtin_prefetch(&d[8 * i + 7]);
            __builtin_prefetch(&d[8 * i + 11]);
            const size_t r = l + 1 < m ? l + 1 : l;
            const size_t s = d[r].key.smaller(d[l].key) ? r : l;
            d[i] = d[s];
            i = s;
        }
        while (i > 0) {
            const size_t parent = (i - 1) >> 1;
            if (d[parent].key.smaller(x.key)) break;
            d[i] = d[parent];
            i = parent;
        }
        d[i] = x;
        return top;
    }
};

// one pixel of the padded frame: lab -1 = closed (outside the image, or a line: upstream's mask), 0 = undecided, > 0 = label
template <typename V> struct FloodCell { int32_t lab; V value; };

template <typename Key, typename V, typename ValueOf>
int flood_exact_t(const double *img, const int32_t *markers, int32_t *labels, int Y, int X, ValueOf value_of)
{
    typedef FloodCell<V> Cell;
    typedef HeapEntry<Key> E;
    const long PX = (long)X + 2, PN = ((long)Y + 2) * PX;
    std::vector<Cell> cells((size_t)PN + 1, Cell{-1, 0});
    Cell *c = cells.data();
    long nmark = 0;
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            const long p = (long)(y + 1) * PX + x + 1, i = (long)y * X + x;
            c[p].lab = markers[i] > 0 ? markers[i] : 0;
            c[p].value = value_of(img[i]);
            nmark += markers[i] > 0;
        }
    ArrayHeap<Key> hp((size_t)std::max<long>(1024, nmark + (long)Y * X / 2));
    for (long p = 0; p < PN; ++p)
        if (c[p].lab > 0) hp.push(E{Key::make(c[p].value, 0), (uint32_t)p});     // age 0, raster order
    const long nb[4] = {-PX, -1, 1, PX};     // upstream's push order: up, left, right, down
    uint64_t age = 1;
    while (hp.n) {
        const E e = hp.pop();
        if (hp.n > 2)
            for (int t = 0; t < 3; ++t) {
                const Cell *z = c + hp.d[t].pixel;
                __builtin_prefetch(z); __builtin_prefetch(z - PX); __builtin_prefetch(z + PX);
            }
        const long p = (long)e.pixel;
        const bool seed = e.key.seed();
        const int32_t own = c[p].lab;
        if (!seed && own != 0) continue;                     // reached earlier through another neighbour, or a line by now
        const Cell q[4] = {c[p + nb[0]], c[p + nb[1]], c[p + nb[2]], c[p + nb[3]]};
        // upstream's _diff_neighbors: a pixel that is no longer open, or whose open neighbours carry two labels, is a line
        bool line = own < 0;
        int32_t l0 = 0;
        if (!line)
            for (int k = 0; k < 4; ++k) {
                const int32_t l = q[k].lab;
                if (l < 0) continue;
                if (l0 == 0) l0 = l;
                else if (l != 0 && l != l0) { line = true; break; }
            }
        if (line) { c[p].lab = -1; continue; }
        // the entry that pops first for a pixel was pushed by its earliest-labelled neighbour, and with no second label
        // around, every labelled neighbour carries that neighbour's label: upstream's output[source]
        if (!seed) c[p].lab = l0;
        for (int k = 0; k < 4; ++k) {
            if (q[k].lab != 0) continue;
            ++age;
            hp.push(E{Key::make(q[k].value, age), (uint32_t)(p + nb[k])});
        }
    }
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            const int32_t l = c[(long)(y + 1) * PX + x + 1].lab;
            labels[(long)y * X + x] = l > 0 ? l : 0;
        }
    return TIP_OK;
}

}  // namespace

// img, markers, labels: Y x X row-major.  markers > 0 are the seeds; labels receives the flood (0 = line / unreached).
int flood_exact(const double *img, const int32_t *markers, int32_t *labels, int Y, int X)
{
    const long PX = (long)X + 2, PN = ((long)Y + 2) * PX, n = (long)Y * X;
    if (PN >= (1L << 31)) return fail(TIP_ERR_ARG, "watershed: image too large for the serial stage");
    bool small_ints = true;                       // every value a whole number in [0, 2^31): one-word keys
    for (long i = 0; i < n && small_ints; ++i) {
        const double v = img[i];
        small_ints = v >= 0.0 && v < 2147483648.0 && v == (double)(uint32_t)v;
    }
    if (small_ints)
        return flood_exact_t<KeyInt, uint32_t>(img, markers, labels, Y, X, [](double v) { return (uint32_t)v; });
    return flood_exact_t<KeyF64, uint64_t>(img, markers, labels, Y, X, [](double v) { return order_key(v); });
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
