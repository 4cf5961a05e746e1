// tip_heaporder.hip -- pop order of equal-keyed marker entries in skimage's binary heap (host stage of watershed mode B).
//
// skimage's flood (skimage/segmentation/_watershed_cy.pyx + heap_general.pxi; the tests replay it literally on the
// CPU) pushes every marker pixel with key (value, age 0) in raster order and pops by (value, age).  On
// the two-valued image of pl.py:194 every marker has the SAME key, so the order in which the markers pop -- and with it
// the age of every pixel they push, i.e. who wins each contested boundary pixel -- is decided by the mechanics of the
// array heap alone (strict `<` in sift-up and sift-down, left child preferred among equals).  Those mechanics have a
// closed structure, which this file evaluates instead of replaying the heap:
//
//   * all M markers sit in array positions 0..M-1 in raster order (equal keys never swap on push); every entry pushed
//     later (a non-marker pixel) is larger than every marker and unique, so it can only sink.  The markers therefore
//     always occupy a connected "crown" of the implicit binary tree that contains the root, and the crown only shrinks.
//   * a pop returns the root and moves the LAST array element to the root:
//       - if that element is a marker (its position is still in the crown) it simply becomes the next root ("back" step:
//         the array end sweeps down through the marker positions, raster-last pixels first);
//       - if it is a pushed (larger) entry it sinks along the path "left child if it is a marker, else right child if it
//         is a marker", and every marker on that path moves up one level ("front" step).  Successive front steps emit
//         the crown's markers in PRE-ORDER of the implicit tree and vacate crown positions in POST-ORDER.
//   * after popping marker i, c[i] entries are appended (c = its number of non-marker 4-neighbours inside the image), so
//     which kind of step comes next depends only on the array length and on which positions have been vacated.
//
// That is a two-stream merge (pre-order stream vs the array end) steered by the counts c, O(1) per marker with two
// bitsets, instead of O(log M) sift work on 24-byte heap items; it is still a sequential recurrence over the markers
// (the step kind depends on every earlier count), so it runs on the host between two device stages of mode B -- DESIGN.md
// 5.5 has the derivation, the validation against the literal heap, and what a device version would need.
#include "tip_internal.h"

namespace tip {

namespace {

struct Bits {
    std::vector<uint64_t> w;
    explicit Bits(long n, bool one) : w((size_t)((n + 63) >> 6), one ? ~0ULL : 0ULL) {}
    bool get(long i) const { return (w[(size_t)(i >> 6)] >> (i & 63)) & 1ULL; }
    void set(long i) { w[(size_t)(i >> 6)] |= 1ULL << (i & 63); }
    void clr(long i) { w[(size_t)(i >> 6)] &= ~(1ULL << (i & 63)); }
};

}  // namespace

// c[i]: entries pushed when marker i (raster rank i among the markers) pops.  order[t] = the marker popped t-th (written
// front to back: the inverse permutation, which the flood needs, is a scatter the device does better than the host).
int marker_pop_order(const uint8_t *c, long M, uint32_t *order)
{
    if (M <= 0) return TIP_OK;
    Bits in_crown(M, true), taken(M, false);   // position still holds a marker / marker already emitted or sitting at the root
    // pre-order successor of position p in the implicit tree over positions 0..M-1 (-1 after the last)
    auto pre_next = [M](long p) -> long {
        if (2 * p + 1 < M) return 2 * p + 1;
        while (p > 0 && ((p & 1) == 0 || p + 1 >= M)) p = (p - 1) / 2;
        return p == 0 ? -1 : p + 1;
    };
    auto descend = [M](long q) -> long {
        while (2 * q + 1 < M) q = 2 * q + 1;
        return q;
    };
    // post-order successor
    auto post_next = [M, &descend](long p) -> long {
        if (p == 0) return -1;
        if ((p & 1) == 1 && p + 1 < M) return descend(p + 1);
        return (p - 1) / 2;
    };
    long fq = pre_next(0);        // next candidate of the pre-order (front) stream
    long pp = descend(0);         // next candidate position of the post-order (vacate) stream
    long n = M;                   // array length
    long root = 0;                // marker sitting at the root
    long last_front = -1;         // position vacated by the latest front step
    uint32_t t = 0;
    taken.set(0);
    for (;;) {
        order[t++] = (uint32_t)root;
        const long last = n - 1;
        n -= 1;
        if (n == 0) break;
        const int pushes = c[root];
        if (last >= M || !in_crown.get(last)) {
            // front step: the array's last element is a pushed entry; it sinks from the root, the crown shifts up
            while (fq >= 0 && taken.get(fq)) fq = pre_next(fq);
            if (fq < 0) break;    // no marker left
            while (pp >= 0 && !in_crown.get(pp)) pp = post_next(pp);
            in_crown.clr(pp);
            last_front = pp;
            pp = post_next(pp);
            root = fq;
        } else {
            // back step: the marker at the array end becomes the root.  It is the marker that started there unless the
            // front stream is inside this position's subtree at this very moment (then markers have been shifting up
            // through it: the crown, read in pre-order, is the untaken part of the pre-order stream)
            long tok = last;
            bool meet = false;
            if (last_front > last) {
                long q = last_front;
                while (q > last) q = (q - 1) / 2;
                meet = q == last;
            }
            if (meet) {
                int depth = 0;
                for (long q = last; q > 0; q = (q - 1) / 2) ++depth;
                long f = fq;
                for (int k = depth - 1;; --k) {
                    while (f >= 0 && taken.get(f)) f = pre_next(f);
                    if (k <= 0 || f < 0) break;
                    f = pre_next(f);
                }
                if (f < 0) return fail(TIP_ERR_HIP, "marker_pop_order: inconsistent heap model state");
                tok = f;
            }
            in_crown.clr(last);
            root = tok;
        }
        taken.set(root);
        n += pushes;
    }
    if ((long)t != M) return fail(TIP_ERR_HIP, "marker_pop_order: emitted %ld of %ld markers", (long)t, M);
    return TIP_OK;
}

}  // namespace tip

// the recurrence on host arrays: a diagnostic entry (no device involved), which lets the CPU test suite compare it with
// a literal replay of the heap
extern "C" __attribute__((visibility("default"))) int tip_marker_pop_order_host(const uint8_t *c, long m, uint32_t *e)
{
    if (!c || !e || m < 0) return tip::fail(TIP_ERR_ARG, "tip_marker_pop_order_host: bad arguments");
    std::vector<uint32_t> order((size_t)m);
    int rc = tip::marker_pop_order(c, m, order.data());
    if (rc) return rc;
    for (long t = 0; t < m; ++t) e[order[(size_t)t]] = (uint32_t)t;
    return TIP_OK;
}
