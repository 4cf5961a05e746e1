"""CPU: the closed-form evaluation of skimage's equal-key heap pop order (csrc/tip_heaporder.hip, a host stage of the
two-valued watershed, pl.py:194) against a literal replay of the heap (oracle).  No GPU involved: host arrays only."""
import ctypes

import numpy as np
import pytest


def product_order(c):
    from tissue_image_processing_amd import _lib
    lib = _lib.load()
    c = np.ascontiguousarray(c, np.uint8)
    e = np.empty(c.size, np.uint32)
    rc = lib.tip_marker_pop_order_host(c.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(c.size),
                                       e.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    order = np.empty(c.size, np.int64)
    order[e] = np.arange(c.size)
    return order


def test_small_cases_by_hand():
    from oracle import oracle as orc
    # no pushes at all: first entry, then the array end backwards
    np.testing.assert_array_equal(orc.equal_key_pop_order(np.zeros(6, np.uint8)), [0, 5, 4, 3, 2, 1])
    for c in ([0], [3], [0, 0], [4, 4], [1, 0, 0], [0, 1, 0, 2, 0, 0, 1, 0]):
        np.testing.assert_array_equal(product_order(c), orc.equal_key_pop_order(np.array(c, np.uint8)))


@pytest.mark.parametrize("seed", range(4))
def test_random_counts_match_literal_heap(seed):
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    for _ in range(400):
        m = int(rng.integers(1, 600))
        p = rng.uniform(0, 0.8)
        c = ((rng.uniform(size=m) < p) * rng.integers(1, 5, size=m)).astype(np.uint8)
        np.testing.assert_array_equal(product_order(c), orc.equal_key_pop_order(c))


def test_image_like_counts_large():
    """Counts as a boundary image produces them (runs of interior markers, bursts at cell borders), 1.5 M markers."""
    from oracle import oracle as orc
    rng = np.random.default_rng(7)
    m = 1_500_000
    c = np.zeros(m, np.uint8)
    edges = rng.uniform(size=m) < 0.12
    c[edges] = rng.integers(1, 4, size=int(edges.sum()))
    c[::2048] = 1
    np.testing.assert_array_equal(product_order(c), orc.equal_key_pop_order(c))
