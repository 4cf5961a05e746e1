"""CPU: the product's own serial (value, age) flood (csrc/tip_ws_serial.hip, host code of libtissue_hip.so) against the
reference's goldens -- skimage.segmentation.watershed(watershed_line=True) run on tie-free, quantised (value ties) and
two-valued images (tests/golden/watershed.npz, unet_tail.npz, made by tools/make_goldens.py from the reference's calls at
bim.py:475 / pl.py:194).  No device is involved: the markers come from the oracle's local-minima labelling here, from the
device's in the product (tests/test_gpu_segmentation.py checks the two together)."""
import ctypes
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tissue_image_processing_amd import _lib

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def serial_flood(img, markers):
    lib = _lib.load()
    img = np.ascontiguousarray(img, np.float64)
    markers = np.ascontiguousarray(markers, np.int32)
    out = np.empty(img.shape, np.int32)
    rc = lib.tip_watershed_serial_host(_lib.ptr(img), _lib.ptr(markers), _lib.ptr(out), img.shape[0], img.shape[1])
    assert rc == 0
    return out


def markers_of(img):
    return orc.label4(orc.local_minima(np.ascontiguousarray(img, np.float64)).astype(np.int32), 0)[0].astype(np.int32)


@pytest.mark.parametrize("case", ["ii", "iii", "iv", "v"])
def test_serial_flood_equals_reference_goldens(case):
    g = np.load(os.path.join(G, "watershed.npz"))
    img = g[case + "_img"]
    out = serial_flood(img, markers_of(img))
    assert np.array_equal(out, g[case + "_labels"]), case


def test_serial_flood_on_non_adjacent_ties():
    """images whose only value ties are between non-marker pixels that share a neighbour (diagonals, distance two): skimage's
    push-age order decides them (tools/make_goldens_ties.py keeps only cases where a raster order gets it wrong)"""
    g = np.load(os.path.join(G, "watershed_diag_ties.npz"))
    for k in range(6):
        img = g["img%d" % k]
        assert np.array_equal(serial_flood(img, markers_of(img)), g["labels%d" % k]), k
        assert np.array_equal(orc.watershed(img), g["labels%d" % k]), k


def test_serial_flood_with_the_reference_markers():
    g = np.load(os.path.join(G, "watershed.npz"))
    out = serial_flood(g["i2_blurred"], g["i2_markers"])
    assert np.array_equal(out, g["i2_labels"])


def test_serial_flood_two_valued_goldens():
    g = np.load(os.path.join(G, "watershed.npz"))
    out = serial_flood(g["vi_boundary"], markers_of(g["vi_boundary"]))
    assert np.array_equal(out, g["vi_labels"])
    t = np.load(os.path.join(G, "unet_tail.npz"))
    out = serial_flood(t["boundary"], markers_of(t["boundary"]))
    assert np.array_equal(out, t["labels"])


def test_serial_flood_equals_oracle_on_quantised_landscapes():
    rng = np.random.default_rng(5)
    for shape, levels in (((70, 90), 7), ((1, 40), 3), ((33, 1), 4), ((64, 64), 40)):
        base = rng.random(shape)
        from oracle.oracle import gaussian_filter
        img = np.round(gaussian_filter(base, 2.0) * levels * 4).astype(np.float64)
        out = serial_flood(img, markers_of(img))
        assert np.array_equal(out, orc.watershed(img)), (shape, levels)
    flat = np.zeros((9, 11))
    assert np.array_equal(serial_flood(flat, markers_of(flat)), orc.watershed(flat))
