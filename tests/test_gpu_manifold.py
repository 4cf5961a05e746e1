"""GPU: build_continues_manifold (sp.py:87-165) -- the spiral z-map as a device scan of function tables -- against goldens
from the reference (every start position class, the row-0 wrap quirk, truncated plane means) and against the oracle's
serial restatement at sizes that cross the scan's chunk boundaries, up to the headline frame."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(ROOT, "tests", "golden", "manifold.npz"))


def test_manifold_goldens(g):
    from tissue_image_processing_amd import surface_projection as sp
    names = [k[2:-6] for k in g.files if k.startswith("m_") and k.endswith("_score")]
    assert len(names) >= 9
    for n in names:
        z = sp.build_continues_manifold(g["m_%s_score" % n])
        assert z.dtype == np.int64
        np.testing.assert_array_equal(z, g["m_%s_z" % n], err_msg=n)


def test_projection_with_manifold_golden(g, monkeypatch, golden_taps):
    from tissue_image_processing_amd import surface_projection as sp
    from gpu_util import taps_patch
    taps_patch(monkeypatch, golden_taps)          # the taps of the interpreter that produced the goldens
    proj, zmap = sp.time_point_surface_projection(g["p_stack"][None], "TCZYX", 0, airyscan=False, z_map=True, build_manifold=True)
    np.testing.assert_array_equal(zmap, g["p_zmap"])
    np.testing.assert_array_equal(proj, g["p_proj"])
    proj, zmap = sp.time_point_surface_projection(g["p_stack"], "CZYX", 1, min_z=1, max_z=9, airyscan=False, z_map=True,
                                                  build_manifold=True, atoh_shift=-1)
    np.testing.assert_array_equal(zmap, g["p2_zmap"])        # (min_z is NOT added on this path, as upstream)
    np.testing.assert_array_equal(proj, g["p2_proj"])


def test_projection_with_manifold_and_bin_size_golden(monkeypatch, golden_taps):
    """build_manifold with bin_size > 1 (sp.py:56-65): goldens from the reference's own function (spiral on the binned score,
    plane maps through skimage's 2-D resize, np.round), all three scoring methods and an atoh shift; plus the oracle on a
    ragged shape."""
    from oracle import oracle as orc
    from tissue_image_processing_amd import surface_projection as sp, synthetic
    from gpu_util import taps_patch
    taps_patch(monkeypatch, golden_taps)
    gb = np.load(os.path.join(ROOT, "tests", "golden", "manifold_binned.npz"))
    for name, kw in (("avg4", dict(bin_size=4, method="max_averages")), ("std4", dict(bin_size=4, method="max_std")),
                     ("multi10", dict(bin_size=10, method="multi_channel")), ("avg5_shift", dict(bin_size=5, method="max_averages", atoh_shift=1))):
        proj, zmap = sp.time_point_surface_projection(gb["p_stack"][None], "TCZYX", 0, airyscan=False, z_map=True, build_manifold=True, **kw)
        np.testing.assert_array_equal(zmap, gb["p_%s_zmap" % name], err_msg=name)
        np.testing.assert_array_equal(proj, gb["p_%s_proj" % name], err_msg=name)
    st = synthetic.make_stack(9, 75, 131, seed=530)
    for kw in (dict(bin_size=7, method="max_std"), dict(bin_size=3, method="max_averages", atoh_shift=-1)):
        proj, zmap = sp.time_point_surface_projection(st, "CZYX", 0, airyscan=False, z_map=True, build_manifold=True, **kw)
        p_ref, z_ref = orc.time_point_surface_projection(st, "CZYX", 0, airyscan=False, z_map=True, build_manifold=True, **kw)
        np.testing.assert_array_equal(zmap, z_ref)
        np.testing.assert_array_equal(proj, p_ref)


@pytest.mark.parametrize("shape,start", [((30, 300, 340), None), ((50, 90, 1500), (7, 40, 2)), ((9, 2600, 37), (3, 2599, 36)),
                                         ((12, 1300, 1300), (5, 0, 650)), ((3, 1200, 1100), (1, 600, 0))])
def test_manifold_vs_oracle(shape, start):
    """Runs longer than one scan chunk (1024 pixels; 512 for Z = 50), starts on edges and corners, thin frames."""
    from oracle import oracle as orc
    from tissue_image_processing_amd import surface_projection as sp
    rng = np.random.default_rng(sum(shape))
    s = rng.random(shape).astype(np.float32)
    # smooth in z so that neighbouring planes compete (all three branches of the plane rule are taken)
    s = (s + np.roll(s, 1, axis=0) + np.roll(s, -1, axis=0)).astype(np.float32)
    if start is not None:
        s[start] = 100.0
    ref = orc.build_continues_manifold(s)
    out = sp.build_continues_manifold(s)
    assert int((out != ref).sum()) == 0


def test_manifold_headline_frame():
    """2048 x 2048 x 30 (BASELINE's frame size): equal to the serial restatement; prints both times."""
    from oracle import oracle as orc
    from tissue_image_processing_amd import surface_projection as sp
    rng = np.random.default_rng(2048)
    small = rng.random((30, 64, 64)).astype(np.float32)
    s = np.repeat(np.repeat(small, 32, axis=1), 32, axis=2) + 0.05 * rng.random((30, 2048, 2048), dtype=np.float32)
    t0 = time.perf_counter(); ref = orc.build_continues_manifold(s); t1 = time.perf_counter()
    out = sp.build_continues_manifold(s); t2 = time.perf_counter()
    out = sp.build_continues_manifold(s); t3 = time.perf_counter()
    print("manifold 2048^2 x 30: oracle (C, one core) %.2f s, device %.3f s (first call %.3f s, both incl. the 503 MB upload)"
          % (t1 - t0, t3 - t2, t2 - t1))
    assert int((out != ref).sum()) == 0
