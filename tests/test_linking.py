"""T1 (ti.py:1881-1938): the trackpy-model linker in linking.py.  Parity with trackpy itself is unpinned (third-party,
not installed); these tests check the model's own properties and the Tissue method around it on synthetic movement."""
import numpy as np
import pandas as pd
import pytest

from tissue_image_processing_amd import linking


def _frame(rng, n=400, size=600.0):
    pts = rng.uniform(0, size, (n, 2))
    area = rng.uniform(500, 1500, n)
    return pts, area


def test_embedding_is_the_reference_distance():
    from tissue_image_processing_amd.tissue_info import TissueHipMixin
    rng = np.random.default_rng(0)
    a, b = rng.uniform(0, 100, 3) + 1, rng.uniform(0, 100, 3) + 1
    ea, eb = linking.embed(a[0], a[1], a[2]), linking.embed(b[0], b[1], b[2])
    assert abs(np.linalg.norm(ea - eb) - TissueHipMixin.tracking_dist_func(a, b)) < 1e-12


def test_identity_survives_drift_jitter_and_reordering():
    rng = np.random.default_rng(1)
    pts, area = _frame(rng)
    lk = linking.FrameLinker()
    ids = lk.link(linking.embed(pts[:, 0], pts[:, 1], area))
    assert np.array_equal(ids, np.arange(pts.shape[0]))           # consecutive particle numbers in feature order
    truth = ids.copy()
    for _ in range(4):
        perm = rng.permutation(pts.shape[0])
        pts = pts[perm] + rng.normal(0, 1.0, pts.shape) + np.array([0.5, -0.3])
        area = area[perm] * rng.uniform(0.98, 1.02, area.shape)
        truth = truth[perm]
        got = lk.link(linking.embed(pts[:, 0], pts[:, 1], area))
        assert (got == truth).mean() > 0.995


def test_memory_relinks_after_a_gap_and_forgets_after_it():
    lk = linking.FrameLinker(search_range=20, memory=2)
    base = np.array([[10.0, 10.0, 800.0], [200.0, 200.0, 900.0]])
    ids0 = lk.link(linking.embed(base[:, 0], base[:, 1], base[:, 2]))
    only_first = linking.embed(base[:1, 0], base[:1, 1], base[:1, 2])
    assert lk.link(only_first)[0] == ids0[0]                      # second particle missing: age 1
    assert lk.link(only_first)[0] == ids0[0]                      # age 2
    back = lk.link(linking.embed(base[:, 0] + 1, base[:, 1], base[:, 2]))
    assert back[1] == ids0[1]                                     # still remembered
    lk2 = linking.FrameLinker(search_range=20, memory=1)
    i0 = lk2.link(linking.embed(base[:, 0], base[:, 1], base[:, 2]))
    lk2.link(only_first); lk2.link(only_first)
    again = lk2.link(linking.embed(base[:, 0], base[:, 1], base[:, 2]))
    assert again[0] == i0[0] and again[1] not in i0               # forgotten: a new particle number


def test_subnet_assignment_is_globally_optimal_not_greedy():
    # two tracks, two features: the greedy nearest link (a->x, 1.0) forces b->y at 9.0 (sum d^2 = 82);
    # the optimum is a->y (5), b->x (5): sum d^2 = 50
    lk = linking.FrameLinker(search_range=50, adaptive_stop=None)
    a, b = (0.0, 0.0), (0.0, 6.0)
    x, y = (0.0, 1.0), (0.0, -5.0)
    ids0 = lk.link(linking.embed([a[0], b[0]], [a[1], b[1]], [800, 800]))
    ids1 = lk.link(linking.embed([x[0], y[0]], [x[1], y[1]], [800, 800]))
    assert ids1[0] == ids0[1] and ids1[1] == ids0[0]


def test_oversize_subnets_shrink_the_range_or_fail_like_trackpy():
    rng = np.random.default_rng(3)
    pts = rng.uniform(0, 300, (200, 2))                           # everything within reach of everything at range 100
    area = np.full(200, 900.0)
    lk = linking.FrameLinker(search_range=100, adaptive_stop=None)
    lk.link(linking.embed(pts[:, 0], pts[:, 1], area))
    with pytest.raises(linking.SubnetOversizeError):
        lk.link(linking.embed(pts[:, 0] + 0.5, pts[:, 1], area))
    lk = linking.FrameLinker(search_range=100, adaptive_stop=2)
    ids0 = lk.link(linking.embed(pts[:, 0], pts[:, 1], area))
    ids1 = lk.link(linking.embed(pts[:, 0] + 0.5, pts[:, 1], area))
    assert np.array_equal(ids0, ids1)


def test_tissue_method_labels_tracks_across_frames():
    from tissue_image_processing_amd.tissue_info import Tissue, make_df, CELL_INFO_SPECS
    rng = np.random.default_rng(5)
    n, T = 150, 4
    pts, area = _frame(rng, n=n, size=500.0)
    tis = Tissue(T)
    tis.drifts[1:] = np.array([0.5, -0.3])                        # stored drifts are reused (ti.py:1884,1912-1916)
    truth = []
    order = np.arange(n)
    for t in range(T):
        if t:
            perm = rng.permutation(n)
            pts = pts[perm] + rng.normal(0, 0.7, pts.shape) - np.array([0.5, -0.3])
            area, order = area[perm], order[perm]
        df = make_df(n, CELL_INFO_SPECS).astype({"cy": float, "cx": float, "area": float})
        df.loc[:, "cy"], df.loc[:, "cx"], df.loc[:, "area"] = pts[:, 0], pts[:, 1], area
        df.loc[:, "valid"], df.loc[:, "empty_cell"] = 1, 0
        df.loc[:, "label"] = np.arange(1, n + 1)
        tis.set_cells_info(t + 1, df)
        truth.append(order.copy())
    assert tis.track_cells_with_trackpy() == T
    first = tis.get_cells_info(1).label.to_numpy()
    assert np.array_equal(first, np.arange(1, n + 1))
    for t in range(1, T):
        got = tis.get_cells_info(t + 1).label.to_numpy()
        assert (got == truth[t] + 1).mean() > 0.99
        assert np.unique(got).size == n                           # no duplicated ids in a frame


def test_duplicated_ids_are_resolved_like_upstream():
    from tissue_image_processing_amd.tissue_info import Tissue, make_df, CELL_INFO_SPECS
    tis = Tissue(1)
    df = make_df(5, CELL_INFO_SPECS)
    df.loc[:, "label"] = [3, 3, 7, 3, 9]
    df.loc[:, "valid"] = [0, 1, 1, 1, 1]
    tis.set_cells_info(1, df)
    tis.fix_duplicated_label_cells_in_frame(1)
    # row 1 (first valid) keeps 3; row 0 -> index+1 = 1; row 3 -> index+1 = 4
    assert tis.get_cells_info(1).label.tolist() == [1, 3, 7, 4, 9]
