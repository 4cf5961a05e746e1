"""GPU parity: rank filters, CCL, watershed, cell tables, tracking (through the C-ABI) vs goldens and the oracle."""
import numpy as np
import pytest

from gpu_util import taps_patch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def env(monkeypatch, golden_taps, oracle_with_golden_taps):
    from tissue_image_processing_amd import basic_image_manipulations as bim
    from tissue_image_processing_amd import _segmentation as seg
    from tissue_image_processing_amd import tissue_info as ti
    taps_patch(monkeypatch, golden_taps)
    return bim, seg, ti, oracle_with_golden_taps


def label_iou(test, ref):
    ious = []
    for l in np.unique(ref):
        if l == 0:
            continue
        m = ref == l
        cand = np.bincount(test[m])
        cand[0] = 0
        if cand.sum() == 0:
            ious.append(0.0)
            continue
        k = cand.argmax()
        ious.append((m & (test == k)).sum() / float((m | (test == k)).sum()))
    return float(np.mean(ious))


def test_rank_filters_golden(env, golden):
    _, seg, _, _ = env
    g = golden("rank_filters")
    lab, img = g["lab"], g["img"]
    np.testing.assert_array_equal(seg.maximum_filter(lab, (5, 5), mode="constant"), g["max5_const"])
    np.testing.assert_array_equal(seg.maximum_filter(lab, (3, 3), mode="constant"), g["max3_const"])
    np.testing.assert_array_equal(seg.maximum_filter(lab, footprint=True, mode="constant"), g["max_cross_const"])
    np.testing.assert_array_equal(seg.minimum_filter(lab, footprint=True, mode="constant"), g["min_cross_const"])
    np.testing.assert_array_equal(seg.maximum_filter(img, 7), g["max7_reflect_f64"])
    np.testing.assert_array_equal(seg.maximum_filter(img, 4), g["max4_reflect_f64"])
    b = g["binimg"]
    np.testing.assert_array_equal(seg.maximum_filter(b, 5), g["dil5"])
    np.testing.assert_array_equal(seg.minimum_filter(b, 5), g["ero5"])
    np.testing.assert_array_equal(seg.minimum_filter(b, 7), g["ero7"])
    np.testing.assert_array_equal(seg.minimum_filter(img, 7), g["ero7_gray"])


def test_label_golden(env, golden):
    _, seg, _, _ = env
    g = golden("label")
    out, n = seg.label(g["bin"], background=0, return_num=True, connectivity=1)
    np.testing.assert_array_equal(out, g["label_bg0"])
    assert n == g["label_bg0"].max()
    np.testing.assert_array_equal(seg.label(g["img255"], background=255, connectivity=1), g["label_bg255"])
    np.testing.assert_array_equal(seg.label(g["multi"], background=0, connectivity=1), g["label_multi_bg0"])
    np.testing.assert_array_equal(seg.label(g["snake"], background=0, connectivity=1), g["label_snake"])


def test_label_large_vs_oracle(env):
    _, seg, _, orc = env
    rng = np.random.default_rng(3)
    a = (rng.random((700, 900)) > 0.42).astype(np.int32)
    a[100:400, 200:600] = 1   # one huge component
    a[::7, :] = 0
    ref, n_ref = orc.label4(a, 0)
    out, n = seg.label(a, background=0, return_num=True, connectivity=1)
    assert n == n_ref
    np.testing.assert_array_equal(out, ref)


@pytest.mark.parametrize("shape", [(64, 64), (65, 97), (300, 421), (700, 900), (2048, 2048)])
def test_tiled_union_find_equals_one_level(env, shape):
    """Connected components by tiles in LDS + a border pass (the default from 64 x 64 pixels on) against the one-level union-find
    on global memory and the oracle: blobs that span many tiles, a spiral that crosses every tile border many times, noise; the
    raster-order numbering (roots = raster-first pixels) has to survive both levels."""
    _, seg, _, orc = env
    from tissue_image_processing_amd import _lib
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    a = (rng.random(shape) > 0.45).astype(np.int32)
    a[(np.sin(yy / 9.0) * np.cos(xx / 13.0)) > 0.3] = 1                     # blobs of a few hundred to a few thousand pixels
    r = np.hypot(yy - shape[0] / 2, xx - shape[1] / 2)
    a[(np.floor(r + 4 * np.arctan2(yy - shape[0] / 2, xx - shape[1] / 2) / np.pi) % 8) == 0] = 1   # a spiral arm
    a[::11, ::3] = 0
    out, n = seg.label(a, background=0, return_num=True, connectivity=1)
    with _lib.tuning(TIP_UF_ONE_LEVEL="1"):
        out1, n1 = seg.label(a, background=0, return_num=True, connectivity=1)
    assert n == n1
    np.testing.assert_array_equal(out, out1)
    if shape[0] * shape[1] <= 700 * 900:
        ref, n_ref = orc.label4(a, 0)
        assert n == n_ref
        np.testing.assert_array_equal(out, ref)


@pytest.mark.parametrize("case", ["ii", "iii", "iv"])
def test_watershed_float_golden_bit_exact(env, golden, case):
    """Distinct-valued float landscapes: labels identical to skimage's serial flood."""
    _, seg, _, _ = env
    g = golden("watershed")
    out, flags = seg.watershed(g[case + "_img"], return_flags=True)
    assert out.dtype == np.int32
    mism = int((out != g[case + "_labels"]).sum())
    assert mism == 0, "label mismatches: %d (flags %d)" % (mism, flags)
    assert not (flags & 2)


def test_watershed_misc_golden(env, golden):
    _, seg, _, _ = env
    g = golden("watershed")
    np.testing.assert_array_equal(seg.watershed(np.full((12, 14), 3.5)), g["vii_const_labels"])
    np.testing.assert_array_equal(seg.watershed(g["vii_bowl"]), g["vii_bowl_labels"])


def test_watershed_segmentation_golden_bit_exact(env, golden):
    bim, _, _, _ = env
    g = golden("watershed")
    for key, args, lk in [("i_img", (0.03, 3, 3), "i_labels"), ("i2_img", (0.03, 3, 3), "i2_labels"),
                          ("i2_img", (0.2, 2, 4), "i3_labels")]:
        out = bim.watershed_segmentation(g[key], *args)
        assert out.dtype == np.int32
        mism = int((out != g[lk]).sum())
        assert mism == 0, "%s: %d label mismatches" % (lk, mism)


def test_watershed_binary_bit_exact(env, golden):
    """Two-valued boundary image (pl.py:194): every marker has the same heap key, so skimage's result depends on the pop
    order its array heap gives equal keys and on FIFO (age) order afterwards.  Mode B reproduces both: labels identical."""
    _, seg, _, _ = env
    g = golden("watershed")
    out, flags = seg.watershed(g["vi_boundary"], return_flags=True)
    assert flags & 2
    mism = int((out != g["vi_labels"]).sum())
    assert mism == 0, "binary watershed: %d label mismatches" % mism


def boundary_image(N, seed, invert=False):
    """A {0, 255} boundary image as pl.py:167-193 produces it, from a synthetic tessellation (oracle rank filters)."""
    from oracle import oracle as orc
    from tissue_image_processing_amd import synthetic
    sites = synthetic.make_sites(N, N, seed=seed)[0]
    d1, d2, i1 = synthetic._two_nearest(sites, N, N)
    blob = (np.exp(-(d2 - d1) ** 2 / 4) < 0.5) & (i1 % 3 != 0)
    if invert:
        blob = ~blob
    closed = orc.erosion(orc.dilation(255.0 * blob, 5), 5)
    hc = orc.erosion(closed, 7)
    return orc.dilation(closed - hc, 5)


@pytest.mark.parametrize("N,seed,invert", [(256, 1, False), (384, 2, False), (300, 3, True), (1024, 4, False)])
def test_watershed_binary_vs_oracle(env, N, seed, invert):
    """Mode B against the oracle's literal heap flood on boundary images of growing size (up to ~0.8 M markers)."""
    _, seg, _, orc = env
    img = boundary_image(N, seed, invert)
    ref = orc.watershed(img)
    out, flags = seg.watershed(img, return_flags=True)
    assert flags & 2
    mism = int((out != ref).sum())
    print("binary %d^2: %d labels, %.0f%% markers, mismatches %d" % (N, ref.max(), 100 * float((img == 0).mean()), mism))
    assert mism == 0


def test_watershed_binary_degenerate_shapes(env):
    """Two-valued images with thick high-valued regions (hundreds of generations), single rows / columns, and a high-valued
    pixel enclosed by two markers."""
    _, seg, _, orc = env
    rng = np.random.default_rng(9)
    imgs = []
    a = np.full((120, 150), 255.0)
    a[5:9, 5:9] = 0; a[100:104, 130:140] = 0; a[60, 70] = 0            # three small markers flood a big plateau
    imgs.append(a)
    imgs.append((rng.random((90, 110)) > 0.35) * 255.0)                # salt-and-pepper: thousands of one-pixel markers
    imgs.append(((np.arange(97) % 7) > 2)[None, :] * 255.0)            # 1 x N
    imgs.append(((np.arange(53) % 5) > 1)[:, None] * 255.0)            # N x 1
    b = np.zeros((9, 9)); b[4, :] = 255.0; b[:, 4] = 255.0             # a cross separating four markers
    imgs.append(b)
    imgs.append(np.where(rng.random((64, 200)) > 0.5, 7.25, -3.0))     # any two values, not just {0, 255}
    for k, img in enumerate(imgs):
        ref = orc.watershed(img)
        out, flags = seg.watershed(np.ascontiguousarray(img, np.float64), return_flags=True)
        assert flags & 2, k
        mism = int((out != ref).sum())
        assert mism == 0, "case %d: %d mismatches" % (k, mism)


@pytest.mark.parametrize("small,batch", [("0", "8"), ("1", "1"), ("48", "2"), ("700", "3"), ("32768", "4")])
def test_watershed_binary_small_generation_kernel(env, small, batch):
    """The two-valued flood's late generations run in ONE workgroup (k_mb_small_gens) as soon as a generation has at most
    TIP_MB_SMALL pixels, and go back to the grid-wide kernels when a generation grows past it again.  Every threshold (0 = never)
    and every batching of the host's looks at the device state gives the oracle's labels: generations that shrink (boundary
    bands), grow (three small markers flooding a plateau: the front widens for a hundred generations) and do both."""
    _, seg, _, orc = env
    from tissue_image_processing_amd import _lib
    rng = np.random.default_rng(21)
    imgs = [boundary_image(384, 2)]
    a = np.full((150, 190), 255.0)
    a[5:9, 5:9] = 0; a[100:104, 130:140] = 0; a[60, 70] = 0
    imgs.append(a)
    b = np.full((200, 200), 255.0)                                      # a plateau behind a one-pixel gate: small, large, small again
    b[:, 60] = 0; b[100, 60] = 255.0; b[:, :3] = 0; b[20:24, 150:154] = 0
    imgs.append(b)
    imgs.append((rng.random((90, 110)) > 0.35) * 255.0)
    refs = [orc.watershed(im) for im in imgs]
    with _lib.tuning(TIP_MB_SMALL=small, TIP_MB_BATCH=batch):
        for k, (im, ref) in enumerate(zip(imgs, refs)):
            out, flags = seg.watershed(np.ascontiguousarray(im, np.float64), return_flags=True)
            assert flags & 2, k
            mism = int((out != ref).sum())
            assert mism == 0, "case %d (small %s, batch %s): %d mismatches" % (k, small, batch, mism)


def _uint16_frame_and_reference(orc, N, seed=44):
    """A uint16-normalised projection (save_tiff's normalisation, bim.py:183-188) and the oracle's restatement of what
    bim.py:446-476 does to it: threshold, scipy's integer-dtype blur (truncation after every axis), serial flood."""
    from tissue_image_processing_amd import synthetic, surface_projection as sp
    st = synthetic.make_stack(10, N, N, seed=seed)
    proj, _ = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
    img16 = np.round(proj[0] / proj[0].max() * 65535).astype(np.uint16)
    s16 = img16.copy()
    thr = orc.threshold_local_generic_max(s16.astype(np.float64), 0.03, 3)
    s16[s16 < thr] = 0
    cur = s16.astype(np.float64)
    for ax in range(2):
        sg = [0, 0]
        sg[ax] = 3
        cur = np.trunc(orc.blur_image(cur, tuple(sg)))
    return img16, orc.watershed(cur)


def test_watershed_value_ties_are_exact(env, golden):
    """Landscapes whose non-marker pixels tie in value (integer images: the GUI's uint16 frames, gui.py:1841-1845; golden `v`,
    a landscape quantised to a few levels): skimage orders equal values by heap push age and equal-keyed markers by the
    mechanics of its array heap.  The default tie policy reproduces that bit for bit (markers on the device, the flood
    itself as the serial replay of csrc/tip_ws_serial.hip): golden `v` from the reference, and the classical path on a
    512^2 uint16 frame against the oracle."""
    _, seg, _, orc = env
    from tissue_image_processing_amd import _lib
    g = golden("watershed")
    out, flags = seg.watershed(g["v_img"], return_flags=True)
    assert flags & _lib.WS_FLAG_TIES and flags & _lib.WS_FLAG_SERIAL_EXACT and not (flags & _lib.WS_FLAG_TWO_VALUED)
    np.testing.assert_array_equal(out, g["v_labels"])
    img16, ref16 = _uint16_frame_and_reference(orc, 512)
    lab, flags = seg.watershed_segmentation(img16, 0.03, 3, 3, return_flags=True)
    assert flags & _lib.WS_FLAG_SERIAL_EXACT
    np.testing.assert_array_equal(lab, ref16)


def test_watershed_ties_between_pixels_that_only_share_a_neighbour(env, golden):
    """Value ties between NON-ADJACENT non-marker pixels (diagonals, distance two; no 4-adjacent tie anywhere): the detector
    flags them too (a pulled pixel between the two sees one before the other), the exact policy replays them, and the result
    equals skimage's on images picked because a raster tie-break gets them wrong (tools/make_goldens_ties.py)."""
    _, seg, _, _ = env
    from tissue_image_processing_amd import _lib
    g = golden("watershed_diag_ties")
    for k in range(6):
        out, flags = seg.watershed(g["img%d" % k], return_flags=True)
        assert flags & _lib.WS_FLAG_TIES and flags & _lib.WS_FLAG_SERIAL_EXACT, (k, flags)
        np.testing.assert_array_equal(out, g["labels%d" % k])


def test_watershed_value_ties_exact_at_2048(env):
    """The same at the headline frame size: a 2048^2 uint16-normalised frame through watershed_segmentation, 0 mismatches
    against the oracle's serial flood."""
    _, seg, _, orc = env
    from tissue_image_processing_amd import _lib
    import time
    img16, ref16 = _uint16_frame_and_reference(orc, 2048, seed=45)
    t0 = time.perf_counter()
    lab, flags = seg.watershed_segmentation(img16, 0.03, 3, 3, return_flags=True)
    dt = time.perf_counter() - t0
    mism = int((lab != ref16).sum())
    print("uint16 2048^2 frame: %d labels, exact tie policy %.2f s, mismatches %d" % (ref16.max(), dt, mism))
    assert flags & _lib.WS_FLAG_SERIAL_EXACT
    assert mism == 0


def test_watershed_fast_tie_policy_deviation_is_bounded(env, golden):
    """tip_set_tuning("TIP_WS_TIES", "fast") keeps tie landscapes on the device: mode A breaks ties by raster index instead of
    push age, so watershed lines on plateaus can sit a pixel off (same markers, same label count).  The deviation is
    measured here so that it cannot grow silently; the default policy (above) has none."""
    _, seg, _, orc = env
    from tissue_image_processing_amd import _lib
    g = golden("watershed")
    with _lib.tuning(TIP_WS_TIES="fast"):
        out, flags = seg.watershed(g["v_img"], return_flags=True)
        ref = g["v_labels"]
        assert flags & _lib.WS_FLAG_TIES and not (flags & (_lib.WS_FLAG_TWO_VALUED | _lib.WS_FLAG_SERIAL_EXACT))
        assert out.max() == ref.max()
        frac_v = float((out != ref).mean())
        print("fast policy, golden v (few-level landscape): mismatching pixels %.2f%%, IoU %.3f" % (100 * frac_v, label_iou(out, ref)))
        assert frac_v < 0.40
        img16, ref16 = _uint16_frame_and_reference(orc, 512)
        lab, flags = seg.watershed_segmentation(img16, 0.03, 3, 3, return_flags=True)
        frac = float((lab != ref16).mean())
        iou = label_iou(lab, ref16)
        print("fast policy, uint16 512^2 frame: %d labels (ref %d), mismatching pixels %.3f%%, IoU %.4f, serial finish %d px" % (
            lab.max(), ref16.max(), 100 * frac, iou, flags >> _lib.WS_FLAG_COUNT_SHIFT))
        assert flags & _lib.WS_FLAG_TIES and not (flags & _lib.WS_FLAG_SERIAL_EXACT)
        assert lab.max() == ref16.max()
        assert frac < 0.02 and iou > 0.97


def test_watershed_large_plateau_does_not_crawl(env):
    """A noisy integer image with large equal-valued plateaus under the FAST policy: mode A's certificates cannot close
    plateau-sized pockets, and the rest used to be committed one pixel per host round trip (minutes).  Now the stalled rest
    is finished in one serial pass on the host: seconds, and equal to a serial flood with the same (value, raster index)
    order."""
    _, seg, _, orc = env
    from tissue_image_processing_amd import _lib
    import time
    rng = np.random.default_rng(9)
    img = rng.integers(0, 3, (600, 700)).astype(np.float64)
    img[100:400, 150:600] = 1.0                     # a 135 000-pixel plateau
    with _lib.tuning(TIP_WS_TIES="fast"):
        t0 = time.perf_counter()
        out, flags = seg.watershed(img, return_flags=True)
        dt = time.perf_counter() - t0
    print("large plateau, fast policy: %.2f s, flags %#x, serial finish %d px" % (dt, flags & 0xff, flags >> _lib.WS_FLAG_COUNT_SHIFT))
    assert dt < 20.0
    # raster-index tie order == the serial flood of an image whose ties are broken by a tiny raster-increasing ramp
    ramp = img + np.arange(img.size).reshape(img.shape) * 1e-9
    mk = orc.label4(orc.local_minima(img).astype(np.int32), 0)[0]
    np.testing.assert_array_equal(out, orc.watershed(ramp, markers=mk))
    out_exact, fl = seg.watershed(img, return_flags=True)
    assert fl & _lib.WS_FLAG_SERIAL_EXACT
    np.testing.assert_array_equal(out_exact, orc.watershed(img))


def test_watershed_vs_oracle_synthetic_frame(env):
    """Full classical segmentation of a synthetic frame, labels bit-identical to the oracle."""
    bim, _, _, orc = env
    from tissue_image_processing_amd import synthetic
    from tissue_image_processing_amd import surface_projection as sp
    st = synthetic.make_stack(10, 384, 512, seed=33)
    proj, _ = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
    zo = proj[0].T.copy()   # gui.py:1841-1844 passes the transposed ZO projection
    out, flags = __import__("tissue_image_processing_amd._segmentation", fromlist=["x"]).watershed_segmentation(
        zo, 0.03, 3, 3, return_flags=True)
    ref = orc.watershed_segmentation(zo, 0.03, 3, 3)
    mism = int((out != ref).sum())
    print("synthetic frame: %d labels, flags %#x, mismatches %d" % (ref.max(), flags, mism))
    assert flags == 0          # tie-free float landscape: the data-parallel flood, no serial stage
    assert mism == 0


def test_watershed_tile_flavours_and_openings_agree(env):
    """The tile kernel's selectable flavours (TIP_WS_TILE: interior-only / evaluated margins, event-driven list or not, 8-,
    16-, 32-pixel tiles) and openings (TIP_WS_OPEN) only change the schedule of certified decisions: every one of them gives
    the oracle's labels."""
    bim, _, _, orc = env
    from tissue_image_processing_amd import synthetic, _segmentation as seg, _lib
    from tissue_image_processing_amd import surface_projection as sp
    st = synthetic.make_stack(8, 300, 340, seed=35)
    proj = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False)
    ref = orc.watershed_segmentation(proj[0], 0.03, 3, 3)
    for variant, opening in (("0", "10,8"), ("4", "6,6"), ("5", "8,6"), ("6", "6,6"), ("9", "3,2"), ("12", "12,9"), ("14", "1,1"),
                             ("15", "8,6")):
        with _lib.tuning(TIP_WS_TILE=variant, TIP_WS_OPEN=opening):
            out = seg.watershed_segmentation(proj[0], 0.03, 3, 3)
        assert int((out != ref).sum()) == 0, (variant, opening)
    with _lib.tuning(TIP_WS_NO_SKIP="1"):                     # stuck tiles re-run on every wake-up
        assert int((seg.watershed_segmentation(proj[0], 0.03, 3, 3) != ref).sum()) == 0


def test_headline_frame_segmentation_and_tables_bit_exact(env):
    """BASELINE's headline frame (2048x2048x30, C=2): the classical segmentation of the GPU projection and the cell
    tables on it, against the oracle's exact heap flood at FULL size (about 20 s of CPU)."""
    bim, _, _, orc = env
    from tissue_image_processing_amd import synthetic, _segmentation as seg
    from tissue_image_processing_amd import surface_projection as sp
    st = synthetic.make_stack(30, 2048, 2048, seed=100)        # bench.py's frame
    proj, _ = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
    zo = proj[0]
    out, flags = seg.watershed_segmentation(zo, 0.03, 3, 3, return_flags=True)
    ref = orc.watershed_segmentation(zo, 0.03, 3, 3)
    assert int(ref.max()) > 4000
    mism = int((out != ref).sum())
    print("headline frame: %d labels, flags %#x, mismatches %d" % (ref.max(), flags, mism))
    assert flags == 0          # tie-free float landscape: the data-parallel flood, no serial stage
    assert mism == 0
    got = seg.regionprops_arrays(out)
    want = orc.regionprops(ref)
    for k in ("area", "bbox", "cy", "cx"):
        np.testing.assert_array_equal(got[k], want[k])
    np.testing.assert_allclose(got["perimeter"], want["perimeter"], rtol=1e-13)
    gp = seg.neighbor_pairs(out)
    wp = orc.neighbor_pairs(ref)
    assert set(map(tuple, np.asarray(gp).tolist())) == set(map(tuple, np.asarray(wp).tolist()))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_cellinfo_golden(env, golden, tag):
    _, seg, ti, _ = env
    g = golden("cellinfo")
    lab = g[tag + "_labels"]
    t = ti.Tissue(1)
    t.set_labels(1, lab.copy(), reset_data=True)
    t.calculate_frame_cellinfo(1)
    ci = t.get_cells_info(1)
    np.testing.assert_array_equal(ci.area.to_numpy(), g[tag + "_area"])
    np.testing.assert_array_equal(ci.label.to_numpy(), g[tag + "_label"])
    np.testing.assert_allclose(ci.perimeter.to_numpy(), g[tag + "_perimeter"], rtol=1e-13)
    np.testing.assert_array_equal(ci.cx.to_numpy(), g[tag + "_cx"])
    np.testing.assert_array_equal(ci.cy.to_numpy(), g[tag + "_cy"])
    for col in ["bounding_box_min_row", "bounding_box_min_col", "bounding_box_max_row", "bounding_box_max_col",
                "valid", "n_neighbors"]:
        np.testing.assert_array_equal(ci[col].to_numpy(), g[tag + "_" + col], err_msg=col)
    nb = g[tag + "_neighbors"]
    for i, s in enumerate(ci.neighbors):
        assert sorted(int(v) for v in s) == [int(v) for v in nb[i] if v > 0], i
    np.testing.assert_array_equal(t.calc_neighbors_contact_matrix(1), g[tag + "_contact"])


def test_contact_matrix_vs_oracle_synthetic_frame(env):
    """C6 on a few hundred cells: the device pair histogram (tip_contact_pairs_i32) against the reference's per-cell,
    per-neighbour counting inside bounding boxes (oracle), including labels that do not occur and cells at the border."""
    bim, seg, ti, orc = env
    rng = np.random.default_rng(8)
    land = orc.blur_image(rng.random((300, 420)), 4.0)
    lab = orc.watershed(land)
    lab[lab == 7] = 0                      # a label that does not occur any more
    lab[120:130, 200:260] = 0              # a thick line: neighbours (5x5 rule) that do not touch (cross rule)
    t = ti.Tissue(1)
    t.set_labels(1, lab.copy(), reset_data=True)
    t.calculate_frame_cellinfo(1)
    t.find_neighbors(1)
    ci = t.get_cells_info(1)
    want = orc.contact_matrix(lab, list(ci.neighbors))
    got = t.calc_neighbors_contact_matrix(1)
    np.testing.assert_array_equal(got, want)
    assert got.sum() > 1000


def test_update_labels_golden(env, golden):
    _, _, ti, _ = env
    g = golden("cellinfo")
    t = ti.Tissue(1)
    t.set_labels(1, g["u_in"].copy(), reset_data=True)
    t.calculate_frame_cellinfo(1)
    t.update_labels(1)
    np.testing.assert_array_equal(t.get_labels(1), g["u_out"])


def test_celltypes_pins(env, golden):
    _, seg, ti, _ = env
    g = golden("celltypes")
    lab, inten = g["labels"], g["intensity"]
    rp = seg.regionprops_arrays(lab, intensity=inten)
    np.testing.assert_allclose(rp["intensity_mean"], g["mean"], rtol=1e-12)
    np.testing.assert_array_equal(ti.find_local_maxima(inten, window_size=7), g["local_maxima"])
    t = ti.Tissue(1)
    t.set_labels(1, lab.copy(), reset_data=True)
    t.calculate_frame_cellinfo(1)
    t.calc_cell_types(inten, 1, "HC", threshold=0.5, percentage_above_threshold=90)
    ci = t.get_cells_info(1)
    # positives are exactly the cells whose 10th percentile exceeds 0.5 * p99 (ti.py:2369-2373)
    expect = g["p10"] > 0.5 * g["p99"]
    got = np.asarray(ti.is_positive_for_type(ci.type.to_numpy(), 0))
    np.testing.assert_array_equal(got, expect)
    types = t.get_cell_types(1)
    assert types.shape == lab.shape


def test_percentiles_by_radix_select_equal_numpy(env, golden):
    """C5's dense part: np.percentile per label and over the frame (ti.py:2349-2355, 2371) from exact order statistics
    found by radix select on the device; values with many ties, negative values, single-pixel labels, absent labels."""
    _, seg, _, _ = env
    g = golden("celltypes")
    lab, inten = g["labels"], g["intensity"]
    n = int(lab.max())
    counts = np.bincount(lab.ravel(), minlength=n + 1)[1:]
    np.testing.assert_array_equal(seg.percentile_per_label(lab, inten, n, counts, 10), g["p10"])
    assert seg.percentile_frame(inten, 99) == float(g["p99"])
    rng = np.random.default_rng(2)
    lab2 = rng.integers(0, 400, (300, 500)).astype(np.int32)
    lab2[lab2 == 17] = 0                                      # label 17 absent
    lab2[0, 0] = 399
    img = np.round(rng.normal(0, 50, lab2.shape))             # integers around zero: heavy ties, both signs
    img[5:40, 5:90] = -0.0
    n2 = 400
    c2 = np.bincount(lab2.ravel(), minlength=n2 + 1)[1:]
    for q in (0, 10, 37.5, 50, 99, 100):
        want = np.array([np.percentile(img[lab2 == l], q) if c2[l - 1] else np.nan for l in range(1, n2 + 1)])
        got = seg.percentile_per_label(lab2, img, n2, c2, q)
        np.testing.assert_array_equal(got, want)
        assert seg.percentile_frame(img, q) == np.percentile(img, q)
    big = rng.random((1200, 1500)) * 1e6
    for q in (1, 99):
        assert seg.percentile_frame(big, q) == np.percentile(big, q)


def test_tracking_golden(env, golden):
    _, _, ti, _ = env
    g = golden("tracking")
    labs = g["labels"]
    t = ti.Tissue(labs.shape[0])
    for f in range(labs.shape[0]):
        t.set_labels(f + 1, labs[f].copy(), reset_data=True)
        t.calculate_frame_cellinfo(f + 1)
    t.drifts[1] = (0.5, -0.3)
    t.drifts[2] = (0.5, -0.3)
    frames = list(t.track_cells_iterator(1, labs.shape[0]))
    assert frames == [2, 3]
    for f in range(labs.shape[0]):
        np.testing.assert_array_equal(t.get_cells_info(f + 1).label.to_numpy(), g["ids_%d" % f])
    tl = t.get_trackking_labels(3)
    lut = np.insert(g["ids_2"], 0, 0)
    np.testing.assert_array_equal(tl, lut[labs[2]])


def test_calculate_mean_intensity(env, golden):
    """ti.py:1135-1150 (regionprops 'intensity_mean' per cell, cached in the table): against numpy's per-label mean; the
    property name upstream uses exists only in skimage >= 0.19, so there is no golden from the 0.18.3 environment."""
    from tissue_image_processing_amd import tissue_info as ti
    g = golden("celltypes")
    lab, inten = g["labels"].copy(), g["intensity"]
    lab[lab == 7] = 0                                   # a label that does not occur
    t = ti.Tissue(1, "movie", ["zo", "atoh"])
    t.set_labels(1, lab, reset_data=True)
    t.calculate_frame_cellinfo(1)
    valid = t.get_cells_info(1).query("valid == 1 and empty_cell == 0")
    means = t.calculate_mean_intensity(1, valid, inten, "atoh")
    want = np.array([inten[lab == i + 1].mean() for i in valid.index])
    np.testing.assert_allclose(means, want, rtol=1e-12)
    col = t.get_cells_info(1)["mean_intensity_atoh"].to_numpy()
    assert np.isnan(col[6]) and not np.isnan(col[valid.index]).any()
    again = t.calculate_mean_intensity(1, t.get_cells_info(1).query("valid == 1 and empty_cell == 0"), inten * 0, "atoh")
    np.testing.assert_array_equal(again, means)          # cached column wins, as upstream


def test_whole_movie_refreshes(env, golden):
    """update_bounding_box_for_all_cells / update_neighbors_for_all_cells (ti.py:4230-4247) restore what
    calculate_frame_cellinfo wrote after the columns have been wiped."""
    from tissue_image_processing_amd import tissue_info as ti
    g = golden("cellinfo")
    t = ti.Tissue(2, "movie", ["zo"])
    for f, key in ((1, "a_labels"), (2, "b_labels")):
        t.set_labels(f, g[key].copy(), reset_data=True)
        t.calculate_frame_cellinfo(f)
    before = [t.get_cells_info(f).copy(deep=True) for f in (1, 2)]
    for f in (1, 2):
        info = t.get_cells_info(f)
        for edge in ("min_row", "min_col", "max_row", "max_col"):
            info["bounding_box_" + edge] = -5
        info["neighbors"] = [set() for _ in range(len(info))]
        info["n_neighbors"] = 0
    assert t.update_bounding_box_for_all_cells() == 0 and t.update_neighbors_for_all_cells() == 0
    for f in (1, 2):
        info = t.get_cells_info(f)
        for edge in ("min_row", "min_col", "max_row", "max_col"):
            present = before[f - 1]["area"].to_numpy() > 0
            np.testing.assert_array_equal(info["bounding_box_" + edge].to_numpy()[present],
                                          before[f - 1]["bounding_box_" + edge].to_numpy()[present])
        valid = before[f - 1]["valid"].to_numpy() == 1
        got = [sorted(s) for s in info.neighbors]
        want = [sorted(s) for s in before[f - 1].neighbors]
        # (find_neighbors without a label list covers every non-empty cell: a superset of what the valid-only pass recorded)
        assert all(set(w) <= set(gv) for gv, w, v in zip(got, want, valid) if v)
