"""Pins the CPU oracle (oracle/) against golden vectors generated from the reference itself
(tools/make_goldens.py under the container's Anaconda interpreter).  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc


@pytest.fixture(autouse=True)
def _taps(oracle_with_golden_taps):
    # goldens were produced with numpy 1.26.4's (not correctly rounded) np.exp; use that environment's taps
    yield


def test_gaussian_bit_exact(golden):
    g = golden("gaussian")
    vol = g["vol_f32"]
    for tag, sig in [("s05_1_1", (0.5, 1, 1)), ("s05_30_30", (0.5, 30, 30)), ("s1_2_2", (1, 2, 2))]:
        out = orc.blur_image(vol, sig)
        assert out.dtype == np.float32
        np.testing.assert_array_equal(out, g["out_" + tag], err_msg=tag)
    np.testing.assert_array_equal(orc.blur_image(g["img_f64"], 3), g["out2d_s3"])
    np.testing.assert_array_equal(orc.blur_image(g["img_f64"], 7), g["out2d_s7"])
    np.testing.assert_array_equal(orc.blur_image(g["img_f64"].astype(np.float32), 3), g["out2d_f32_s3"])
    np.testing.assert_array_equal(orc.blur_image(g["tiny_f32"], (0.5, 30, 30)), g["tiny_out_s05_30_30"])


def test_gaussian_weights_c_matches_numpy():
    # libm exp (correctly rounded) vs np.exp (last-bit differences by numpy build): agree to ~2 ulp
    import ctypes
    for sigma in (0.5, 1.0, 2.0, 3.0, 7.0, 30.0):
        w = np.zeros(512)
        n = orc.lib().orc_gaussian_weights(ctypes.c_double(sigma), ctypes.c_double(4.0), orc._p(w), ctypes.c_long(512))
        ref = orc.gaussian_kernel1d(sigma)
        assert n == ref.size
        np.testing.assert_allclose(w[:n], ref, rtol=2e-15, atol=0)


def test_percentile(golden):
    g = golden("percentile")
    a = g["a"]
    assert orc.percentile_linear(a[a > 0], 95) == g["p95_nonzero"]
    assert orc.percentile_linear(a, 99) == g["p99"]
    assert orc.percentile_linear(a, 1) == g["p1"]
    assert orc.percentile_linear(a[a > 0].astype(np.float64), 95) == g["p95_f64"]


@pytest.mark.parametrize("case", ["a", "b", "d", "e", "f"])
def test_projection(golden, case):
    g = golden("projection")
    st = g[case + "_stack"]
    kw = dict(a=dict(axes="TCZYX", reference_channel=0, airyscan=False),
              b=dict(axes="CZYX", reference_channel=1, airyscan=True),
              d=dict(axes="CZYX", reference_channel=0, min_z=0, max_z=9, airyscan=False, atoh_shift=-2),
              e=dict(axes="CZYX", reference_channel=0, airyscan=False),
              f=dict(axes="TCZYX", reference_channel=0, airyscan=False))[case]
    tp = st[None] if kw["axes"].startswith("T") else st
    axes = kw.pop("axes")
    ref_ch = kw.pop("reference_channel")
    proj, zmap = orc.time_point_surface_projection(tp.copy(), axes, ref_ch, z_map=True, **kw)
    assert proj.dtype == np.float64 and zmap.dtype == np.int64
    np.testing.assert_array_equal(zmap, g[case + "_zmap"])
    np.testing.assert_array_equal(proj, g[case + "_proj"])


def test_projection_no_channel_axis_error(golden):
    g = golden("projection")
    assert str(g["c_error"]) == "RuntimeError"
    with pytest.raises(RuntimeError):
        orc.time_point_surface_projection(np.zeros((4, 8, 8), np.uint16), "ZYX", 0, airyscan=False)


def test_block_reduce_and_resize_pieces(golden):
    """P4' building blocks against skimage's own block_reduce(np.mean / np.var) and transform.resize outputs."""
    g = golden("projection_binned")
    vol = g["vol"]
    for b in (2, 3, 7, 10, 16, 20):
        np.testing.assert_array_equal(orc.block_mean(vol, b), g["mean_b%d" % b])
        np.testing.assert_array_equal(orc.block_var(vol, b), g["var_b%d" % b])
    np.testing.assert_array_equal(orc.resize_linear(g["small"], (47, 53)), g["small_resized"])
    np.testing.assert_array_equal(orc.resize_linear(g["small"][:, :1, :2], (9, 11)), g["small_resized_b"])


BINNED_CASES = {"avg10": dict(method="max_averages", bin_size=10), "std4": dict(method="max_std", bin_size=4),
                "multi10": dict(method="multi_channel", bin_size=10),
                "avg7_shift": dict(method="max_averages", bin_size=7, atoh_shift=1)}


@pytest.mark.parametrize("tag", sorted(BINNED_CASES))
def test_projection_binned(golden, tag):
    """sp.py:39-65 (bin_size > 1): z-map and projection from the reference itself."""
    g = golden("projection_binned")
    p, z = orc.time_point_surface_projection(g["g_stack"][None].copy(), "TCZYX", 0, airyscan=False, z_map=True,
                                             **BINNED_CASES[tag])
    np.testing.assert_array_equal(z, g["g_%s_zmap" % tag])
    np.testing.assert_array_equal(p, g["g_%s_proj" % tag])


def test_projection_binned_three_channels_airyscan(golden):
    g = golden("projection_binned")
    p, z = orc.time_point_surface_projection(g["h_stack"].copy(), "CZYX", 2, airyscan=True, z_map=True,
                                             method="multi_channel", bin_size=16)
    np.testing.assert_array_equal(z, g["h_multi16_zmap"])
    np.testing.assert_array_equal(p, g["h_multi16_proj"])
    assert str(g["bad_method_error"]) == "TypeError"
    with pytest.raises(TypeError):
        orc.time_point_surface_projection(g["h_stack"].copy(), "CZYX", 0, airyscan=True, method="nope", bin_size=2)


def test_display_ops(golden, oracle_with_golden_taps):
    """band_pass_filter, set_brightness, save_tiff's normalisation (bim.py:160-188, 233-414) from the reference itself."""
    o = oracle_with_golden_taps
    g = golden("display_ops")
    np.testing.assert_array_equal(o.band_pass_filter(g["bp_f64"], 1.0, 4.0), g["bp_f64_out"])
    np.testing.assert_array_equal(o.band_pass_filter(g["bp_u16"], 2.0, 3.0), g["bp_u16_out"])
    np.testing.assert_array_equal(o.band_pass_filter(g["bp_f32"], 0.5, 2.0), g["bp_f32_out"])
    np.testing.assert_array_equal(o.set_brightness(g["sb_movie"].copy(), "TCYX"), g["sb_bestfit"])
    np.testing.assert_array_equal(o.set_brightness(g["sb_movie"].copy(), "TCYX", method="minMax", clearExtreamPrecentage=0),
                                  g["sb_minmax0"])
    np.testing.assert_array_equal(o.set_brightness(g["sb_u8"].copy(), "YX", clearExtreamPrecentage=5, minVal=20), g["sb_u8_out"])
    np.testing.assert_array_equal(o.set_brightness(g["sb_movie"].copy(), "TCYX", metadata={"min": 150, "max": 30000}),
                                  g["sb_meta_out"])
    np.testing.assert_array_equal(o.tiff_normalise(g["st_in"], "uint16"), g["st_u16"])
    np.testing.assert_array_equal(o.tiff_normalise(g["st_in"], "uint8"), g["st_u8"])
    np.testing.assert_array_equal(o.tiff_normalise(g["bp_u16"], "uint16"), g["st_same"])


def test_rank_filters(golden):
    g = golden("rank_filters")
    lab, img = g["lab"], g["img"]
    cross = np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]])
    np.testing.assert_array_equal(orc.maximum_filter(lab, (5, 5), mode="constant"), g["max5_const"])
    np.testing.assert_array_equal(orc.maximum_filter(lab, (3, 3), mode="constant"), g["max3_const"])
    np.testing.assert_array_equal(orc.maximum_filter(lab, footprint=cross, mode="constant"), g["max_cross_const"])
    np.testing.assert_array_equal(orc.minimum_filter(lab, footprint=cross, mode="constant"), g["min_cross_const"])
    np.testing.assert_array_equal(orc.maximum_filter(img, 7), g["max7_reflect_f64"])
    np.testing.assert_array_equal(orc.maximum_filter(img, 4), g["max4_reflect_f64"])
    b = g["binimg"]
    np.testing.assert_array_equal(orc.dilation(b, 5), g["dil5"])
    np.testing.assert_array_equal(orc.erosion(b, 5), g["ero5"])
    np.testing.assert_array_equal(orc.erosion(b, 7), g["ero7"])
    np.testing.assert_array_equal(orc.dilation(img, 5), g["dil5_gray"])
    np.testing.assert_array_equal(orc.erosion(img, 7), g["ero7_gray"])
    closed = orc.erosion(orc.dilation(b, 5), 5)
    np.testing.assert_array_equal(closed, g["closed_once"])
    # closing is idempotent: 101 iterations == 1; pinned by the golden
    np.testing.assert_array_equal(closed, g["closed_101"])
    for blk in (3, 5):
        np.testing.assert_array_equal(orc.threshold_local_generic_max(img, 0.03, blk), g["thrloc_b%d" % blk])


def test_label(golden):
    g = golden("label")
    np.testing.assert_array_equal(orc.label4(g["bin"], 0)[0], g["label_bg0"])
    np.testing.assert_array_equal(orc.label4(g["img255"], 255)[0], g["label_bg255"])
    np.testing.assert_array_equal(orc.label4(g["multi"], 0)[0], g["label_multi_bg0"])
    np.testing.assert_array_equal(orc.label4((g["bin"] != 0).astype(np.int32), 0)[0], g["ndi_label"])
    np.testing.assert_array_equal(orc.label4(g["snake"], 0)[0], g["label_snake"])


def test_watershed_pieces(golden):
    g = golden("watershed")
    np.testing.assert_array_equal(orc.local_minima(g["i2_blurred"]), g["i2_minima"])
    np.testing.assert_array_equal(orc.label4(g["i2_minima"].astype(np.int32), 0)[0], g["i2_markers"])


@pytest.mark.parametrize("case", ["ii", "iii", "iv", "v", "vi"])
def test_watershed_flood(golden, case):
    g = golden("watershed")
    key = {"vi": "vi_boundary"}.get(case, case + "_img")
    np.testing.assert_array_equal(orc.watershed(g[key]), g[case + "_labels"])


def test_watershed_misc(golden):
    g = golden("watershed")
    np.testing.assert_array_equal(orc.watershed(g["ii_img"], watershed_line=False), g["ii_labels_nowsl"])
    np.testing.assert_array_equal(orc.watershed(np.full((12, 14), 3.5)), g["vii_const_labels"])
    np.testing.assert_array_equal(orc.watershed(g["vii_bowl"]), g["vii_bowl_labels"])


def test_watershed_segmentation(golden):
    g = golden("watershed")
    np.testing.assert_array_equal(orc.watershed_segmentation(g["i_img"], 0.03, 3, 3), g["i_labels"])
    np.testing.assert_array_equal(orc.watershed_segmentation(g["i2_img"], 0.03, 3, 3), g["i2_labels"])
    np.testing.assert_array_equal(orc.watershed_segmentation(g["i2_img"], 0.2, 2, 4), g["i3_labels"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_cellinfo(golden, tag):
    g = golden("cellinfo")
    lab = g[tag + "_labels"]
    rp = orc.regionprops(lab)
    present = rp["area"] > 0
    np.testing.assert_array_equal(rp["label"][present], g[tag + "_rp_label"])
    np.testing.assert_array_equal(rp["area"][present], g[tag + "_rp_area"])
    np.testing.assert_allclose(rp["perimeter"][present], g[tag + "_rp_perimeter"], rtol=1e-13)
    np.testing.assert_array_equal(rp["cy"][present], g[tag + "_rp_centroid-0"])
    np.testing.assert_array_equal(rp["cx"][present], g[tag + "_rp_centroid-1"])
    for k in range(4):
        np.testing.assert_array_equal(rp["bbox"][present, k], g[tag + "_rp_bbox-%d" % k])
    ci = orc.frame_cellinfo(lab)
    np.testing.assert_array_equal(ci["area"], g[tag + "_area"])
    np.testing.assert_array_equal(ci["valid"], g[tag + "_valid"])
    np.testing.assert_array_equal(ci["n_neighbors"], g[tag + "_n_neighbors"])
    nb = g[tag + "_neighbors"]
    for i, s in enumerate(ci["neighbors"]):
        assert sorted(s) == [int(v) for v in nb[i] if v > 0], i
    np.testing.assert_array_equal(orc.contact_matrix(lab, ci["neighbors"]), g[tag + "_contact"])


def test_update_labels(golden):
    g = golden("cellinfo")
    np.testing.assert_array_equal(orc.update_labels(g["u_in"]), g["u_out"])


def test_celltypes_pins(golden):
    g = golden("celltypes")
    lab, inten = g["labels"], g["intensity"]
    rp = orc.regionprops(lab, inten)
    np.testing.assert_allclose(rp["intensity_mean"], g["mean"], rtol=1e-12)
    p10 = np.array([orc.percentile_linear(inten[lab == i], 10) for i in range(1, lab.max() + 1)])
    np.testing.assert_allclose(p10, g["p10"], rtol=1e-15)
    assert orc.percentile_linear(inten, 99) == g["p99"]
    blurred = orc.blur_image(inten, 7)
    lm = np.abs(blurred - orc.maximum_filter(blurred, 7)) < 1e-6
    np.testing.assert_array_equal(lm, g["local_maxima"])


def test_unet_tail(golden):
    g = golden("unet_tail")
    labels, hc, boundary, closed = orc.closing_tail(g["p0"], thr=0.55)
    np.testing.assert_array_equal(closed, g["closed"])
    np.testing.assert_array_equal(hc, g["hc"])
    np.testing.assert_array_equal(boundary, g["boundary"])
    np.testing.assert_array_equal(labels, g["labels"])


def test_tracking(golden):
    g = golden("tracking")
    assert bool(g["ok"])
    labs = g["labels"]
    tables = []
    for f in range(labs.shape[0]):
        ci = orc.frame_cellinfo(labs[f])
        np.testing.assert_array_equal(ci["cx"], g["cx_%d" % f])
        np.testing.assert_array_equal(ci["cy"], g["cy_%d" % f])
        tables.append(ci)
    drifts = np.zeros((3, 2))
    drifts[1] = (0.5, -0.3)
    drifts[2] = (0.5, -0.3)
    ids = orc.track_simple(list(labs), tables, drifts)
    for f in range(labs.shape[0]):
        np.testing.assert_array_equal(ids[f], g["ids_%d" % f])


def test_drift(golden):
    g = golden("drift")
    for tag in "abc":
        imgs = g[tag + "_images"]
        sy, sx = orc.update_drift(imgs[0], imgs[1])
        np.testing.assert_array_equal(np.array([sy, sx]), g[tag + "_drift"])
        np.testing.assert_array_equal(orc.phase_cross_correlation(imgs[0], imgs[1], 100), g[tag + "_calc"])
        np.testing.assert_array_equal(orc.phase_cross_correlation(imgs[0], imgs[1], 1), g[tag + "_calc_whole"])
        np.testing.assert_array_equal(orc.phase_cross_correlation(g[tag + "_prev_f64"], g[tag + "_cur_f64"], 100),
                                      g[tag + "_calc_f64"])


def test_manifold_golden(golden):
    """build_continues_manifold (sp.py:87-165): the oracle's serial C restatement equals the reference on every golden
    case (start in the middle / corner / on an edge, the row-0 wrap quirk on short frames) and through the whole
    projection with build_manifold=True."""
    g = golden("manifold")
    for k in g.files:
        if k.startswith("m_") and k.endswith("_score"):
            n = k[2:-6]
            np.testing.assert_array_equal(orc.build_continues_manifold(g[k]), g["m_%s_z" % n], err_msg=n)
    p, z = orc.time_point_surface_projection(g["p_stack"][None], "TCZYX", 0, airyscan=False, z_map=True, build_manifold=True)
    np.testing.assert_array_equal(z, g["p_zmap"])
    np.testing.assert_array_equal(p, g["p_proj"])
    p, z = orc.time_point_surface_projection(g["p_stack"], "CZYX", 1, min_z=1, max_z=9, airyscan=False, z_map=True,
                                             build_manifold=True, atoh_shift=-1)
    np.testing.assert_array_equal(z, g["p2_zmap"])
    np.testing.assert_array_equal(p, g["p2_proj"])


_BINNED_MANIFOLD = (("avg4", dict(bin_size=4, method="max_averages")), ("std4", dict(bin_size=4, method="max_std")),
                    ("multi10", dict(bin_size=10, method="multi_channel")), ("avg5_shift", dict(bin_size=5, method="max_averages", atoh_shift=1)))


def test_manifold_with_bin_size_golden(golden):
    """build_manifold together with bin_size > 1 (sp.py:56-65): the spiral on the binned score, the plane maps resized to the
    frame by skimage's 2-D bilinear warp and rounded.  The restated warp agrees with skimage's float output to 1e-5 (its affine
    matrix comes out of a least-squares estimate) and exactly for power-of-two factors; np.round of it -- what the reference
    uses -- equals the reference on every case, exact .5 ties included; so do the whole projections."""
    g = golden("manifold_binned")
    for k in range(int(g["r_n"])):
        got = orc.resize_warp2d(g["r%d_in" % k], g["r%d_out" % k].shape)
        np.testing.assert_allclose(got, g["r%d_out" % k], rtol=0, atol=1e-5)
        np.testing.assert_array_equal(np.round(got), np.round(g["r%d_out" % k]))
    np.testing.assert_array_equal(orc.resize_warp2d(g["r0_in"], g["r0_out"].shape), g["r0_out"])
    for name, kw in _BINNED_MANIFOLD:
        p, z = orc.time_point_surface_projection(g["p_stack"][None], "TCZYX", 0, airyscan=False, z_map=True, build_manifold=True, **kw)
        np.testing.assert_array_equal(z, g["p_%s_zmap" % name], err_msg=name)
        np.testing.assert_array_equal(p, g["p_%s_proj" % name], err_msg=name)


def test_display_stretch_golden(golden):
    """The display stretch of gui.py:445-452 (levels from np.percentile, equal levels, 0 / 100): numpy statements evaluated by
    the golden interpreter (gui.py itself needs PyQt5 and a window)."""
    g = golden("display_stretch")
    for k in range(int(g["n"])):
        lo, hi = g["d%d_levels" % k]
        np.testing.assert_array_equal(orc.stretch_for_display(g["d%d_in" % k], lo, hi), g["d%d_out" % k])
