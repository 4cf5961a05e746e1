"""GPU parity: HIP Gaussian + surface projection (through the C-ABI) vs golden fixtures and the CPU oracle."""
import ctypes

import numpy as np
import pytest

from gpu_util import taps_patch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def mods(monkeypatch, golden_taps, oracle_with_golden_taps):
    from tissue_image_processing_amd import basic_image_manipulations as bim
    from tissue_image_processing_amd import surface_projection as sp
    taps_patch(monkeypatch, golden_taps)
    return bim, sp, oracle_with_golden_taps


def test_native_library_loaded():
    from tissue_image_processing_amd import _lib
    lib = _lib.lib()
    assert lib.tip_device_count() >= 1
    assert lib.tip_version() >= 100


def test_gaussian_golden_bit_exact(mods, golden):
    bim, _, _ = mods
    g = golden("gaussian")
    vol = g["vol_f32"]
    for tag, sig in [("s05_1_1", (0.5, 1, 1)), ("s05_30_30", (0.5, 30, 30)), ("s1_2_2", (1, 2, 2))]:
        out = bim.blur_image(vol, sig)
        assert out.dtype == np.float32 and out.shape == vol.shape
        np.testing.assert_array_equal(out, g["out_" + tag], err_msg=tag)
    np.testing.assert_array_equal(bim.blur_image(g["img_f64"], 3), g["out2d_s3"])
    np.testing.assert_array_equal(bim.blur_image(g["img_f64"], 7), g["out2d_s7"])
    np.testing.assert_array_equal(bim.blur_image(g["img_f64"].astype(np.float32), 3), g["out2d_f32_s3"])
    np.testing.assert_array_equal(bim.blur_image(g["tiny_f32"], (0.5, 30, 30)), g["tiny_out_s05_30_30"])


@pytest.mark.parametrize("shape,sigma", [((5, 70, 300), (0.5, 30, 30)), ((3, 257, 130), (0, 30, 0)),
                                         ((2, 64, 513), (0, 0, 30)), ((1, 300, 300), (0, 7, 7)),
                                         ((7, 33, 65), (1, 2, 2)), ((1, 1, 1000), (0, 0, 3))])
def test_gaussian_vs_oracle_ragged(mods, shape, sigma):
    bim, _, orc = mods
    rng = np.random.default_rng(sum(shape))
    vol = (rng.random(shape) * 4000).astype(np.float32)
    np.testing.assert_array_equal(bim.blur_image(vol, sigma), orc.blur_image(vol, sigma))
    vol64 = rng.random(shape[1:]) * 100
    np.testing.assert_array_equal(bim.blur_image(vol64, sigma[1:]), orc.blur_image(vol64, sigma[1:]))


def test_long_kernel_equals_generic_kernel(mods):
    """The LDS-tiled long-radius kernel and the generic kernel are the same arithmetic: bit-identical."""
    bim, _, _ = mods
    from tissue_image_processing_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(5)
    Z, Y, X = 3, 300, 333
    vol = (rng.random((Z, Y, X)) * 1000).astype(np.float32)
    taps = bim.gaussian_taps(30.0)
    din = _lib.DeviceBuffer(vol.nbytes).upload(vol)
    outs = []
    for force in (100, 200):
        for axis in (1, 2):
            dout = _lib.DeviceBuffer(vol.nbytes)
            _lib.check(lib.tip_correlate1d_dev(_lib.dptr(din.ptr), _lib.dptr(dout.ptr), 0, Z, Y, X, force + axis,
                                               _lib.ptr(taps), taps.size))
            _lib.check(lib.tip_sync())
            outs.append(dout.download(vol.shape, np.float32))
    np.testing.assert_array_equal(outs[0], outs[2])
    np.testing.assert_array_equal(outs[1], outs[3])


def test_blur_errors(mods):
    bim, _, _ = mods
    with pytest.raises(RuntimeError):
        bim.blur_image(np.zeros((4, 4), np.float32), (1, 1, 1))


@pytest.mark.parametrize("case", ["a", "b", "d", "e", "f"])
def test_projection_golden(mods, golden, case):
    _, sp, _ = mods
    g = golden("projection")
    st = g[case + "_stack"]
    kw = dict(a=dict(axes="TCZYX", reference_channel=0, airyscan=False),
              b=dict(axes="CZYX", reference_channel=1, airyscan=True),
              d=dict(axes="CZYX", reference_channel=0, min_z=0, max_z=9, airyscan=False, atoh_shift=-2),
              e=dict(axes="CZYX", reference_channel=0, airyscan=False),
              f=dict(axes="TCZYX", reference_channel=0, airyscan=False))[case]
    tp = st[None] if kw["axes"].startswith("T") else st
    axes = kw.pop("axes")
    ref = kw.pop("reference_channel")
    proj, zmap = sp.time_point_surface_projection(tp.copy(), axes, ref, z_map=True, **kw)
    assert proj.dtype == np.float64 and zmap.dtype == np.int64
    mism = int((zmap != g[case + "_zmap"]).sum())
    assert mism == 0, "z-map mismatches: %d" % mism
    np.testing.assert_array_equal(proj, g[case + "_proj"])
    # the stated float tolerance (north_star: 1e-5 relative) holds a fortiori
    np.testing.assert_allclose(proj, g[case + "_proj"], rtol=1e-5, atol=0)


BINNED_CASES = {"avg10": dict(method="max_averages", bin_size=10), "std4": dict(method="max_std", bin_size=4),
                "multi10": dict(method="multi_channel", bin_size=10),
                "avg7_shift": dict(method="max_averages", bin_size=7, atoh_shift=1)}


@pytest.mark.parametrize("tag", sorted(BINNED_CASES))
def test_projection_binned_golden(mods, golden, tag):
    """P4' (sp.py:39-65): bin_size > 1 with each score method, against the reference's own outputs."""
    _, sp, _ = mods
    g = golden("projection_binned")
    p, z = sp.time_point_surface_projection(g["g_stack"][None].copy(), "TCZYX", 0, airyscan=False, z_map=True,
                                            **BINNED_CASES[tag])
    assert p.dtype == np.float64 and z.dtype == np.int64
    assert int((z != g["g_%s_zmap" % tag]).sum()) == 0
    np.testing.assert_array_equal(p, g["g_%s_proj" % tag])


def test_projection_binned_three_channels_airyscan_golden(mods, golden):
    _, sp, _ = mods
    g = golden("projection_binned")
    p, z = sp.time_point_surface_projection(g["h_stack"].copy(), "CZYX", 2, airyscan=True, z_map=True,
                                            method="multi_channel", bin_size=16)
    assert int((z != g["h_multi16_zmap"]).sum()) == 0
    np.testing.assert_array_equal(p, g["h_multi16_proj"])
    with pytest.raises(TypeError):      # sp.py:53 raises a str -> TypeError (golden bad_method_error)
        sp.time_point_surface_projection(g["h_stack"].copy(), "CZYX", 0, airyscan=True, method="nope", bin_size=2)


@pytest.mark.parametrize("method,bin_size,shape", [("max_averages", 10, (12, 250, 333)), ("max_std", 3, (7, 129, 131)),
                                                   ("multi_channel", 16, (9, 200, 260)), ("max_std", 128, (5, 300, 140)),
                                                   ("multi_channel", 2, (6, 64, 64))])
def test_projection_binned_vs_oracle(mods, method, bin_size, shape):
    """Ragged shapes and bin sizes (block rows shorter / longer than numpy's 8-way pairwise threshold, frames that are
    not multiples of the bin, a bin larger than half the frame) against the oracle."""
    _, sp, orc = mods
    from tissue_image_processing_amd import synthetic
    Z, Y, X = shape
    st = synthetic.make_stack(Z, Y, X, seed=77 + bin_size)
    p_ref, z_ref = orc.time_point_surface_projection(st.copy(), "CZYX", 1, airyscan=False, z_map=True, method=method,
                                                     bin_size=bin_size)
    p, z = sp.time_point_surface_projection(st.copy(), "CZYX", 1, airyscan=False, z_map=True, method=method,
                                            bin_size=bin_size)
    assert int((z != z_ref).sum()) == 0
    np.testing.assert_array_equal(p, p_ref)


def test_projection_vs_oracle_config1(mods):
    """BASELINE config[0]-like case (512x512, z=10) against the oracle on the same seeded stack."""
    _, sp, orc = mods
    from tissue_image_processing_amd import synthetic
    st = synthetic.make_stack(10, 256, 320, seed=21)
    p_ref, z_ref = orc.time_point_surface_projection(st[None].copy(), "TCZYX", 0, airyscan=False, z_map=True)
    p, z = sp.time_point_surface_projection(st[None].copy(), "TCZYX", 0, airyscan=False, z_map=True)
    assert int((z != z_ref).sum()) == 0
    np.testing.assert_array_equal(p, p_ref)


def test_projection_errors(mods):
    _, sp, _ = mods
    with pytest.raises(RuntimeError):
        sp.time_point_surface_projection(np.zeros((4, 8, 8), np.uint16), "ZYX", 0, airyscan=False)
    st = np.zeros((2, 4, 8, 8), np.uint16)
    with pytest.raises(IndexError):
        sp.time_point_surface_projection(st, "CZYX", 5, airyscan=False)
    # atoh_shift pushing the clipped index to Z (np.clip upper bound is Z, sic) -> IndexError like the reference
    st2 = np.zeros((2, 4, 8, 8), np.uint16)
    st2[0, 3] = 1000
    with pytest.raises(IndexError):
        sp.time_point_surface_projection(st2, "CZYX", 0, airyscan=False, atoh_shift=1)


def test_projection_full_size_properties(mods, monkeypatch):
    """At BASELINE full size (2048x2048x30, C=2), size-independent properties (the bit-for-bit comparison with the
    oracle at this size is test_projection_headline_frame_vs_oracle): z-map range, projection >= 0, bounded by the
    per-pixel z-max of the stack, certified == all-exact argmax, and invariance of the z-map under a global intensity
    scaling of the non-reference channel."""
    _, sp, _ = mods
    from tissue_image_processing_amd import synthetic
    st = synthetic.make_stack(30, 2048, 2048, seed=1)
    proj, zmap = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
    assert proj.shape == (2, 2048, 2048) and zmap.shape == (2048, 2048)
    assert zmap.min() >= 0 and zmap.max() < 30
    assert (proj >= 0).all()
    assert (proj <= st.max(axis=1).astype(np.float64) + 1e-9).all()
    # certified (fast float32 score + exact fix-up) argmax == the all-exact float64 score path, at full size
    monkeypatch.setenv("TIP_PROJECT_EXACT_SCORE", "1")
    proj_e, zmap_e = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
    monkeypatch.delenv("TIP_PROJECT_EXACT_SCORE")
    assert int((zmap != zmap_e).sum()) == 0
    np.testing.assert_array_equal(proj, proj_e)
    st2 = st.copy()
    st2[1] //= 2
    proj2, zmap2 = sp.time_point_surface_projection(st2[None], "TCZYX", 0, airyscan=False, z_map=True)
    np.testing.assert_array_equal(zmap, zmap2)
    np.testing.assert_array_equal(proj[0], proj2[0])


def test_projection_fast_path_equals_generic_path(mods, monkeypatch):
    """Register-sliding / sparse-mask kernels are the same arithmetic as the generic kernels: bit-identical output."""
    _, sp, _ = mods
    from tissue_image_processing_amd import synthetic
    for shape, seed in [((9, 200, 264), 1), ((5, 77, 136), 2), ((30, 128, 512), 3)]:
        st = synthetic.make_stack(*shape, seed=seed)
        st[1, :, :10, :] = 0
        monkeypatch.delenv("TIP_PROJECT_GENERIC", raising=False)
        p_fast, z_fast = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True, atoh_shift=-1)
        monkeypatch.setenv("TIP_PROJECT_GENERIC", "1")
        p_gen, z_gen = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True, atoh_shift=-1)
        monkeypatch.delenv("TIP_PROJECT_GENERIC", raising=False)
        np.testing.assert_array_equal(z_fast, z_gen)
        np.testing.assert_array_equal(p_fast, p_gen)


def test_fused_mask_kernel_equals_separate_kernels(mods, monkeypatch):
    """k_mask_wmax_fused (y pass + x pass of the blurred one-hot mask + weighted z-max in one kernel, the mask volume never
    written) against the separate sparse kernels: bit-identical, with airyscan offset, three channels, a shifted second
    mask, frames that are not multiples of the tile and a rough z-map (random planes: wide z ranges per tile)."""
    _, sp, _ = mods
    from tissue_image_processing_amd import synthetic
    rng = np.random.default_rng(4)
    cases = [(synthetic.make_stack(9, 200, 264, seed=1), dict(airyscan=False, atoh_shift=-1)),
             (synthetic.make_stack(12, 77, 136, seed=2, channels=3, offset=10000), dict(airyscan=True, atoh_shift=0)),
             (synthetic.make_stack(30, 130, 520, seed=3), dict(airyscan=False, atoh_shift=2)),
             (rng.integers(0, 4000, (2, 16, 90, 300)).astype(np.uint16), dict(airyscan=False, atoh_shift=0))]
    for st, kw in cases:
        try:
            monkeypatch.delenv("TIP_PROJECT_UNFUSED_MASK", raising=False)
            p_f, z_f = sp.time_point_surface_projection(st, "CZYX", 0, z_map=True, **kw)
            monkeypatch.setenv("TIP_PROJECT_UNFUSED_MASK", "1")
            p_s, z_s = sp.time_point_surface_projection(st, "CZYX", 0, z_map=True, **kw)
        except IndexError:
            continue      # (a shifted plane fell off the stack: the reference's IndexError, both paths)
        finally:
            monkeypatch.delenv("TIP_PROJECT_UNFUSED_MASK", raising=False)
        np.testing.assert_array_equal(z_f, z_s)
        np.testing.assert_array_equal(p_f, p_s)


def test_fused_preblur_kernel_equals_separate_kernels(mods, monkeypatch):
    """k_preblur_fused (uint16 -> z 0.5 -> y 1 -> x 1 -> z 0.5 in one kernel) against the four separate register-sliding
    kernels: identical z-maps and projections, incl. few planes (Z = 1, 2, 3), ragged frames, the airyscan offset."""
    _, sp, _ = mods
    from tissue_image_processing_amd import synthetic
    rng = np.random.default_rng(6)
    cases = [(synthetic.make_stack(9, 200, 264, seed=1), dict(airyscan=False)),
             (synthetic.make_stack(12, 77, 136, seed=2, channels=3, offset=10000), dict(airyscan=True)),
             (synthetic.make_stack(30, 130, 520, seed=3), dict(airyscan=False, atoh_shift=-2)),
             (rng.integers(0, 4000, (2, 1, 40, 132)).astype(np.uint16), dict(airyscan=False)),
             (rng.integers(0, 4000, (2, 2, 33, 260)).astype(np.uint16), dict(airyscan=False)),
             (rng.integers(0, 4000, (1, 3, 64, 128)).astype(np.uint16), dict(airyscan=False)),
             (rng.integers(0, 60000, (2, 5, 31, 12)).astype(np.uint16), dict(airyscan=True))]
    for st, kw in cases:
        monkeypatch.delenv("TIP_PROJECT_UNFUSED_PREBLUR", raising=False)
        p_f, z_f = sp.time_point_surface_projection(st, "CZYX", 0, z_map=True, **kw)
        monkeypatch.setenv("TIP_PROJECT_UNFUSED_PREBLUR", "1")
        p_s, z_s = sp.time_point_surface_projection(st, "CZYX", 0, z_map=True, **kw)
        monkeypatch.delenv("TIP_PROJECT_UNFUSED_PREBLUR", raising=False)
        np.testing.assert_array_equal(z_f, z_s)
        np.testing.assert_array_equal(p_f, p_s)


def test_certified_argmax_equals_exact_score_path(mods, monkeypatch):
    """The fast float32 score passes + certification + exact fix-up give the same z-map as the exact float64 passes,
    including on data built to make neighbouring planes nearly tie."""
    _, sp, _ = mods
    from tissue_image_processing_amd import synthetic
    rng = np.random.default_rng(9)
    cases = [synthetic.make_stack(12, 300, 328, seed=11), synthetic.make_stack(30, 256, 256, seed=12)]
    tie = np.zeros((2, 6, 160, 200), np.uint16)            # identical planes -> exact ties everywhere
    tie[:, :, :, :] = rng.integers(50, 4000, (1, 1, 160, 200)).astype(np.uint16)
    tie[0, 3, 80:, :] += 1                                  # and a one-count edge on plane 3
    cases.append(tie)
    for st in cases:
        monkeypatch.delenv("TIP_PROJECT_EXACT_SCORE", raising=False)
        p_c, z_c = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
        monkeypatch.setenv("TIP_PROJECT_EXACT_SCORE", "1")
        p_e, z_e = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
        monkeypatch.delenv("TIP_PROJECT_EXACT_SCORE", raising=False)
        assert int((z_c != z_e).sum()) == 0
        np.testing.assert_array_equal(p_c, p_e)


def test_baseline_config_sizes_vs_oracle(mods):
    """BASELINE.json configs[0] (512x512, z=10) and configs[1] (1024x1024, z=20): projection + Gaussian on the GPU
    diffed against the CPU oracle -- required tolerance 1e-5 relative, achieved: bit-identical."""
    bim, sp, orc = mods
    from tissue_image_processing_amd import synthetic
    for (Z, Y, X, seed) in [(10, 512, 512, 31), (20, 1024, 1024, 32)]:
        st = synthetic.make_stack(Z, Y, X, seed=seed)
        p_ref, z_ref = orc.time_point_surface_projection(st[None].copy(), "TCZYX", 0, airyscan=False, z_map=True)
        p, z = sp.time_point_surface_projection(st[None].copy(), "TCZYX", 0, airyscan=False, z_map=True)
        assert int((z != z_ref).sum()) == 0
        np.testing.assert_allclose(p, p_ref, rtol=1e-5, atol=0)
        np.testing.assert_array_equal(p, p_ref)
        vol = st[0].astype(np.float32)
        np.testing.assert_array_equal(bim.blur_image(vol, (0.5, 1, 1)), orc.blur_image(vol, (0.5, 1, 1)))


def test_projection_headline_frame_vs_oracle(mods):
    """BASELINE's headline frame (2048x2048x30, C=2), projection and z-map bit-identical to the oracle's CPU path at
    FULL size (the C correlate kernels make that about half a minute of one host core)."""
    _, sp, _ = mods
    from oracle import oracle as orc
    from tissue_image_processing_amd import synthetic
    st = synthetic.make_stack(30, 2048, 2048, seed=100)        # bench.py's frame
    proj, zmap = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
    rproj, rzmap = orc.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
    assert int((zmap != rzmap).sum()) == 0
    np.testing.assert_array_equal(proj, rproj)


def test_every_fast_pass_variant_gives_the_same_zmap(mods, monkeypatch):
    """The approximate sigma-30 score comes from the matrix-core kernels by default (k_corr_long_mfma for y,
    k_corr_long_mfma2 for x); the tuning hook selects the other variants (both MFMA kernels on both axes, the packed and
    the scalar VALU kernels).  Certification makes the z-map independent of which variant produced the score: all equal
    the all-exact float64 path, on a frame whose extents are not multiples of the tiles (ragged edge tiles, more tiles
    than persistent blocks)."""
    _, sp, _ = mods
    from tissue_image_processing_amd import synthetic
    for shape, seed in [((12, 515, 777), 77), ((30, 1100, 1300), 78)]:
        st = synthetic.make_stack(*shape, seed=seed)
        monkeypatch.setenv("TIP_PROJECT_EXACT_SCORE", "1")
        proj_e, zmap_e = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
        monkeypatch.delenv("TIP_PROJECT_EXACT_SCORE")
        for cfg in (None, "3,3", "4,4", "4,3", "11616,11616", "1616,1616"):
            if cfg:
                monkeypatch.setenv("TIP_FAST_CFG", cfg)
            proj, zmap = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
            monkeypatch.delenv("TIP_FAST_CFG", raising=False)
            assert int((zmap != zmap_e).sum()) == 0, cfg
            np.testing.assert_array_equal(proj, proj_e)
