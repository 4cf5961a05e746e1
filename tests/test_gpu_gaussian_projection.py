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
        for cfg in (None, "5,5", "3,4", "3,3", "4,4", "4,3", "5,4", "11616,11616", "1616,1616"):      # (5: the fp16 tiles, the default)
            if cfg:
                monkeypatch.setenv("TIP_FAST_CFG", cfg)
            proj, zmap = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
            monkeypatch.delenv("TIP_FAST_CFG", raising=False)
            assert int((zmap != zmap_e).sum()) == 0, cfg
            np.testing.assert_array_equal(proj, proj_e)


def _gauss30():
    from oracle import oracle as orc
    return np.asarray(orc.gaussian_kernel1d(30.0), dtype=np.float64)


def _exact_pass(vol, w, axis):
    """scipy's correlate1d(mode='nearest') of a float32 volume in float64 (not rounded): the real-number sum the bound refers to"""
    r = len(w) // 2
    pad = [(0, 0)] * 3
    pad[axis] = (r, r)
    v = np.pad(vol.astype(np.float64), pad, mode="edge")
    out = np.zeros(vol.shape, np.float64)
    n = vol.shape[axis]
    for k in range(len(w)):
        sl = [slice(None)] * 3
        sl[axis] = slice(k, k + n)
        out += w[k] * v[tuple(sl)]
    return out


def test_mfma_f16_rounding_structure_on_this_device():
    """What the certified bound of csrc/tip_corr_f16.h counts: one v_mfma_f32_32x32x16_f16 adds its sixteen products to the
    accumulator as TWO exactly-summed halves of eight (k 0..7, k 8..15), each rounded once (to nearest, ties to even) -- so a term
    passes through two roundings per instruction, not sixteen.  Probed with exact power-of-two products (tools/ubench/
    mfma_f16_rounding.hip prints the same cases); a device that rounds elsewhere fails here instead of silently voiding the bound."""
    import ctypes
    from tissue_image_processing_amd import _lib
    lib = _lib.lib()
    cases = []

    def case(c, exps):                    # product k = 2^-exps[k] (None: no product)
        a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
        for k, e in enumerate(exps):
            if e is not None:
                a[k], b[k] = 2.0 ** -(e // 2), 2.0 ** -(e - e // 2)
        cases.append((a, b, np.float32(c)))

    case(1.0, [25] * 16)                                   # 0: exact sum 4 ulp
    case(1.0, [26] * 16)                                   # 1: 2 ulp (two halves of 1 ulp each)
    case(1.0, [27] * 16)                                   # 2: each half is HALF an ulp: ties to even -> 1 (one exact sum of 16 would give +1 ulp)
    case(1.0, [26] * 8 + [None] * 8)                       # 3: a half of eight is summed exactly: +1 ulp (groups of four would tie away)
    case(1.0, [None] * 8 + [26] * 8)                       # 4
    case(1.0, [26 if k % 2 == 0 else None for k in range(16)])    # 5: four per half = half an ulp each -> 1
    case(1.0, [25, 25, 25, 25] + [None] * 12)              # 6: +1 ulp
    case(1.0, [26] * 6 + [None] * 10)                      # 7: 0.75 ulp in one half: +1 ulp to nearest (0 if the adder truncated)
    # 8, 9: half an ulp plus a little (1/128, 1/16384 of an ulp): +1 ulp if the half-sum is exact before its one rounding
    for extra in (3, 10):
        a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
        a[:8], b[:8] = 2.0 ** -13, 2.0 ** -14
        a[0] = 2.0 ** -13 * (1 + 2.0 ** -extra)
        cases.append((a, b, np.float32(1.0)))
    a = np.stack([c[0] for c in cases]).astype(np.float32)
    b = np.stack([c[1] for c in cases]).astype(np.float32)
    cv = np.asarray([c[2] for c in cases], np.float32)
    out = np.zeros(len(cases), np.float32)
    _lib.check(lib.tip_mfma_f16_probe(_lib.ptr(a), _lib.ptr(b), _lib.ptr(cv), _lib.ptr(out), len(cases)))
    ulps = (out.astype(np.float64) - 1.0) / 2.0 ** -23
    print("mfma f16 probe, ulps above 1:", ulps.tolist())
    assert ulps[:7].tolist() == [4.0, 2.0, 0.0, 1.0, 1.0, 0.0, 1.0]
    assert ulps[7] in (0.0, 1.0)          # (either rounding mode of the half-sums is inside the 2 u the bound takes per rounding)
    assert ulps[8] in (0.0, 1.0) and ulps[9] in (0.0, 1.0)


@pytest.mark.parametrize("axis", [1, 2])
def test_f16_score_pass_error_bound(axis):
    """One sigma-30 pass on the fp16 matrix cores against the float64 sum: relative error below the 38.2 u the certification
    assumes (measured: a few u), on data spanning the clip range, on a black-background volume whose blurred fringes decay through
    thirty orders of magnitude (22 bits relative down to 2^-36 of the clip value, the absolute floor below), ragged extents;
    a zero score means zero inputs."""
    import ctypes
    from tissue_image_processing_amd import _lib
    lib = _lib.lib()
    w = _gauss30()
    assert len(w) == 241
    rng = np.random.default_rng(5 + axis)
    u = 2.0 ** -24
    for shape, kind in (((2, 300, 333), "dense"), ((1, 520, 290), "fringes"), ((3, 64, 40), "small")):
        clip = np.float32(3187.25)
        if kind == "fringes":
            vol = np.zeros(shape, np.float32)
            vol[0, 255:259, 140:150] = clip                      # an isolated blob: its tails are all there is
            vol[0, 10, 5] = np.float32(1e-20)                    # far below the floor, but not zero
            vol = _exact_pass(vol, w, 1 if axis == 2 else 2).astype(np.float32)     # a first pass: tails of every magnitude
        else:
            vol = (rng.random(shape) ** 3 * clip).astype(np.float32)
            vol[rng.random(shape) < 0.2] = 0
        out = np.empty_like(vol)
        flag = ctypes.c_int(0)
        _lib.check(lib.tip_score_pass_f16(_lib.ptr(vol), _lib.ptr(out), shape[0], shape[1], shape[2], axis, _lib.ptr(w), 241,
                                          ctypes.c_float(float(clip)), ctypes.byref(flag)))
        assert flag.value == 0
        ref = _exact_pass(vol, w, axis)
        err = np.abs(out.astype(np.float64) - ref)
        floor = float(clip) * 2.0 ** -47
        rel = np.where(ref > 0, (err - floor).clip(0) / np.where(ref > 0, ref, 1), 0)
        print("f16 score pass axis %d %s %s: max relative error %.2f u (beyond the absolute floor), max |error| %.3g"
              % (axis, kind, shape, rel.max() / u, err.max()))
        assert rel.max() < 38.2 * u
        assert np.array_equal(out == 0, ref == 0)                # zero score <=> zero inputs
    # a sample beyond the clip's range raises the flag instead of silently saturating
    bad = np.full((1, 64, 64), 4.0 * 3187.25, np.float32)
    flag = ctypes.c_int(0)
    _lib.check(lib.tip_score_pass_f16(_lib.ptr(bad), _lib.ptr(np.empty_like(bad)), 1, 64, 64, axis, _lib.ptr(w), 241, ctypes.c_float(3187.25),
                                      ctypes.byref(flag)))
    assert flag.value & 8


def test_black_background_zmap_is_exact(mods, monkeypatch):
    """airyscan=True subtracts 10000 and clamps: large exactly-zero regions with blurred fringes of every magnitude around the signal.
    The fp16 score tiles keep the certification meaningful there (relative accuracy down to 2^-36 of the clip value, zero scores only
    from zero inputs): the z-map equals the all-exact path's."""
    _, sp, _ = mods
    rng = np.random.default_rng(12)
    Z, Y, X = 9, 600, 700
    st = np.full((2, Z, Y, X), 9000, np.uint16)                  # below the airyscan offset: zero after the subtraction
    for _ in range(40):                                          # sparse bright blobs at random planes
        z, y, x = int(rng.integers(0, Z)), int(rng.integers(0, Y - 12)), int(rng.integers(0, X - 12))
        st[:, z, y:y + 12, x:x + 12] = 10000 + rng.integers(50, 4000, (2, 12, 12))
        if z + 1 < Z:
            st[:, z + 1, y:y + 12, x:x + 12] = 10000 + rng.integers(50, 4000, (2, 12, 12))
    monkeypatch.setenv("TIP_PROJECT_EXACT_SCORE", "1")
    proj_e, zmap_e = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=True, z_map=True)
    monkeypatch.delenv("TIP_PROJECT_EXACT_SCORE")
    proj, zmap = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=True, z_map=True)
    assert int((zmap != zmap_e).sum()) == 0
    np.testing.assert_array_equal(proj, proj_e)
