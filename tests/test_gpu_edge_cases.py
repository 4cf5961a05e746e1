"""GPU: ragged / degenerate shapes and error paths, against the oracle."""
import numpy as np
import pytest

from gpu_util import taps_patch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def env(monkeypatch, golden_taps, oracle_with_golden_taps):
    from tissue_image_processing_amd import basic_image_manipulations as bim
    from tissue_image_processing_amd import surface_projection as sp
    from tissue_image_processing_amd import _segmentation as seg
    taps_patch(monkeypatch, golden_taps)
    return bim, sp, seg, oracle_with_golden_taps


@pytest.mark.parametrize("shape", [(2, 5, 33, 47), (2, 1, 40, 64), (1, 7, 64, 36), (3, 6, 1, 128), (2, 4, 130, 1),
                                   (2, 9, 96, 132)])
def test_projection_ragged_shapes(env, shape):
    """X not a multiple of 4 takes the generic kernels, Z = 1, single rows / columns, one channel."""
    _, sp, _, orc = env
    rng = np.random.default_rng(sum(shape))
    st = rng.integers(0, 4000, shape).astype(np.uint16)
    st[:, :, : shape[2] // 2, :] //= 3
    p_ref, z_ref = orc.time_point_surface_projection(st.copy(), "CZYX", 0, airyscan=False, z_map=True)
    p, z = sp.time_point_surface_projection(st.copy(), "CZYX", 0, airyscan=False, z_map=True)
    np.testing.assert_array_equal(z, z_ref)
    np.testing.assert_array_equal(p, p_ref)


def test_projection_airyscan_clamps_and_saturates(env):
    _, sp, _, orc = env
    rng = np.random.default_rng(3)
    st = rng.integers(9000, 12000, (2, 6, 40, 48)).astype(np.uint16)
    st[0, 2, 5:9, 5:9] = 65535
    p_ref, z_ref = orc.time_point_surface_projection(st.copy(), "CZYX", 1, airyscan=True, z_map=True)
    p, z = sp.time_point_surface_projection(st.copy(), "CZYX", 1, airyscan=True, z_map=True)
    np.testing.assert_array_equal(z, z_ref)
    np.testing.assert_array_equal(p, p_ref)


@pytest.mark.parametrize("shape", [(1, 50), (50, 1), (33, 70), (3, 3), (64, 64), (65, 97)])
def test_watershed_ragged_shapes(env, shape):
    _, _, seg, orc = env
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    img = rng.random(shape)
    if min(shape) > 8:
        img = orc.blur_image(img, 1.5)
    out = seg.watershed(img)
    np.testing.assert_array_equal(out, orc.watershed(img))


def test_watershed_plateau_minima_and_zero_background(env):
    """Exact-zero plateaus (thresholded background) are marker plateaus; bit-exact like any tie-free flood."""
    bim, _, seg, orc = env
    rng = np.random.default_rng(17)
    img = orc.blur_image(rng.random((120, 150)), 2.0)
    img[img < 0.47] = 0.0
    img = orc.blur_image(img, 1.0)
    img[20:40, 30:80] = 0.0
    out, flags = seg.watershed(img, return_flags=True)
    ref = orc.watershed(img)
    assert int((out != ref).sum()) == 0, flags


def test_label_edge_cases(env):
    _, _, seg, orc = env
    for a in [np.zeros((5, 7), np.int32), np.ones((4, 4), np.int32), np.arange(12, dtype=np.int32).reshape(3, 4) % 2,
              np.ones((1, 9), np.int32), np.ones((9, 1), np.int32)]:
        out, n = seg.label(a, background=0, return_num=True, connectivity=1)
        ref, n_ref = orc.label4(a, 0)
        np.testing.assert_array_equal(out, ref)
        assert n == n_ref


def test_cell_tables_with_gaps_in_labels(env):
    """Label ids that are absent (area 0 rows) keep zero rows, like regionprops_table + make_df in the reference."""
    _, _, seg, orc = env
    lab = np.zeros((40, 50), np.int32)
    lab[2:10, 3:12] = 1
    lab[15:30, 20:45] = 4        # ids 2, 3 absent
    lab[33:38, 5:9] = 6          # id 5 absent
    rp = seg.regionprops_arrays(lab)
    ref = orc.regionprops(lab)
    np.testing.assert_array_equal(rp["area"], ref["area"])
    np.testing.assert_array_equal(rp["bbox"][ref["area"] > 0], ref["bbox"][ref["area"] > 0])
    np.testing.assert_allclose(rp["perimeter"], ref["perimeter"], rtol=1e-13)
    np.testing.assert_array_equal(seg.neighbor_pairs(lab), orc.neighbor_pairs(lab))


def test_argument_errors(env):
    bim, sp, seg, _ = env
    with pytest.raises(TypeError):
        bim.blur_image(np.zeros((4, 4), np.complex64), 1.0)
    with pytest.raises(TypeError):
        bim.watershed_segmentation(np.zeros((8, 8), np.complex64), 0.03, 3, 3)
    with pytest.raises(ValueError):
        seg.watershed(np.zeros((2, 3, 4)))
    with pytest.raises(TypeError):
        sp.time_point_surface_projection(np.zeros((2, 3, 8, 8), np.float64) - 1.0, "CZYX", 0, airyscan=False)
    with pytest.raises(TypeError):
        sp.time_point_surface_projection(np.zeros((2, 3, 8, 8), np.float32) + 0.5, "CZYX", 0, airyscan=False)


def test_blur_image_rank_4_and_5(env):
    """scipy's gaussian_filter takes any rank; so does blur_image: one device pass per axis over the (leading, axis, trailing)
    view, rounding to the array dtype in between.  float32 arrays: bit-identical to this interpreter's scipy; every dtype:
    identical to the rank-3 path applied axis by axis (whose parity with the reference's scipy the goldens pin -- this
    interpreter's newer scipy differs from that one in the last bit of float64 results)."""
    from scipy import ndimage as ndi
    bim = env[0]
    rng = np.random.default_rng(12)
    for shape, sig in (((3, 5, 14, 17), (0.0, 0.6, 1.1, 2.3)), ((2, 3, 4, 9, 11), 1.5), ((4, 6, 8, 10), (1.2, 0.0, 2.9, 0.7))):   # (sigmas outside the golden tap set: scipy builds its own taps here)
        sg = np.broadcast_to(np.asarray(sig, float), (len(shape),))
        for dt in (np.float32, np.float64, np.uint16):
            a = (rng.random(shape) * 3000).astype(dt)
            out = bim.blur_image(a, sig)
            assert out.dtype == a.dtype and out.shape == a.shape
            if dt == np.float32:
                np.testing.assert_array_equal(out, ndi.gaussian_filter(a, sig, mode="nearest"), err_msg=str(shape))
            ref = a
            for ax in range(a.ndim):          # axis by axis through rank-3 calls on moved axes
                if sg[ax] > 0:
                    moved = np.ascontiguousarray(np.moveaxis(ref, ax, -1))
                    flat = bim.blur_image(moved.reshape(-1, 1, moved.shape[-1]), (0.0, 0.0, sg[ax]))
                    ref = np.moveaxis(flat.reshape(moved.shape), -1, ax)
            np.testing.assert_array_equal(out, ref, err_msg=str((shape, dt)))


def test_projection_takes_other_dtypes_holding_uint16_values(env):
    """Upstream casts any stack to float32 (sp.py:26); stacks of other dtypes that hold uint16 values (a uint16 movie that went
    through float32 / float64 / int32) give exactly the uint16 result."""
    _, sp, _, _ = env
    from tissue_image_processing_amd import synthetic
    st = synthetic.make_stack(8, 64, 96, seed=9)
    proj, zmap = sp.time_point_surface_projection(st, "CZYX", 0, airyscan=False, z_map=True)
    for dt in (np.float32, np.float64, np.int32, np.uint32):
        p2, z2 = sp.time_point_surface_projection(st.astype(dt), "CZYX", 0, airyscan=False, z_map=True)
        np.testing.assert_array_equal(z2, zmap)
        np.testing.assert_array_equal(p2, proj)


def test_thread_reentrancy(env):
    """Two Python threads (as the reference's Qt workers) call the library concurrently; results stay bit-exact."""
    import threading
    bim, _, _, orc = env
    rng = np.random.default_rng(5)
    vols = [(rng.random((4, 90, 110)) * 1000).astype(np.float32) for _ in range(4)]
    refs = [orc.blur_image(v, (0.5, 2, 2)) for v in vols]
    outs = [None] * 4

    def work(i):
        from tissue_image_processing_amd import _lib
        _lib.init(0)
        for _ in range(5):
            outs[i] = bim.blur_image(vols[i], (0.5, 2, 2))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for o, r in zip(outs, refs):
        np.testing.assert_array_equal(o, r)


def test_integer_images_follow_scipy_truncation(env):
    """uint16 frames (what the GUI loads from the projection TIFF): scipy keeps the dtype and truncates after every axis."""
    ndi = pytest.importorskip("scipy.ndimage")
    bim, _, seg, _ = env
    rng = np.random.default_rng(12)
    img = rng.integers(0, 4000, (90, 120)).astype(np.uint16)
    from tissue_image_processing_amd import basic_image_manipulations as b
    taps = {}
    np.testing.assert_array_equal(bim.blur_image(img, 3), ndi.gaussian_filter(img, 3, mode="nearest"))
    i16 = (rng.integers(-300, 300, (40, 50))).astype(np.int16)
    np.testing.assert_array_equal(bim.blur_image(i16, (1, 2)), ndi.gaussian_filter(i16, (1, 2), mode="nearest"))
    labels, flags = seg.watershed_segmentation(img, 0.03, 3, 3, return_flags=True)
    assert labels.dtype == np.int32 and labels.max() > 10
    assert flags & 1 and flags & 4        # integer landscape: value ties are reported and flooded by the exact serial replay


def test_cell_tables_with_more_labels_than_tile_slots():
    """64x64 label tiles aggregate per block in 64 / 256 LDS slots; a tile full of 2x2 cells overflows both tables and
    must fall back to the global atomics / hash set without losing anything."""
    from oracle import oracle as orc
    from tissue_image_processing_amd import _segmentation as seg
    Y, X = 198, 301
    yy, xx = np.mgrid[0:Y, 0:X]
    labels = ((yy // 2) * ((X + 1) // 2) + xx // 2 + 1).astype(np.int32)      # every 2x2 block its own label
    labels[(yy % 7 == 0) & (xx % 5 == 0)] = 0                                 # some background holes
    got = seg.regionprops_arrays(labels)
    want = orc.regionprops(labels)
    for k in ("area", "bbox"):
        np.testing.assert_array_equal(got[k], want[k])
    ok = want["area"] > 0
    np.testing.assert_array_equal(got["cy"][ok], want["cy"][ok])
    np.testing.assert_array_equal(got["cx"][ok], want["cx"][ok])
    np.testing.assert_allclose(got["perimeter"], want["perimeter"], rtol=1e-13)
    gp = seg.neighbor_pairs(labels, cap=64 * int(labels.max()))
    wp = orc.neighbor_pairs(labels)
    assert set(map(tuple, np.asarray(gp).tolist())) == set(map(tuple, np.asarray(wp).tolist()))


def test_display_stretch_golden(env, golden):
    """gui.py:445-452: the composite's level stretch with its two percentiles from the device radix select."""
    bim, _, _, _ = env
    g = golden("display_stretch")
    for k in range(int(g["n"])):
        lo, hi = g["d%d_levels" % k]
        out = bim.stretch_for_display(g["d%d_in" % k], lo, hi)
        assert out.dtype == np.float64
        np.testing.assert_array_equal(out, g["d%d_out" % k])


def test_display_ops_golden(env, golden, tmp_path):
    """band_pass_filter, set_brightness (device order statistics + scipy's weighting), save_tiff's
    normalisation and the self-contained TIFF writer, against the reference's own outputs."""
    bim, _, _, _ = env
    g = golden("display_ops")
    np.testing.assert_array_equal(bim.band_pass_filter(g["bp_f64"], 1.0, 4.0), g["bp_f64_out"])
    np.testing.assert_array_equal(bim.band_pass_filter(g["bp_u16"], 2.0, 3.0), g["bp_u16_out"])
    out32 = bim.band_pass_filter(g["bp_f32"], 0.5, 2.0)
    assert out32.dtype == np.float32
    np.testing.assert_array_equal(out32, g["bp_f32_out"])
    np.testing.assert_array_equal(bim.set_brightness(g["sb_movie"].copy(), "TCYX"), g["sb_bestfit"])
    np.testing.assert_array_equal(bim.set_brightness(g["sb_movie"].copy(), "TCYX", method="minMax", clearExtreamPrecentage=0),
                                  g["sb_minmax0"])
    np.testing.assert_array_equal(bim.set_brightness(g["sb_u8"].copy(), "YX", clearExtreamPrecentage=5, minVal=20), g["sb_u8_out"])
    adj, meta = bim.set_brightness(g["sb_movie"].copy(), "TCYX", metadata={"min": 150, "max": 30000, "Ranges": (0, 1, 0, 1)})
    np.testing.assert_array_equal(adj, g["sb_meta_out"])
    assert meta["max"] == int(g["sb_meta_max"]) and meta["min"] == 0 and meta["Ranges"] == (0, 65535, 0, 65535)
    np.testing.assert_array_equal(bim.tiff_normalise(g["st_in"], "uint16"), g["st_u16"])
    np.testing.assert_array_equal(bim.tiff_normalise(g["st_in"], "uint8"), g["st_u8"])
    assert bim.tiff_normalise(g["bp_u16"], "uint16") is not None
    np.testing.assert_array_equal(bim.tiff_normalise(g["bp_u16"], "uint16"), g["st_same"])
    # the writer: pages come back as written (parsed here with struct; tifffile reads the same file in the build container)
    import struct
    path = str(tmp_path / "proj.tif")
    bim.save_tiff(path, g["st_in"], axes="CYX", data_type="uint16")
    raw = open(path, "rb").read()
    assert raw[:4] == b"II*\x00"
    off, pages = struct.unpack("<I", raw[4:8])[0], []
    while off:
        n = struct.unpack("<H", raw[off:off + 2])[0]
        tags = {}
        for i in range(n):
            tag, typ, cnt, val = struct.unpack("<HHII", raw[off + 2 + 12 * i:off + 14 + 12 * i])
            tags[tag] = val
        pages.append(np.frombuffer(raw[tags[273]:tags[273] + tags[279]], "<u2").reshape(tags[257], tags[256]))
        off = struct.unpack("<I", raw[off + 2 + 12 * n:off + 6 + 12 * n])[0]
    np.testing.assert_array_equal(np.stack(pages), g["st_u16"])


def test_blur_image_radius_beyond_the_tap_table(env):
    """sigma > 31.8 (more than 255 taps): the taps travel through device memory, same arithmetic -- equal to the oracle's
    scipy restatement for float32 / float64 / uint16 arrays, along every axis."""
    bim, _, _, orc = env
    rng = np.random.default_rng(40)
    for shape, sig in (((70, 90), 40.0), ((5, 60, 48), (0.0, 33.0, 45.5)), ((3, 300), (0.0, 100.0)), ((40, 6, 7), (50.0, 0.0, 0.0))):
        for dt in (np.float32, np.float64, np.uint16):
            a = (rng.random(shape) * 4000).astype(dt)
            out = bim.blur_image(a, sig)
            assert out.dtype == a.dtype
            if dt == np.uint16:          # scipy keeps the integer dtype: float64 passes, truncated after every axis
                ref = a.astype(np.float64)
                sg = np.broadcast_to(np.asarray(sig, float), (a.ndim,))
                for ax in range(a.ndim):
                    one = np.zeros(a.ndim); one[ax] = sg[ax]
                    if sg[ax] > 0:
                        ref = np.trunc(orc.blur_image(ref, tuple(one)))
                ref = ref.astype(np.uint16)
            else:
                ref = orc.blur_image(a, sig)
            np.testing.assert_array_equal(out, ref, err_msg=str((shape, dt)))
