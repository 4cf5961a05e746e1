"""GPU: the movie / large-image drivers and the chunk iterator (sp.py:168-316, bim.py:89-159) against goldens produced by
the REFERENCE's own functions over an in-memory stand-in for its absent reader / writer (tools/make_goldens.py
gold_drivers): projections, z-maps, uint16 TIFF contents, file names, stage records."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(ROOT, "tests", "golden", "drivers.npz"))


@pytest.fixture(autouse=True)
def _golden_taps(monkeypatch, golden_taps):
    from gpu_util import taps_patch
    taps_patch(monkeypatch, golden_taps)          # the taps of the interpreter that produced the goldens


class _NS(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


class FakeAics(object):
    """Duck-typed aicsimageio.AICSImage over arrays, with the OME fields save_stage_positions reads (the same numbers the
    golden generator's stand-in carries)."""

    def __init__(self, scenes):
        self.scenes, self.scene = scenes, 0

    def set_scene(self, i):
        self.scene = i

    @property
    def dims(self):
        t, c, z, y, x = self.scenes[self.scene].shape
        return _NS(T=t, C=c, Z=z, Y=y, X=x)

    def get_image_dask_data(self):
        return self.scenes[self.scene]

    @property
    def metadata(self):
        images = []
        for i, a in enumerate(self.scenes):
            images.append(_NS(name="s%d" % i,
                              stage_label=_NS(x=100.0 * i + 1.5, y=-20.0 - i, z=3.25 + i, x_unit="um", y_unit="um", z_unit="um"),
                              pixels=_NS(size_t=a.shape[0], size_c=a.shape[1], size_z=a.shape[2], dimension_order="XYZCT",
                                         type="uint16", physical_size_x=0.1, physical_size_y=0.1, physical_size_z=0.5,
                                         planes=list(range(a.shape[1] * 3)))))
        return _NS(images=images)


def test_read_image_in_chunks_with_projection(g):
    from tissue_image_processing_amd import basic_image_manipulations as bim, surface_projection as sp
    proj, zmap = np.zeros((2, 2, 1, 40, 56)), np.zeros((2, 1, 1, 40, 56))
    n = sum(1 for _ in bim.read_image_in_chunks(g["ch_stack"], dx=32, dy=24, dt=1, apply_function=sp.time_point_surface_projection,
                                                output=[proj, zmap], axes="TCZYX", reference_channel=0, z_map=True, airyscan=False))
    assert n == int(g["ch_n"])
    np.testing.assert_array_equal(zmap, g["ch_zmap"])
    np.testing.assert_array_equal(proj, g["ch_proj"])
    raw = [c.shape for c in bim.read_image_in_chunks(g["ch_stack"], dx=30, dz=4, dc=1)]
    np.testing.assert_array_equal(np.array(raw), g["ch_raw_shapes"])


def test_large_image_projection(g, tmp_path):
    from tissue_image_processing_amd import basic_image_manipulations as bim, surface_projection as sp
    d = str(tmp_path)
    assert sp.large_image_projection(d, d, "missing.npy") == 0
    np.save(os.path.join(d, "big.npy"), g["li_stack"])
    sp.large_image_projection(d, d, "big.npy", position=1, reference_channel=0, chunk_size=32, method="max_averages")
    img, axes, shape, _ = bim.read_tiff(os.path.join(d, "big_projection.tif"))
    assert axes == str(g["li_axes"]) and img.dtype == np.uint16
    np.testing.assert_array_equal(img, g["li_tif"])
    np.testing.assert_array_equal(np.load(os.path.join(d, "big_zmap.npy")), g["li_zmap"])
    # a list of positions over a two-scene source (T = 2, channel shift -1, reference channel 1)
    src = FakeAics([g["lt_stack"], g["lt_stack"][:, ::-1].copy()])
    open(os.path.join(d, "bigt.czi"), "w").close()
    real_open = bim.open_image
    try:
        bim.open_image = lambda source, series=0: real_open(src if str(source).endswith("bigt.czi") else source, series)
        sp.large_image_projection(d, d, "bigt.czi", position=[1, 2], reference_channel=1, chunk_size=40, method="max_averages",
                                  channels_shift=-1)
    finally:
        bim.open_image = real_open
    for k, key in ((1, "lt_tif1"), (2, "lt_tif2")):
        img, axes, _, _ = bim.read_tiff(os.path.join(d, "bigt_position%d_projection.tif" % k))
        assert axes == str(g["lt_axes"])
        np.testing.assert_array_equal(img, g[key])
    np.testing.assert_array_equal(np.load(os.path.join(d, "bigt_position2_zmap.npy")), g["lt_zmap2"])


def _check_movie_outputs(g, odir, stage=True):
    from tissue_image_processing_amd import basic_image_manipulations as bim
    for k in (1, 2):
        img, axes, _, _ = bim.read_tiff(os.path.join(odir, "x_position%d.tif" % k))
        assert axes == "TCYX" and img.dtype == np.uint16
        np.testing.assert_array_equal(img, g["mv_tif%d" % k])
        z = np.load(os.path.join(odir, "x_zmap_position%d.npy" % k))
        assert z.dtype == np.uint16
        np.testing.assert_array_equal(z, g["mv_zmap%d" % k])
        if stage:
            st = pickle.load(open(os.path.join(odir, "x_stage_locations_position%d.pkl" % k), "rb"))
            np.testing.assert_array_equal(np.array([st["x"], st["y"], st["z"]]), g["mv_stage%d_xyz" % k])
            misc = [st["x_unit"], st["y_unit"], st["z_unit"], repr(st["physical_size_x"]), repr(st["physical_size_y"]),
                    repr(st["physical_size_z"])]
            assert misc == [str(v) for v in g["mv_stage%d_misc" % k]]


def test_movie_surface_projection(g, tmp_path):
    from tissue_image_processing_amd import surface_projection as sp
    odir = str(tmp_path)
    files = [FakeAics([g["mv_m1a"], g["mv_m1b"]]), FakeAics([g["mv_m2b"]])]
    sp.movie_surface_projection(files, 0, (1, 2), 2, odir, "max_averages", 1, False, 0, 0, 0, False, output_name="x_")
    _check_movie_outputs(g, odir)
    # what is left in the output directory: the reference's listing (its stand-in writer creates no .tif) + the two movies;
    # the per-movie .npy intermediates are removed
    assert sorted(f for f in os.listdir(odir) if not f.endswith(".tif")) == [str(v) for v in g["mv_left"]]
    assert sorted(f for f in os.listdir(odir) if f.endswith(".tif")) == ["x_position1.tif", "x_position2.tif"]


def test_drivers_sharded_over_two_processes(g, tmp_path):
    """Time points dealt to two processes (rank t % 2), outputs written by rank 0: identical files."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    odir = str(tmp_path)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_driver_worker.py"), odir], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    _check_movie_outputs(g, odir, stage=False)
    from tissue_image_processing_amd import basic_image_manipulations as bim
    img, axes, _, _ = bim.read_tiff(os.path.join(odir, "bigt_projection.tif"))
    np.testing.assert_array_equal(img, g["lt_tif2"])
    np.testing.assert_array_equal(np.load(os.path.join(odir, "bigt_zmap.npy")), g["lt_zmap2"])
    assert not [f for f in os.listdir(odir) if ".t0" in f]          # the per-time-point parts are gone
