"""GPU: device-resident FramePipeline and the movie driver (one GPU) against the oracle."""
import numpy as np
import pytest

from gpu_util import taps_patch

pytestmark = pytest.mark.gpu


def test_pipeline_and_movie_vs_oracle(monkeypatch, golden_taps, oracle_with_golden_taps):
    orc = oracle_with_golden_taps
    taps_patch(monkeypatch, golden_taps)
    from tissue_image_processing_amd import synthetic, movie
    Z, Y, X, T = 8, 192, 256, 3
    sites_t, is_hc = synthetic.make_movie_sites(Y, X, T, seed=5)
    stacks = [synthetic.make_stack(Z, Y, X, seed=50 + t, sites=sites_t[t], is_hc=is_hc) for t in range(T)]
    backend = movie.GpuFrameBackend(2, Z, Y, X, device=0)
    drifts = np.zeros((T, 2))
    drifts[1:] = (0.5, -0.3)
    tabs, ids = movie.process_movie(T, lambda t: stacks[t], backend, 0, 1, None, "cpu", drifts)
    # oracle: same frames through the CPU restatement
    labs, otabs = [], []
    for t in range(T):
        proj, _ = orc.time_point_surface_projection(stacks[t][None], "TCZYX", 0, airyscan=False, z_map=True)
        lab = orc.watershed_segmentation(proj[0], 0.03, 3, 3)
        labs.append(lab)
        otabs.append(orc.frame_cellinfo(lab))
        got = backend.pipe  # last frame's labels are still in the pipeline buffer
    np.testing.assert_array_equal(backend.labels[T - 1].download((Y, X), np.int32), labs[-1])
    for t in range(T):
        np.testing.assert_array_equal(tabs[t]["area"], otabs[t]["area"])
        np.testing.assert_array_equal(tabs[t]["cx"], otabs[t]["cx"])
        np.testing.assert_array_equal(tabs[t]["cy"], otabs[t]["cy"])
    oids = orc.track_simple(labs, otabs, drifts)
    for t in range(T):
        np.testing.assert_array_equal(ids[t], oids[t])


def test_library_is_reentrant_per_thread():
    """Four host threads, each with its own stream and workspaces (tip_init per thread), run whole frames concurrently --
    what the reference's Qt workers (gui.py:1821-2137) and bench.py's frames in flight rely on.  Every thread must get
    exactly what a single-threaded run gets."""
    import threading
    from tissue_image_processing_amd import _lib, synthetic
    from tissue_image_processing_amd.pipeline import FramePipeline
    Z, Y, X = 10, 256, 384
    stacks = [synthetic.make_stack(Z, Y, X, seed=300 + i) for i in range(4)]

    def run_frame(stack):
        pipe = FramePipeline(2, Z, Y, X, reference_channel=0, airyscan=False)
        d = pipe.upload_stack(stack)
        out = []
        for _ in range(3):                       # several rounds per thread so that the kernels really interleave
            pipe.project(d)
            pipe.segment(0)
            tabs = pipe.cell_tables()
            out.append((pipe.fetch_projection()[1].copy(), pipe.fetch_labels().copy(), tabs["area"].copy(),
                        np.sort(tabs["pairs"].view([("a", np.int32), ("b", np.int32)]).ravel())))
        return out

    serial = [run_frame(s) for s in stacks]
    results, errors = [None] * 4, []

    def worker(i):
        try:
            _lib.init(0)
            results[i] = run_frame(stacks[i])
        except BaseException as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(4):
        for rnd in range(3):
            for a, b in zip(serial[i][rnd], results[i][rnd]):
                np.testing.assert_array_equal(a, b)
