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
