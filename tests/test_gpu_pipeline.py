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


def test_config5_frame_4096x4096x60():
    """BASELINE config 5's frame size on one GPU (the frame and its workspaces fit HBM many times over): certified z-map ==
    exact-score z-map at full size, projection and labels of a crop-sized sub-problem against the oracle, and
    size-independent properties of the full-size outputs (label map vs its own cell tables)."""
    import os
    from oracle import oracle as orc
    from tissue_image_processing_amd import synthetic, _lib
    from tissue_image_processing_amd.pipeline import FramePipeline
    Z, N, B = 60, 4096, 1024
    base = synthetic.make_stack(Z, B, B, seed=55)                 # (2, Z, 1024, 1024); mirrored 4 x 4 -> seamless 4096^2
    row = np.concatenate([base, base[..., ::-1], base, base[..., ::-1]], axis=3)
    st = np.concatenate([row, row[:, :, ::-1, :], row, row[:, :, ::-1, :]], axis=2)
    del row
    assert st.shape == (2, Z, N, N)
    pipe = FramePipeline(2, Z, N, N, reference_channel=0, airyscan=False)
    d = pipe.upload_stack(st)
    pipe.project(d)
    proj, zmap = pipe.fetch_projection()
    os.environ["TIP_PROJECT_EXACT_SCORE"] = "1"
    try:
        pipe.project(d)
        proj_x, zmap_x = pipe.fetch_projection()
    finally:
        del os.environ["TIP_PROJECT_EXACT_SCORE"]
    assert int((zmap != zmap_x).sum()) == 0
    np.testing.assert_array_equal(proj, proj_x)
    assert zmap.min() >= 0 and zmap.max() < Z
    # the mirrored construction makes the frame symmetric: so must be the outputs (filters use edge replication, which
    # commutes with the mirror; argmax and z-max are pointwise)
    np.testing.assert_array_equal(zmap[:, :2 * B], zmap[:, 4 * B - 1:2 * B - 1:-1])
    np.testing.assert_array_equal(proj[0][:2 * B, :], proj[0][4 * B - 1:2 * B - 1:-1, :])
    # classical segmentation + tables at full size: consistency of the label map with its own tables
    pipe.project(d)
    pipe.segment(0)
    tabs = pipe.cell_tables()
    lab = pipe.fetch_labels()
    n = int(lab.max())
    assert n == tabs["area"].size and n > 10000
    np.testing.assert_array_equal(np.bincount(lab.ravel(), minlength=n + 1)[1:], tabs["area"])
    assert int((lab == 0).sum()) + int(tabs["area"].sum()) == N * N
    # a crop of the projection through the oracle's watershed_segmentation == the device result on the same crop (a crop
    # inside one mirror image: across a mirror line the landscape has exact value ties between non-marker pixels)
    from tissue_image_processing_amd import _segmentation as seg
    crop = np.ascontiguousarray(proj[0][1100:1700, 1200:1900])
    np.testing.assert_array_equal(seg.watershed_segmentation(crop, 0.03, 3, 3), orc.watershed_segmentation(crop, 0.03, 3, 3))
    d.free()


@pytest.mark.parametrize("shape,grid", [((8, 700, 900), (2, 3)), ((5, 300, 1100), (1, 4)), ((6, 512, 512), (2, 2))])
def test_tiled_projection_equals_untiled(shape, grid):
    """Config 5's spatial tiling (tiling.py): tiles + 132-pixel halo + the frame's summed percentile histogram give the
    untiled projection and z-map bit for bit (one process walks all tiles here; the 2-rank exchange is the gloo test)."""
    from tissue_image_processing_amd import synthetic, tiling, surface_projection as sp
    Z, Y, X = shape
    st = synthetic.make_stack(Z, Y, X, seed=91)
    proj, zmap = sp.time_point_surface_projection(st, "CZYX", 0, airyscan=False, z_map=True)
    backend = tiling.GpuTileBackend(reference_channel=0, airyscan=False)
    tp, tz = tiling.project_tiled(lambda a, b, c, d: st[:, :, a:b, c:d], 2, Y, X, grid, backend)
    assert int((tz != zmap).sum()) == 0
    np.testing.assert_array_equal(tp, proj)
    p3, z3, lab = tiling.process_tiled_frame(lambda a, b, c, d: st[:, :, a:b, c:d], 2, Y, X, grid, backend)
    from tissue_image_processing_amd import basic_image_manipulations as bim
    np.testing.assert_array_equal(lab, bim.watershed_segmentation(proj[0], 0.03, 3, 3))


def test_movie_tracks_vs_oracle_at_512_with_rounds(monkeypatch, golden_taps, oracle_with_golden_taps):
    """The movie path at a larger size, worked off in rounds (compute of round k+1 overlaps the exchange of round k), four
    frames in flight: track ids equal the oracle tracker's on the same frames, and the track-length statistics say what the
    bench's movie numbers mean.  The synthetic frames are noisy (Poisson 100 on a 3000-count membrane), so the classical
    segmentation splits a Voronoi cell into ~2.5 labels whose fragments are not persistent from frame to frame: many short
    tracks are a property of this data under the reference's tracker (the oracle gives the same ids), not of the sharding."""
    orc = oracle_with_golden_taps
    taps_patch(monkeypatch, golden_taps)
    from tissue_image_processing_amd import synthetic, movie
    Z, Y, X, T = 8, 512, 512, 4
    sites_t, is_hc = synthetic.make_movie_sites(Y, X, T, seed=9)
    stacks = [synthetic.make_stack(Z, Y, X, seed=90 + t, sites=sites_t[t], is_hc=is_hc) for t in range(T)]
    backend = movie.GpuFrameBackend(2, Z, Y, X, device=0, inflight=2)
    drifts = np.zeros((T, 2))
    drifts[1:] = (-0.5, 0.3)
    tabs, ids = movie.process_movie(T, lambda t: stacks[t], backend, 0, 1, None, "cpu", drifts, block_frames=2)
    labs, otabs = [], []
    for t in range(T):
        proj, _ = orc.time_point_surface_projection(stacks[t][None], "TCZYX", 0, airyscan=False, z_map=True)
        lab = orc.watershed_segmentation(proj[0], 0.03, 3, 3)
        labs.append(lab)
        otabs.append(orc.frame_cellinfo(lab))
    oids = orc.track_simple(labs, otabs, drifts)
    for t in range(T):
        np.testing.assert_array_equal(tabs[t]["area"], otabs[t]["area"])
        np.testing.assert_array_equal(ids[t], oids[t])
    n_sites = sites_t[0].shape[0]
    all_ids = np.concatenate(ids)
    lengths = np.bincount(np.unique(all_ids, return_counts=True)[1], minlength=T + 1)[1:]
    carried = [int(np.isin(ids[t], ids[t - 1]).sum()) for t in range(1, T)]
    print("512^2 x %d frames: %d sites, labels per frame %s (%.1f per site), ids carried over %s, tracks %d, length histogram 1..%d: %s"
          % (T, n_sites, [len(i) for i in ids], len(ids[0]) / n_sites, carried, all_ids.max(), T, lengths.tolist()))
    assert all_ids.max() < sum(len(i) for i in ids)          # some tracks do continue


def test_movie_rounds_do_not_leak_device_memory():
    """process_movie calls the backend once per round; the backend's worker threads, their library contexts (stream, workspace
    pool) and pipeline buffers persist across rounds and are released by close(): free device memory stays flat over many rounds
    (round 3 created `inflight` fresh contexts per round and never released them)."""
    import torch
    from tissue_image_processing_amd import synthetic, movie
    Z, Y, X = 6, 256, 256
    st = synthetic.make_stack(Z, Y, X, seed=3)
    torch.cuda.synchronize()
    free_start = torch.cuda.mem_get_info(0)[0]
    backend = movie.GpuFrameBackend(2, Z, Y, X, device=0, inflight=3)
    frees = []
    for rnd in range(24):
        out = backend.process_frames([3 * rnd, 3 * rnd + 1, 3 * rnd + 2], lambda t: st)
        assert sorted(out) == [3 * rnd, 3 * rnd + 1, 3 * rnd + 2]
        for t in list(backend.labels):
            backend.labels.pop(t).free()
        torch.cuda.synchronize()
        frees.append(torch.cuda.mem_get_info(0)[0])
    assert len(backend._workers) == 3
    # after the first rounds have sized the workspace pools nothing grows any more
    assert max(frees[4:]) - min(frees[4:]) <= 8 << 20, [f >> 20 for f in frees]
    backend.close()
    torch.cuda.synchronize()
    assert free_start - torch.cuda.mem_get_info(0)[0] <= 64 << 20        # the contexts' pools and the pipelines' buffers are back
