"""GPU: the frame-sharded movie driver with TWO processes (gloo collectives, both on GPU 0) equals the one-process run;
and the watershed's fallback machinery (wide tile pass + global-minimum commits) gives the same labels as the endgame."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, out):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gpu_movie_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0


def test_world2_equals_world1(tmp_path):
    o1, o2 = str(tmp_path / "w1.npz"), str(tmp_path / "w2.npz")
    _run(1, o1)
    _run(2, o2)
    a, b = np.load(o1), np.load(o2)
    for t in range(int(a["n"])):
        np.testing.assert_array_equal(a["ids_%d" % t], b["ids_%d" % t])
        np.testing.assert_array_equal(a["area_%d" % t], b["area_%d" % t])
        np.testing.assert_array_equal(a["eids_%d" % t], b["eids_%d" % t])
    # drift estimated by each frame's owner from the neighbour rank's plane: identical in both runs, and close to the
    # synthetic movie's global motion (0.5, -0.3) px/frame undone (the sites also random-walk, so only roughly)
    np.testing.assert_array_equal(a["est"], b["est"])
    print("estimated drifts:", a["est"].tolist())
    assert np.all(np.abs(a["est"][1:] - np.array([-0.5, 0.3])) < 1.0)


def test_watershed_fallback_paths_agree(monkeypatch):
    from tissue_image_processing_amd import _segmentation as seg
    rng = np.random.default_rng(4)
    img = rng.random((150, 170))          # white noise: many lines, many stuck pockets
    ref, f0 = seg.watershed(img, return_flags=True)
    monkeypatch.setenv("TIP_WS_NO_ENDGAME", "1")
    out, f1 = seg.watershed(img, return_flags=True)
    monkeypatch.delenv("TIP_WS_NO_ENDGAME")
    np.testing.assert_array_equal(out, ref)
    print("fallback steps with the endgame disabled:", f1 >> 2)


def _groove_image(K=14, H=24, W=41):
    """Two basins left/right of a ridge at the middle column; the ridge column carries a slowly rising GROOVE whose
    flanks rise fast.  The lowest groove pixel becomes a watershed line, every groove pixel above it has that line (or
    a stuck pixel) as its only lower neighbour: a K-cell staircase pocket that is released only when a flank is labelled."""
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    mid = W // 2
    img = 100.0 - np.abs(x - mid) + 0.001 * y + 0.00037 * x
    img[:, mid - 1] = 99.0 + 2.0 * np.arange(H)
    img[:, mid + 1] = 99.0005 + 2.0 * np.arange(H)
    img[:, mid] = 200.0 + np.arange(H)
    img[0, mid] = 99.5
    img[1:K + 1, mid] = 99.5 + 0.1 * np.arange(1, K + 1)
    return img


def test_watershed_global_minimum_fallback(monkeypatch):
    """With the endgame and the wide pass disabled a long stuck pocket is released by committing the pixel with the
    globally smallest pop time, one at a time: slow, always terminates, same labels (and equal to the serial oracle)."""
    from oracle import oracle as orc
    from tissue_image_processing_amd import _segmentation as seg
    img = _groove_image()
    ref = seg.watershed(img)
    np.testing.assert_array_equal(ref, orc.watershed(img))
    monkeypatch.setenv("TIP_WS_NO_ENDGAME", "1")
    out_w, flags_w = seg.watershed(img, return_flags=True)           # wide pass available
    np.testing.assert_array_equal(out_w, ref)
    monkeypatch.setenv("TIP_WS_NO_WIDE", "1")
    out, flags = seg.watershed(img, return_flags=True)                # only the global-minimum commits are left
    np.testing.assert_array_equal(out, ref)
    assert flags >> 2 > 0     # the fallback really ran
    print("global-minimum fallback steps:", flags >> 2, "(wide pass needed", flags_w >> 2, ")")
