"""Two-valued flood on a headline-size U-Net tail image: generation sizes (TIP_WS_DEBUG) and time per frame of the watershed call for a few
settings of the one-workgroup threshold (TIP_MB_SMALL) / batch size (TIP_MB_BATCH).  `gpurun -- python tools/mb_gens.py`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from tissue_image_processing_amd import _lib, _segmentation as seg
from test_gpu_segmentation import boundary_image

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
img = np.ascontiguousarray(boundary_image(N, 4), np.float64)
print("boundary image %d^2, %.0f%% markers" % (N, 100 * float((img == 0).mean())), flush=True)
ref = None
for small, batch in [("0", "8"), ("-1", "-1"), ("2048", "4"), ("8192", "3"), ("8192", "2"), ("32768", "2"), ("32768", "3")]:
    with _lib.tuning(TIP_MB_SMALL=small, TIP_MB_BATCH=batch, TIP_WS_DEBUG="1" if ref is None or small == "-1" else "0"):
        out = seg.watershed(img)
        if ref is None:
            ref = out
        assert (out == ref).all()
    with _lib.tuning(TIP_MB_SMALL=small, TIP_MB_BATCH=batch):
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            seg.watershed(img)
            ts.append(time.perf_counter() - t0)
    print("TIP_MB_SMALL=%s TIP_MB_BATCH=%s: %.2f ms per call (min of 7, host copies included)" % (small, batch, 1e3 * min(ts)), flush=True)
