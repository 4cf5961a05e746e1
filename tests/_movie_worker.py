"""Worker for the world_size-2 gloo test of movie.process_movie (CPU; the per-frame compute is the oracle)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OracleBackend(object):
    """Stands in for GpuFrameBackend on CPU: same interface, oracle arithmetic (test infrastructure)."""

    def __init__(self):
        self.labels = {}

    def process_frame(self, t, labels):
        from oracle import oracle as orc
        self.labels[t] = np.ascontiguousarray(labels, np.int32)
        rp = orc.regionprops(labels)
        area = rp["area"]
        return dict(area=area, cy=np.where(area > 0, rp["cy"], 0.0), cx=np.where(area > 0, rp["cx"], 0.0))

    def lookup(self, t, qy, qx):
        from oracle import oracle as orc
        lab = orc.maximum_filter(self.labels[t], (3, 3), mode="constant")
        Y, X = lab.shape
        ok = (qy >= 0) & (qy < Y) & (qx >= 0) & (qx < X)
        out = np.full(qy.shape, -1, np.int32)
        out[ok] = lab[qy[ok], qx[ok]]
        return out


def main():
    import torch.distributed as dist
    from tissue_image_processing_amd import movie
    out_path, n_rep = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "tracking.npz"))
    labs = list(g["labels"])
    frames = (labs + labs[::-1]) * n_rep          # a longer movie out of the golden frames
    drifts = np.zeros((len(frames), 2))
    drifts[1:] = (0.5, -0.3)
    tabs, ids = movie.process_movie(len(frames), lambda t: frames[t], OracleBackend(), rank, world, dist, "cpu", drifts)
    if rank == 0:
        np.savez(out_path, n=len(frames), **{"ids_%d" % t: ids[t] for t in range(len(frames))})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
