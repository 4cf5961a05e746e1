"""Worker of test_gpu_drivers.py: movie_surface_projection with the time points sharded over WORLD_SIZE processes
(gloo barrier; every process projects on GPU 0)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from tissue_image_processing_amd import surface_projection as sp
    g = np.load(os.path.join(ROOT, "tests", "golden", "drivers.npz"))
    odir = sys.argv[1]
    sp.movie_surface_projection([[g["mv_m1a"], g["mv_m1b"]], [g["mv_m2b"]]], 0, (1, 2), 2, odir, "max_averages", 1, False, 0, 0, 0,
                                False, output_name="x_", rank=rank, world=world)
    big = [g["lt_stack"], g["lt_stack"][:, ::-1].copy()]
    np.save(os.path.join(odir, "bigt.npy"), big[1]) if rank == 0 else None
    if world > 1:
        dist.barrier()
    sp.large_image_projection(odir, odir, "bigt.npy", position=1, reference_channel=1, chunk_size=40, method="max_averages",
                              channels_shift=-1, rank=rank, world=world)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
