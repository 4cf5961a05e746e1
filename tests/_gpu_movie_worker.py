"""Worker for the 2-process GPU movie test: both ranks drive the SAME GPU (device 0), collectives over gloo."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from tissue_image_processing_amd import movie, synthetic
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    Z, Y, X, T = 6, 128, 160, 4
    sites_t, is_hc = synthetic.make_movie_sites(Y, X, T, seed=7)
    stacks = [synthetic.make_stack(Z, Y, X, seed=70 + t, sites=sites_t[t], is_hc=is_hc) for t in range(T)]
    backend = movie.GpuFrameBackend(2, Z, Y, X, device=0, keep_planes=True)
    drifts = np.zeros((T, 2))
    drifts[1:] = (-0.5, 0.3)      # registering frame t onto t-1 undoes the sites' (0.5, -0.3) px/frame motion
    d = dist if world > 1 else None
    tabs, ids = movie.process_movie(T, lambda t: stacks[t], backend, rank, world, d, "cpu", drifts)
    # the same movie with the drift ESTIMATED inside the sharded driver (planes exchanged between the ranks)
    backend2 = movie.GpuFrameBackend(2, Z, Y, X, device=0, keep_planes=True, inflight=2)   # two frames in flight per process
    tabs_e, ids_e = movie.process_movie(T, lambda t: stacks[t], backend2, rank, world, d, "cpu", estimate_drift=True,
                                        block_frames=1)      # rounds of one frame per rank: compute of the next round overlaps the exchange
    if rank == 0:
        np.savez(out_path, n=T, **{"ids_%d" % t: ids[t] for t in range(T)}, **{"area_%d" % t: tabs[t]["area"] for t in range(T)},
                 **{"eids_%d" % t: ids_e[t] for t in range(T)}, est=np.array([tb["drift"] for tb in tabs_e]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
