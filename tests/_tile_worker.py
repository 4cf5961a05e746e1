"""Worker for the world_size-2 gloo test of tiling.process_tiled_frame (CPU; the per-tile compute is the oracle)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OracleTileBackend(object):
    """Stands in for GpuTileBackend on CPU: same interface, oracle arithmetic (test infrastructure)."""

    def __init__(self, reference_channel=0):
        self.ref = reference_channel
        self.tiles = {}

    def histogram(self, key, tile_u16, box):
        self.tiles[key] = np.asarray(tile_u16)
        y0, y1, x0, x1 = box
        return np.bincount(self.tiles[key][self.ref][:, y0:y1, x0:x1].ravel(), minlength=65536).astype(np.uint64)

    def project(self, key, hist):
        from oracle import oracle as orc
        tile = self.tiles.pop(key)
        # the oracle takes the clip source as an array: any array with the frame's histogram has the frame's percentile
        frame_values = np.repeat(np.arange(65536, dtype=np.float32), hist.astype(np.int64))
        return orc.time_point_surface_projection(tile, "CZYX", self.ref, airyscan=False, z_map=True, clip_from=frame_values)


def test_stack():
    from tissue_image_processing_amd import synthetic
    return synthetic.make_stack(6, 420, 500, seed=17)


def main():
    import torch.distributed as dist
    from tissue_image_processing_amd import tiling
    from oracle import oracle as orc
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    st = test_stack()
    C, Z, Y, X = st.shape
    src = lambda a, b, c, d: st[:, :, a:b, c:d]
    proj, zmap, labels = tiling.process_tiled_frame(src, C, Y, X, (2, 2), OracleTileBackend(), rank, world,
                                                    dist if world > 1 else None, "cpu",
                                                    segment=lambda plane: orc.watershed_segmentation(plane, 0.03, 3, 3))
    if rank == 0:
        np.savez(out_path, proj=proj, zmap=zmap, labels=labels)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
