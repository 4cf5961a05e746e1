"""Spatial tiling of ONE large frame across GPUs (BASELINE config 5: 4096 x 4096 x 60).

The 3-D part of the pipeline -- the surface projection, 8x the headline frame's voxels -- is what is worth splitting:
a tile plus a halo of HALO pixels reproduces the untiled projection of its interior bit for bit, because every output
pixel depends on a bounded neighbourhood (sigma 1 -> 4 px, sigma 30 -> 120 px for the z-map; sigma 2 -> 8 px more for
the one-hot mask: 132 px in all) and the filters replicate edges only at the true frame border, where tile and frame
coincide.  The one global quantity is the reference channel's 95th percentile (sp.py:33-36): the tiles add up 65536-bin
histograms of their interiors (all-reduce, 512 KB) before anyone clips.

The 2-D stages (threshold, blur, watershed, cell tables) are NOT tiled: a 4096^2 float64 plane is 134 MB and its
watershed takes milliseconds on one GPU, while an exact watershed across tile seams would need the flood order of the
whole plane.  So the tile interiors are gathered to the frame's owner (rank 0), which stitches them and runs the 2-D
stages on the full plane.  Exchange steps: one all-reduce (histograms), one gather (interiors); no ring.

`backend` supplies the per-tile compute so that the same driver runs on GPUs (GpuTileBackend) and, for the multi-process
CPU tests, on a stand-in with the gloo process group.
"""
import ctypes

import numpy as np

HALO = 4 + 120 + 8   # radius of sigma 1 (4 * 1) + sigma 30 (4 * 30) + sigma 2 (4 * 2): support of one projected pixel


def tile_boxes(Y, X, ny, nx, halo=HALO):
    """ny x nx tiles covering the (Y, X) plane: [(interior (y0, y1, x0, x1), padded (py0, py1, px0, px1)), ...] in raster
    order of the tiles; the padded box is the interior grown by `halo` and clipped to the frame."""
    ys = [round(i * Y / ny) for i in range(ny + 1)]
    xs = [round(j * X / nx) for j in range(nx + 1)]
    out = []
    for i in range(ny):
        for j in range(nx):
            y0, y1, x0, x1 = ys[i], ys[i + 1], xs[j], xs[j + 1]
            out.append(((y0, y1, x0, x1), (max(0, y0 - halo), min(Y, y1 + halo), max(0, x0 - halo), min(X, x1 + halo))))
    return out


class GpuTileBackend(object):
    """Per-tile compute on this rank's MI355X: the uploaded tile stays resident between its histogram and its projection."""

    def __init__(self, reference_channel=0, airyscan=False, atoh_shift=0, device=None):
        from . import _lib
        if device is not None:
            _lib.init(device)
        self.lib = _lib.lib()
        self.ref, self.airy, self.atoh = reference_channel, airyscan, atoh_shift
        self.resident = {}

    def histogram(self, key, tile_u16, box):
        """uint64[65536] histogram of the reference channel inside `box` = (y0, y1, x0, x1) in tile coordinates."""
        from . import _lib
        tile_u16 = np.ascontiguousarray(tile_u16, dtype=np.uint16)
        C, Z, Yt, Xt = tile_u16.shape
        d = _lib.DeviceBuffer(tile_u16.nbytes).upload(tile_u16)
        self.resident[key] = (d, tile_u16.shape)
        h = _lib.DeviceBuffer(65536 * 8)
        _lib.check(self.lib.tip_memset(_lib.dptr(h.ptr), 0, ctypes.c_size_t(65536 * 8)))
        y0, y1, x0, x1 = box
        _lib.check(self.lib.tip_hist_u16_box_dev(_lib.dptr(d.ptr), C, Z, Yt, Xt, self.ref, 0, Z, int(y0), int(y1), int(x0), int(x1),
                                                 1 if self.airy else 0, _lib.dptr(h.ptr)))
        out = h.download((65536,), np.uint64)
        h.free()
        return out

    def project(self, key, hist):
        """Projection of the resident tile `key` clipped with the whole frame's histogram -> (proj f64 (C,Yt,Xt), zmap i64)."""
        from . import _lib
        from .basic_image_manipulations import gaussian_taps
        d, (C, Z, Yt, Xt) = self.resident.pop(key)
        h = _lib.DeviceBuffer(65536 * 8).upload(np.ascontiguousarray(hist, dtype=np.uint64))
        dp, dz = _lib.DeviceBuffer(C * Yt * Xt * 8), _lib.DeviceBuffer(Yt * Xt * 8)
        t05, t1, t2, t30 = (gaussian_taps(s) for s in (0.5, 1.0, 2.0, 30.0))
        _lib.check(self.lib.tip_project_u16_hist_dev(_lib.dptr(d.ptr), C, Z, Yt, Xt, 0, Z, 0, self.ref, 1 if self.airy else 0,
                                                     self.atoh, _lib.ptr(t05), _lib.ptr(t1), _lib.ptr(t2), _lib.ptr(t30),
                                                     _lib.dptr(h.ptr), _lib.dptr(dp.ptr), _lib.dptr(dz.ptr)))
        proj, zmap = dp.download((C, Yt, Xt), np.float64), dz.download((Yt, Xt), np.int64)
        for b in (d, h, dp, dz):
            b.free()
        return proj, zmap


def _gather_f64_to_root(flat, dist, rank, world, device):
    import torch
    flat = np.asarray(flat, np.float64).ravel()
    n = torch.tensor([flat.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    buf = torch.zeros(m, dtype=torch.float64, device=device)
    buf[:flat.size] = torch.from_numpy(flat).to(device)
    out = [torch.zeros(m, dtype=torch.float64, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    return [o[:s].cpu().numpy() for o, s in zip(out, sizes)]


def project_tiled(tile_source, C, Y, X, grid, backend, rank=0, world=1, dist=None, device="cpu", halo=HALO):
    """Tiled surface projection of one (C, Z, Y, X) frame.  tile_source(py0, py1, px0, px1) -> the uint16 sub-stack
    (C, Z, py1-py0, px1-px0) (a rank only ever asks for its own tiles).  grid = (ny, nx); tile k -> rank k % world.
    Returns on rank 0 (proj float64 (C, Y, X), zmap int64 (Y, X)) -- identical to the untiled projection -- else
    (None, None)."""
    boxes = tile_boxes(Y, X, grid[0], grid[1], halo)
    mine = [k for k in range(len(boxes)) if k % world == rank]
    hist = np.zeros(65536, np.uint64)
    for k in mine:
        (y0, y1, x0, x1), (py0, py1, px0, px1) = boxes[k]
        hist += backend.histogram(k, tile_source(py0, py1, px0, px1), (y0 - py0, y1 - py0, x0 - px0, x1 - px0))
    if world > 1:
        import torch
        t = torch.from_numpy(hist.astype(np.int64)).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)          # the frame's histogram on every rank
        hist = t.cpu().numpy().astype(np.uint64)
    payload = []
    for k in mine:
        (y0, y1, x0, x1), (py0, py1, px0, px1) = boxes[k]
        proj, zmap = backend.project(k, hist)
        sy, sx = slice(y0 - py0, y1 - py0), slice(x0 - px0, x1 - px0)
        payload += [np.array([k], np.float64), proj[:, sy, sx].ravel(), zmap[sy, sx].astype(np.float64).ravel()]
    flat = np.concatenate(payload) if payload else np.zeros(0)
    parts = _gather_f64_to_root(flat, dist, rank, world, device) if world > 1 else [flat]
    if rank != 0:
        return None, None
    proj = np.empty((C, Y, X), np.float64)
    zmap = np.empty((Y, X), np.int64)
    for part in parts:
        pos = 0
        while pos < part.size:
            k = int(part[pos])
            (y0, y1, x0, x1), _ = boxes[k]
            n = (y1 - y0) * (x1 - x0)
            proj[:, y0:y1, x0:x1] = part[pos + 1:pos + 1 + C * n].reshape(C, y1 - y0, x1 - x0)
            zmap[y0:y1, x0:x1] = part[pos + 1 + C * n:pos + 1 + (C + 1) * n].reshape(y1 - y0, x1 - x0).astype(np.int64)
            pos += 1 + (C + 1) * n
    return proj, zmap


def process_tiled_frame(tile_source, C, Y, X, grid, backend, rank=0, world=1, dist=None, device="cpu",
                        imgthresh=0.03, stdeviation=3, blocksize=3, segment=None, halo=HALO):
    """Config 5's per-frame path: tiled projection on all ranks, then the 2-D stages on the stitched plane on rank 0
    (`segment(plane) -> int32 labels`; default: basic_image_manipulations.watershed_segmentation on this rank's GPU).
    Returns on rank 0 (proj, zmap, labels), else (None, None, None)."""
    proj, zmap = project_tiled(tile_source, C, Y, X, grid, backend, rank, world, dist, device, halo)
    if rank != 0:
        return None, None, None
    if segment is None:
        from .basic_image_manipulations import watershed_segmentation
        segment = lambda plane: watershed_segmentation(plane, imgthresh, stdeviation, blocksize)
    return proj, zmap, segment(proj[backend.ref])
