"""Device-resident per-frame pipeline (projection -> segmentation -> cell tables) and frame sharding.

One process per GPU; frames are independent units (sp.py:211 loop, gui.py:1840 loop), so a movie shards
`frame t -> rank t % world` with no data-path collective; only the per-frame cell tables are gathered to rank 0
for track stitching (tissue_info.track_cells_iterator, ti.py:2037-2113), which is sequential over frames.
"""
import ctypes

import numpy as np

from . import _lib
from .basic_image_manipulations import gaussian_taps


class FramePipeline:
    """Keeps one frame's buffers resident in HBM between stages (inputs uploaded once, outputs fetched on demand)."""

    def __init__(self, C, Z, Y, X, reference_channel=0, airyscan=False, atoh_shift=0,
                 imgthresh=0.03, stdeviation=3.0, blocksize=3, device=None, use_torch=False):
        if device is not None:
            _lib.init(device)
        self.lib = _lib.lib()
        self.C, self.Z, self.Y, self.X = C, Z, Y, X
        self.ref, self.airy, self.atoh = reference_channel, airyscan, atoh_shift
        self.imgthresh, self.std, self.block = imgthresh, stdeviation, blocksize
        self.t05, self.t1, self.t2, self.t30 = (gaussian_taps(s) for s in (0.5, 1.0, 2.0, 30.0))
        self.tseg = gaussian_taps(stdeviation)
        P = Y * X
        self._proj_t = None
        if use_torch:  # projection buffer owned by torch so that the U-Net path consumes it without a copy
            import torch

            class _View(object):
                pass
            dev = torch.device("cuda", _lib.device_for_thread() or 0)
            self._proj_t = torch.empty((C, Y, X), dtype=torch.float64, device=dev)
            self.d_proj = _View()
            self.d_proj.ptr = self._proj_t.data_ptr()
            self.d_proj.download = lambda shape, dtype, t=self._proj_t: t.cpu().numpy()
        else:
            self.d_proj = _lib.DeviceBuffer(C * P * 8)
        self.d_zmap = _lib.DeviceBuffer(P * 8)
        self.d_labels = _lib.DeviceBuffer(P * 4)
        self.flags = ctypes.c_int32(0)
        self.n_labels = 0

    def upload_stack(self, stack_u16):
        stack_u16 = np.ascontiguousarray(stack_u16, dtype=np.uint16)
        assert stack_u16.shape == (self.C, self.Z, self.Y, self.X)
        buf = _lib.DeviceBuffer(stack_u16.nbytes)
        buf.upload(stack_u16)
        return buf

    def project(self, d_stack):
        """P0-P9 (sp.py:17-85) on a resident uint16 stack; asynchronous."""
        _lib.check(self.lib.tip_project_u16_dev(
            _lib.dptr(d_stack.ptr), self.C, self.Z, self.Y, self.X, 0, self.Z, 0, self.ref,
            1 if self.airy else 0, self.atoh, _lib.ptr(self.t05), _lib.ptr(self.t1), _lib.ptr(self.t2),
            _lib.ptr(self.t30), _lib.dptr(self.d_proj.ptr), _lib.dptr(self.d_zmap.ptr)))

    def segment(self, channel=0):
        """W1-W3 (bim.py:446-476) on the resident projection of `channel`."""
        P = self.Y * self.X
        img = self.d_proj.ptr + channel * P * 8
        _lib.check(self.lib.tip_watershed_segmentation_f64_dev(
            _lib.dptr(img), _lib.dptr(self.d_labels.ptr), self.Y, self.X, ctypes.c_double(self.imgthresh),
            _lib.ptr(self.tseg), self.tseg.size, self.block, ctypes.byref(self.flags)))

    def segment_unet(self, predictor, atoh_channel=1, zo_channel=0):
        """U1-U5 (pl.py:90-198) on the resident projection: (atoh, zo) planes transposed to (X, Y) as gui.py:2059-2061
        hands them over; labels stay on the device (int32 (X, Y))."""
        import torch
        if getattr(self, "_proj_t", None) is None:
            raise RuntimeError("FramePipeline(use_torch=True) is needed for the U-Net path")
        # the projection was written on the library's stream: torch's current stream waits for it (no host round trip)
        _lib.check(self.lib.tip_stream_wait_tip(ctypes.c_void_p(torch.cuda.current_stream(self._proj_t.device).cuda_stream)))
        # (atoh, zo) planes, each transposed: one straight gather + a strided view (prepare_image transposes back while it reads)
        img = self._proj_t[[atoh_channel, zo_channel]].transpose(1, 2)
        lab, hc = predictor.predict(img, return_device=True)
        self._unet_labels = lab
        return lab, hc

    def cell_tables(self, max_cells=None, labels_ptr=None, shape=None):
        """C1-C2 (ti.py:880-909, 1815-1842): per-cell reductions + neighbour pairs on the resident label map; the small
        per-cell arrays come back to the host (they are what a rank gathers for track stitching)."""
        P = self.Y * self.X
        lab_ptr = self.d_labels.ptr if labels_ptr is None else labels_ptr
        LY, LX = (self.Y, self.X) if shape is None else shape
        ncells = max_cells or int(self.lib.tip_last_watershed_labels())   # labels are 1..ncells
        if ncells <= 0:
            self.tables = dict(area=np.zeros(0, np.int64), bbox=np.zeros((0, 4), np.int64), sumy=np.zeros(0, np.int64),
                               sumx=np.zeros(0, np.int64), pc=np.zeros((0, 3), np.int64), pairs=np.zeros((0, 2), np.int32))
            return self.tables
        # ONE device block for every table -- [area n | bbox 4n | sumy n | sumx n | pc 3n] int64, then the pair list -- so
        # that the tables come back in a single device-to-host copy (six separate downloads were six round trips of
        # ~45 us each: 0.3 ms of a 5.9 ms frame)
        if getattr(self, "_tables", None) is None or self._tables[0] < ncells:
            cap = max(1024, int(ncells * 1.5))
            self._tables = (cap, _lib.DeviceBuffer(cap * (80 + 128)))
        cap, d_tab = self._tables
        n = ncells
        base = d_tab.ptr
        o_area, o_bbox, o_sy, o_sx, o_pc, o_pairs = 0, 8 * n, 40 * n, 48 * n, 56 * n, 80 * n
        _lib.check(self.lib.tip_regionprops_i32_dev(_lib.dptr(lab_ptr), None, LY, LX, n,
                                                    _lib.dptr(base + o_area), _lib.dptr(base + o_bbox), _lib.dptr(base + o_sy),
                                                    _lib.dptr(base + o_sx), _lib.dptr(base + o_pc), None))
        npairs = ctypes.c_int64(0)
        _lib.check(self.lib.tip_neighbor_pairs_i32_dev(_lib.dptr(lab_ptr), LY, LX,
                                                       _lib.dptr(base + o_pairs), ctypes.c_int64(16 * cap),
                                                       ctypes.byref(npairs)))
        npair = int(npairs.value)
        blob = d_tab.download((80 * n + 8 * npair,), np.uint8)
        i64 = blob[:80 * n].view(np.int64)
        self.tables = dict(
            area=i64[:n], bbox=i64[n:5 * n].reshape(n, 4), sumy=i64[5 * n:6 * n], sumx=i64[6 * n:7 * n],
            pc=i64[7 * n:10 * n].reshape(n, 3), pairs=blob[80 * n:].view(np.int32).reshape(npair, 2))
        return self.tables

    def sync(self):
        _lib.check(self.lib.tip_sync())

    def fetch_projection(self):
        return (self.d_proj.download((self.C, self.Y, self.X), np.float64),
                self.d_zmap.download((self.Y, self.X), np.int64))

    def fetch_labels(self):
        return self.d_labels.download((self.Y, self.X), np.int32)


def frames_for_rank(n_frames, rank, world):
    """frame t -> rank t % world."""
    return list(range(rank, n_frames, world))
