"""Frame-sharded time-lapse processing: one process per GPU, frame t -> rank t % world.

Every per-frame stage (projection, segmentation, cell tables) is independent, so the data path needs no
collective.  The one exchange step is track stitching (tissue_info.track_cells_iterator, ti.py:2037-2113):

  1. all ranks all-gather the small per-frame centroid tables (cy, cx; a few thousand rows per frame);
  2. the rank that owns frame t looks the drift-corrected centroids of frame t-1 up in ITS resident label map
     (3x3 max-filtered, ti.py:2081) -- the only dense-array part of the tracker stays next to the data;
  3. the resulting index arrays (one int per cell) are gathered to rank 0 (RCCL gather = grouped send/recv over
     xGMI, a direct all-to-one, KBs per frame), which runs the sequential id propagation.

With `estimate_drift=True` the frame-to-frame drift (Tissue.update_drift, ti.py:1982-2035) is computed inside the
sharded driver as well: the owner of frame t needs frame t-1's reference-channel projection, which lives on rank
(t-1) % world -- one point-to-point transfer of a (Y, X) float64 plane per frame (33.5 MB at 2048^2, rank r -> r+1 over
xGMI, all pairs at once) -- and runs the phase correlation next to its own plane.  `stitcher="linker"` replaces the
label-lookup tracker by the trackpy-model linker (ti.py:1881-1933, linking.FrameLinker) on rank 0.

`backend` supplies the per-frame compute so the same driver runs on GPUs (GpuFrameBackend) and, for the
multi-process CPU tests, on a stand-in backend with the gloo process group.
"""
import ctypes

import numpy as np


class _Job(object):
    """One process_frames call: a shared frame iterator the workers pull from."""

    def __init__(self, it, frame_source, nworkers):
        import threading
        self.it, self.frame_source = it, frame_source
        self.lock, self.done = threading.Lock(), threading.Semaphore(0)
        self.out, self.errors = {}, []


class _FrameWorker(object):
    """A worker thread of GpuFrameBackend with its own library context and FramePipeline, alive until it is sent None."""

    def __init__(self, backend):
        import queue
        import threading
        self.backend = backend
        self.jobs = queue.Queue()
        self.thread = threading.Thread(target=self.run, daemon=True)
        self.thread.start()

    def join(self):
        self.thread.join()

    def run(self):
        from . import _lib
        from .pipeline import FramePipeline
        b = self.backend
        pipe, setup_error = None, None
        try:
            _lib.init(b.device)
            pipe = FramePipeline(*b._shape, **b._kw)
        except BaseException as e:
            setup_error = e
        try:
            while True:
                job = self.jobs.get()
                if job is None:
                    return
                try:
                    if setup_error is not None:
                        raise setup_error
                    while True:
                        with job.lock:
                            t = next(job.it, None)
                        if t is None:
                            break
                        job.out[t] = b._process_with(pipe, t, job.frame_source(t))
                except BaseException as e:      # a dead worker must fail the movie, not shorten it silently
                    job.errors.append(e)
                finally:
                    job.done.release()
        finally:
            pipe = None                          # (its DeviceBuffers go back before the context that launched on them)
            try:
                _lib.load().tip_shutdown()
            except Exception:
                pass


class GpuFrameBackend(object):
    """Per-frame compute on this rank's MI355X through FramePipeline.  What later stages need stays resident per owned
    frame: the int32 label map (tracker look-ups) and, when drift is estimated, the reference channel's projection plane
    -- both kept with device-to-device copies.  The planes are torch tensors so that RCCL can send them as they are."""

    def __init__(self, C, Z, Y, X, device=None, keep_planes=False, inflight=1, **kw):
        from .pipeline import FramePipeline
        from . import _lib
        self._shape, self._kw = (C, Z, Y, X), kw
        self.device = device
        self.pipe = FramePipeline(C, Z, Y, X, device=device, **kw)
        if device is None:
            self.device = _lib.device_for_thread()
        self.Y, self.X = Y, X
        self.keep_planes = keep_planes
        self.inflight = max(1, int(inflight))
        self.labels = {}   # frame -> DeviceBuffer (int32 label map)
        self.planes = {}   # frame -> torch tensor (Y, X) float64 on this GPU
        self._workers = []  # persistent worker threads (process_frames)

    def process_frame(self, t, stack_u16):
        return self._process_with(self.pipe, t, stack_u16)

    def process_frames(self, frames, frame_source):
        """All of this rank's frames, `inflight` at a time: worker threads, each with its own HIP stream, workspaces and
        FramePipeline (the library is re-entrant per thread), pull frames from a shared iterator -- one frame's host->device
        upload overlaps the other frames' kernels, and the latency-bound watershed of one overlaps the projection of
        another.  frame_source(t) is called from the worker threads.

        The workers live as long as the backend (process_movie calls this once per round): a thread's library context -- its HIP
        stream, workspace pool and order-statistic state -- and its pipeline buffers are created once and released by close(),
        so a long movie neither re-allocates them every round nor piles up one context per round."""
        import threading
        frames = list(frames)
        if self.inflight <= 1 or len(frames) <= 1:
            return {t: self.process_frame(t, frame_source(t)) for t in frames}
        n = min(self.inflight, len(frames))
        while len(self._workers) < n:
            self._workers.append(_FrameWorker(self))
        job = _Job(iter(frames), frame_source, n)
        for w in self._workers[:n]:
            w.jobs.put(job)
        for _ in range(n):
            job.done.acquire()
        if job.errors:
            raise job.errors[0]
        return job.out

    def close(self):
        """Ends the worker threads; each releases its pipeline buffers and its library context (tip_shutdown) on the way out."""
        workers, self._workers = self._workers, []
        for w in workers:
            w.jobs.put(None)
        for w in workers:
            w.join()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _process_with(self, p, t, stack_u16):
        from . import _lib
        if hasattr(stack_u16, "data_ptr"):        # a (pinned) torch tensor: copied straight from its storage
            nbytes = stack_u16.numel() * stack_u16.element_size()
            d_stack = _lib.DeviceBuffer(nbytes)
            _lib.check(p.lib.tip_memcpy_h2d(_lib.dptr(d_stack.ptr), _lib.dptr(stack_u16.data_ptr()), nbytes))
        else:
            d_stack = p.upload_stack(stack_u16)
        p.project(d_stack)
        p.segment(0)
        nbytes = self.Y * self.X * 4
        keep = _lib.DeviceBuffer(nbytes)
        _lib.check(p.lib.tip_memcpy_d2d(_lib.dptr(keep.ptr), _lib.dptr(p.d_labels.ptr), nbytes))
        self.labels[t] = keep
        if self.keep_planes:
            import torch
            dev = torch.device("cuda", _lib.device_for_thread() or 0)
            plane = torch.empty((self.Y, self.X), dtype=torch.float64, device=dev)
            # (the block may still be in use by kernels queued on torch's stream: order the library's copy after them)
            _lib.check(p.lib.tip_wait_stream(ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            _lib.check(p.lib.tip_memcpy_d2d(_lib.dptr(plane.data_ptr()), _lib.dptr(p.d_proj.ptr + p.ref * self.Y * self.X * 8),
                                            self.Y * self.X * 8))
            self.planes[t] = plane
        tab = p.cell_tables()          # (synchronises: the small per-cell arrays come to the host)
        d_stack.free()
        area = tab["area"].astype(np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            cy = tab["sumy"] / area
            cx = tab["sumx"] / area
        return dict(area=tab["area"], cy=np.where(area > 0, cy, 0.0), cx=np.where(area > 0, cx, 0.0))

    def lookup(self, t, qy, qx):
        from . import _lib
        qy = np.ascontiguousarray(qy, dtype=np.int64)
        qx = np.ascontiguousarray(qx, dtype=np.int64)
        out = np.empty(qy.shape, np.int32)
        _lib.check(self.pipe.lib.tip_lookup_max3_i32_dev(_lib.dptr(self.labels[t].ptr), self.Y, self.X, _lib.ptr(qy),
                                                          _lib.ptr(qx), qy.size, _lib.ptr(out)))
        return out

    # -- drift (T2) ------------------------------------------------------------------------------------------
    def plane(self, t):
        return self.planes[t]

    def empty_plane(self):
        import torch
        from . import _lib
        return torch.empty((self.Y, self.X), dtype=torch.float64, device=torch.device("cuda", _lib.device_for_thread() or 0))

    def drift(self, t, prev_plane):
        """(row shift, column shift) that registers frame t onto frame t-1: Tissue.update_drift without a stage table
        (ti.py:1982-2035 -> calculate_refine_drift with a zero coarse shift -> phase_cross_correlation(upsample 100))."""
        import torch
        from ._registration import phase_cross_correlation_dev
        torch.cuda.current_stream(prev_plane.device).synchronize()      # the received plane is complete
        sh = phase_cross_correlation_dev(prev_plane.data_ptr(), self.planes[t].data_ptr(), self.Y, self.X, 100)
        return float(sh[0]), float(sh[1])


def _all_gather_arrays(arrs, dist, world, device):
    """all-gather a list of float64 1-D arrays of per-rank varying length -> list (per rank) of arrays."""
    import torch
    flat = np.concatenate([np.asarray(a, np.float64).ravel() for a in arrs]) if arrs else np.zeros(0)
    n = torch.tensor([flat.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    buf = torch.zeros(m, dtype=torch.float64, device=device)
    buf[:flat.size] = torch.from_numpy(flat).to(device)
    out = [torch.zeros(m, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(out, buf)
    return [o[:s].cpu().numpy() for o, s in zip(out, sizes)]


def _gather_to_root(flat, dist, rank, world, device):
    """gatherv of an int64 array to rank 0 (counts all-gathered first, then one padded gather)."""
    import torch
    flat = np.asarray(flat, np.int64).ravel()
    n = torch.tensor([flat.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    buf = torch.zeros(m, dtype=torch.int64, device=device)
    buf[:flat.size] = torch.from_numpy(flat).to(device)
    out = [torch.zeros(m, dtype=torch.int64, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    return [o[:s].cpu().numpy() for o, s in zip(out, sizes)]


def assign_track_ids(prev_ids, hit, n_cur, start_ids=None):
    """One step of the label-lookup tracker's id bookkeeping (ti.py:2041-2046, 2092-2106).

    prev_ids[i] = track id of row i of the previous frame; hit[i] = label (row + 1) of the current frame found under
    that row's drift-corrected centroid, 0 / -1 when nothing was hit.  Each previous id is handed on at most once and
    each current row receives at most one id (first occurrence in np.unique order, as upstream); rows left without an
    id get fresh ones above the largest id in use.  With start_ids the step only fills the zeros (first frame)."""
    if start_ids is not None:
        ids = np.asarray(start_ids, dtype=np.int64).copy()
    else:
        ids = np.zeros(n_cur, np.int64)
        row = np.asarray(hit, dtype=np.int64) - 1
        sel = row >= 0
        carried, row = np.asarray(prev_ids)[sel], row[sel]
        _, first = np.unique(carried, return_index=True)      # a previous id is used once ...
        carried, row = carried[first], row[first]
        _, first = np.unique(row, return_index=True)          # ... and a current row is claimed once
        ids[row[first]] = carried[first]
    fresh = ids == 0
    top = ids.max() if ids.size else 0
    ids[fresh] = np.arange(top + 1, top + fresh.sum() + 1)
    return ids


def propagate_ids(tables, lookups):
    """Sequential part of the tracker on rank 0.  tables[t]: dict(area, cy, cx) over row index = label - 1;
    lookups[t] (t >= 1): for every row of frame t-1 the 3x3-max-filtered label of frame t under its drift-corrected
    centroid (-1 = outside / absent row).  Returns the per-frame id arrays as the reference leaves them in cells_info.label."""
    n0 = tables[0]["area"].size
    # calculate_frame_cellinfo leaves label = row + 1 for present rows and 0 for absent ones (ti.py:893)
    first = np.where(tables[0]["area"] > 0, np.arange(1, n0 + 1), 0)
    out = [assign_track_ids(None, None, n0, start_ids=first)]
    for t in range(1, len(tables)):
        out.append(assign_track_ids(out[-1], lookups[t], tables[t]["area"].size))
    return out


def exchange_planes(n_frames, sends, recvs, backend, rank, world, dist):
    """Plane hand-off for drift estimation: frame t's owner needs frame t-1's reference-channel plane, which lives on rank
    (t-1) % world.  `sends`: this rank's frames whose planes go to rank (rank+1) % world; `recvs`: frames t-1 owned by rank
    (rank-1) % world whose planes arrive here -- both in increasing order on either side of a pair, all pairs at once
    (grouped isend/irecv: over RCCL each pair has its own xGMI link, no ring).  Returns {t: plane of frame t-1}."""
    prev = {}
    if world == 1:
        for t in recvs:
            prev[t + 1] = backend.plane(t)
        return prev
    # gloo (the CPU test harness) moves host tensors only: device planes are staged through the host there; RCCL sends
    # the device planes as they are
    staged = dist.get_backend() == "gloo"
    ops, landing = [], {}
    for t in sends:
        src = backend.plane(t)
        ops.append(dist.P2POp(dist.isend, src.cpu() if staged and src.is_cuda else src, (rank + 1) % world))
    for t in recvs:
        prev[t + 1] = backend.empty_plane()
        landing[t + 1] = prev[t + 1].cpu() if staged and prev[t + 1].is_cuda else prev[t + 1]
        ops.append(dist.P2POp(dist.irecv, landing[t + 1], (rank - 1) % world))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for t, buf in landing.items():
        if buf is not prev[t]:
            prev[t].copy_(buf)
    return prev


def link_ids(tables, drifts):
    """stitcher="linker": track ids from the trackpy-model linker (ti.py:1881-1933) on rank 0: features are the present
    rows' (cy, cx, area) shifted by the cumulative drift; id = particle + 1, absent rows 0."""
    from .linking import FrameLinker, embed
    linker = FrameLinker(search_range=100, adaptive_stop=10, adaptive_step=0.95, memory=3)
    total = np.zeros(2)
    out = []
    for t, tb in enumerate(tables):
        if t > 0:
            total = total + np.asarray(drifts[t], dtype=np.float64)
        present = np.flatnonzero(tb["area"] > 0)
        particles = linker.link(embed(tb["cy"][present] + total[0], tb["cx"][present] + total[1], tb["area"][present]))
        ids = np.zeros(tb["area"].size, np.int64)
        ids[present] = np.asarray(particles, dtype=np.int64) + 1
        out.append(ids)
    return out


def process_movie(n_frames, frame_source, backend, rank=0, world=1, dist=None, device="cpu", drifts=None,
                  estimate_drift=False, stitcher="lookup", block_frames=None):
    """Runs the sharded pipeline.  frame_source(t) -> uint16 stack (or whatever backend.process_frame takes).
    Returns on rank 0: (tables per frame, track ids per frame); on other ranks (None, None).  tables[t]["drift"] holds
    the (row, column) drift used between frames t-1 and t (estimated by frame t's owner when estimate_drift).

    The movie is worked off in ROUNDS of `block_frames` frames per rank (round k = global frames [k B W, (k+1) B W)): while
    the workers compute round k+1, this thread runs round k's exchange -- planes to the neighbour rank, drift, all-gather of
    the centroid tables, owner-side look-ups, gather of the index arrays to rank 0 -- so the stitching traffic and the drift
    correlations hide behind the next frames' kernels instead of forming a tail after the last frame.  Every rank runs the same
    number of rounds (a rank without frames in a round takes part with empty payloads: the collectives stay matched), so any
    n_frames works, including fewer frames than ranks.  block_frames=None: the whole shard in one round (no overlap)."""
    import threading
    if stitcher not in ("lookup", "linker"):
        raise ValueError("stitcher must be 'lookup' or 'linker'")
    if drifts is None:
        drifts = np.zeros((n_frames, 2))
    drifts = np.array(drifts, dtype=np.float64)
    per_round = (int(block_frames) if block_frames else max(1, -(-n_frames // world))) * world
    n_rounds = max(1, -(-n_frames // per_round))
    rounds = [[t for t in range(k * per_round, min(n_frames, (k + 1) * per_round)) if t % world == rank] for k in range(n_rounds)]

    def compute(frames, out):
        try:
            if not frames:
                return
            if hasattr(backend, "process_frames"):
                out.update(backend.process_frames(frames, frame_source))        # several frames in flight on this rank's GPU
            else:
                out.update({t: backend.process_frame(t, frame_source(t)) for t in frames})
        except BaseException as e:
            out["error"] = e

    tables, lookups, held_planes = {}, {}, {}
    results = [dict() for _ in range(n_rounds)]
    worker = threading.Thread(target=compute, args=(rounds[0], results[0]))
    worker.start()
    for k in range(n_rounds):
        worker.join()
        if "error" in results[k]:
            raise results[k]["error"]
        local, mine = results[k], rounds[k]
        if k + 1 < n_rounds:                       # the next round computes while this one is exchanged
            worker = threading.Thread(target=compute, args=(rounds[k + 1], results[k + 1]))
            worker.start()
        lo, hi = k * per_round, min(n_frames, (k + 1) * per_round)
        if estimate_drift:
            sends = [t for t in mine if t + 1 < n_frames]
            recvs = [t for t in range(lo, hi) if t % world == (rank - 1) % world and t + 1 < n_frames]
            held_planes.update(exchange_planes(n_frames, sends, recvs, backend, rank, world, dist))
            for t in mine:
                if t >= 1:
                    drifts[t] = backend.drift(t, held_planes.pop(t))
        for t in mine:
            local[t]["drift"] = drifts[t].copy()
        # 1. centroid tables of the round everywhere
        if world > 1:
            payload = []
            for t in mine:
                tb = local[t]
                payload += [np.array([t, tb["area"].size, tb["drift"][0], tb["drift"][1]], np.float64),
                            tb["area"].astype(np.float64), tb["cy"], tb["cx"]]
            for flat in _all_gather_arrays(payload, dist, world, device):
                pos = 0
                while pos < flat.size:
                    t, n = int(flat[pos]), int(flat[pos + 1])
                    drift_t = flat[pos + 2:pos + 4].copy()
                    pos += 4
                    tables[t] = dict(area=flat[pos:pos + n].astype(np.int64), cy=flat[pos + n:pos + 2 * n],
                                     cx=flat[pos + 2 * n:pos + 3 * n], drift=drift_t)
                    pos += 3 * n
        else:
            tables.update({t: local[t] for t in mine})
        if stitcher == "linker":
            continue
        # 2. owners look the previous frame's centroids up in their resident label maps
        my_lookups = {}
        for t in mine:
            if t == 0:
                continue
            prev = tables[t - 1]
            cy = prev["cy"] - tables[t]["drift"][0]
            cx = prev["cx"] - tables[t]["drift"][1]
            qy, qx = np.round(cy).astype(np.int64), np.round(cx).astype(np.int64)
            res = backend.lookup(t, qy, qx)
            my_lookups[t] = np.where(prev["area"] > 0, res, -1)   # absent rows never match (empty_cell / zero-area rows)
        # 3. gather to rank 0
        if world > 1:
            flat = []
            for t, r in my_lookups.items():
                flat += [np.array([t, r.size], np.int64), r.astype(np.int64)]
            flat = np.concatenate(flat) if flat else np.zeros(0, np.int64)
            parts = _gather_to_root(flat, dist, rank, world, device)
            if rank == 0:
                for p in parts:
                    pos = 0
                    while pos < p.size:
                        t, n = int(p[pos]), int(p[pos + 1])
                        lookups[t] = p[pos + 2:pos + 2 + n]
                        pos += 2 + n
        else:
            lookups.update(my_lookups)
    if rank != 0:
        return None, None
    tabs = [tables[t] for t in range(n_frames)]
    if stitcher == "linker":
        return tabs, link_ids(tabs, [tb["drift"] for tb in tabs])
    ids = propagate_ids(tabs, [None] + [lookups[t] for t in range(1, n_frames)])
    return tabs, ids
