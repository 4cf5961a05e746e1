#!/usr/bin/env python3
"""bench.py -- end-to-end frames/s of the per-frame confocal-stack hot path on MI355X.

One "step" = one pass of the hot path over one synthetic 2048x2048x30 (C=2, uint16) frame that is already
resident in HBM: surface projection (sp.py:17-85) -> watershed_segmentation (bim.py:446-476) -> cell tables
(ti.py:880-909).  One process per GPU; frames are independent units, so N GPUs process N frames per step with no
data-path collective (weak scaling).  Prints ONE JSON line on rank 0.

`value` is the classical-segmentation variant (`config.workload` says so).  BASELINE.json's config 3 names the U-Net
variant (projection -> U-Net (pl.py:124-199) -> threshold / closing / watershed tail -> cell tables): the default run
times that too, right after the classical leg, and reports it in the same line under "unet" with its MFMA roofline
(`--workload unet` makes it the headline instead; `--no-unet-leg` skips it).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6   # MI355X datasheet FP64 vector (spec)
FP32_VALU_PEAK_TF = 157.3  # MI355X_MICROARCH.md: FP32 vector (pure FMA stream; the passes' add+fma mix sustains 104-109)


def algorithmic_bytes(kernel, C, Z, Y, X):
    """Compulsory HBM bytes per launch (logical input read once + logical output written once), SURVEY.md 8(d)."""
    V, P = Z * Y * X, Y * X
    table = {
        "hist_u16": V * 2,
        "corr_z_u16clip": V * 2 + V * 4,
        "corr_long_y": 2 * V * 4, "corr_long_x": 2 * V * 4,
        "score_fast_y": 2 * V * 4, "score_fast_x": 2 * V * 4,
        "preblur_fused": V * 2 + V * 4,   # the four short passes in one kernel: uint16 stack in, one float32 volume out
        "zpass_u16clip_x4": V * 2 + V * 4, "zpass_f32_x4": 2 * V * 4, "ypass_slide_r4": 2 * V * 4, "xpass_slide_r4": 2 * V * 4,
        "argmax_certify": V * 4 + P * 4, "mask_y_sparse": P * 4 + V * 4, "xpass_wmax_sparse": V * 4 + C * V * 2 + C * P * 8,
        # fused P6-P8: read the z-map and (logically) the stack once, write the projection; the blurred mask stays in LDS
        "mask_wmax_fused": P * 4 + C * V * 2 + C * P * 8,
        "argmax_z": V * 4 + P * 16,
        "mask_ypass": P * 4 + V * 4,
        "xpass_wmax": V * 4 + C * V * 2 + C * P * 8,
        # 2-D stages: watershed f64 -> i32 = P*8 + P*4 per frame; a launch of the tile kernel is one
        # global iteration over the same frame, so the per-launch figure is the per-frame one
        "ws_tiles": P * 12, "ws_tiles_wide": P * 12,
        "regionprops": P * 4, "neighbor_pairs": P * 4, "local_threshold": 2 * P * 8,
        "corr_generic_y_f64": 2 * P * 8, "corr_generic_x_f64": 2 * P * 8, "yslide_r12_f64": 2 * P * 8, "xslide_r12_f64": 2 * P * 8,
    }
    return table.get(kernel)


PMC_NAMES = {  # bench kernel label -> substring of the rocprofv3 kernel name in profiles/r*_pmc_traffic.json
    "ws_tiles": ("k_ws_tiles<16, 64, 3, 6, 2, 1>", "k_ws_tiles<16, 64, 4, 6, 3, 1>", "k_ws_tiles<16, 64, 3, 6"), "score_fast_y": ("k_corr_long_mfma<1", "k_corr_long_mfma2<1", "k_corr_long_fast<1"),
    "score_fast_x": ("k_corr_long_mfma<2", "k_corr_long_mfma2<2", "k_corr_long_fast<2"),
    "corr_long_y": "k_corr_long_f32<1", "corr_long_x": "k_corr_long_f32<2", "ypass_slide_r4": "k_ypass_slide",
    "xpass_slide_r4": "k_xpass_slide", "zpass_f32_x4": "k_zpass_r2_x4<Src4F32>", "zpass_u16clip_x4": "k_zpass_r2_x4<Src4U16Clip>",
    "regionprops": "k_regionprops", "hist_u16": "k_hist_u16", "mask_y_sparse": "k_mask_y_sparse",
    "xpass_wmax_sparse": "k_xpass_wmax_sparse", "argmax_certify": "k_argmax_certify", "mask_wmax_fused": "k_mask_wmax_fused", "preblur_fused": "k_preblur_fused",
}


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in separate
    runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when no profile is committed."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    key = PMC_NAMES.get(kernel)
    if not key or not paths:
        return None
    path = paths[-1]   # the latest committed set
    recs = json.load(open(path))
    for k in (key if isinstance(key, tuple) else (key,)):      # (alternatives: whichever variant of the kernel the profiled build ran)
        for name, rec in recs.items():
            if k in name:
                return (rec["fetch_MB_per_call_x2corrected"] + rec["write_MB_per_call"]) * 1e6
    return None


def algorithmic_dp_ops(kernel, Z, Y, X):
    """Separately rounded double-precision operations per launch (3 per tap pair + 1): the exact-scipy contract."""
    V = Z * Y * X
    if kernel in ("corr_long_y", "corr_long_x"):
        return V * (120 * 3 + 1)
    return None


def cpu_baseline_worker(Ys, Xs, Z, workload):
    """One CPU-baseline sample: the C/numpy oracle (a port of the reference path, single thread like scipy.ndimage) on a
    crop of the workload; prints the seconds it took."""
    from oracle import oracle as orc
    from tissue_image_processing_amd import synthetic
    st = synthetic.make_stack(Z, Ys, Xs, seed=1234)
    t0 = time.perf_counter()
    proj, zmap = orc.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
    if workload != "projection":
        lab = orc.watershed_segmentation(proj[0], 0.03, 3, 3)
        orc.frame_cellinfo(lab)
    print("CPU_BASELINE_SECONDS %.6f" % (time.perf_counter() - t0))


def cpu_baseline(sample_yx, Z, workload, nproc):
    """Runs the sample in `nproc` child processes at once (started BEFORE this process touches the GPU) and returns the
    list of per-process seconds: nproc = 1 is the single-core figure, nproc = host cores the embarrassingly parallel
    "N frames on N processes" one (scipy.ndimage / skimage are single-threaded)."""
    import subprocess
    Ys, Xs = sample_yx
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(Ys), str(Xs), str(Z), workload]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True) for _ in range(nproc)]
    secs = []
    for p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("cpu baseline worker failed")
        secs.append(float([l for l in out.splitlines() if l.startswith("CPU_BASELINE_SECONDS")][0].split()[1]))
    return secs


def bench_movie(args, rank, local_rank, world, dist, torch):
    """BASELINE config 4 in miniature: a time-lapse of `steps` frames per GPU sharded frame t -> rank t % world, every
    frame through projection -> segmentation -> cell tables, then the track stitching exchange (all-gather of centroid
    tables, owner-side label look-ups, gather of the index arrays to rank 0 over RCCL, sequential id propagation)."""
    from tissue_image_processing_amd import movie, synthetic
    Y, X, Z = args.size
    T = args.steps * world
    use_dist = world > 1
    sites_t, is_hc = synthetic.make_movie_sites(Y, X, T, seed=5)
    mine = list(range(rank, T, world))
    # the movie's frames wait in pinned host memory (where a reader thread would put them); every frame's host->device
    # copy is inside the timed region
    stacks = {t: torch.from_numpy(synthetic.make_stack(Z, Y, X, seed=200 + t, sites=sites_t[t], is_hc=is_hc)).pin_memory()
              for t in mine}
    backend = movie.GpuFrameBackend(2, Z, Y, X, device=local_rank, keep_planes=True, inflight=args.inflight)
    backend.process_frames([-1 - k for k in range(min(args.inflight, len(mine)))], lambda t: stacks[mine[0]])   # warm-up
    for k in range(args.inflight):
        backend.labels.pop(-1 - k, None)
        backend.planes.pop(-1 - k, None)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tabs, ids = movie.process_movie(T, lambda t: stacks[t], backend, rank, world, dist if world > 1 else None,
                                    torch.device("cuda", local_rank) if world > 1 else "cpu", estimate_drift=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        n_tracks = int(max(i.max() for i in ids))
        print(json.dumps({
            "metric": "frames/sec end-to-end (2048^2, z=30)", "value": T / elapsed, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": 1, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dx%dx%d_c2_u16:movie(%d frames, host upload from pinned memory included)+drift estimation+track_stitching" % (Y, X, Z, T),
                       "frames_in_flight_per_gpu": args.inflight,
                       "parallelism": "frame-sharded dp%d, neighbour-rank plane exchange for the drift, RCCL gather of "
                                      "per-frame tables to rank 0" % world,
                       "mean_abs_drift": [float(v) for v in np.mean(np.abs([tb["drift"] for tb in tabs[1:]]), axis=0)],
                       "tracks": n_tracks, "cells_last_frame": int(ids[-1].size)}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-worker":
        return cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 64; 16 for --workload movie, whose every frame is a distinct synthetic stack "
                         "generated on the host and kept in pinned memory)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--size", type=int, nargs=3, default=[2048, 2048, 30], metavar=("Y", "X", "Z"))
    ap.add_argument("--workload", default="auto", choices=["auto", "projection", "classical", "unet", "movie"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-unet-leg", action="store_true", help="skip the secondary U-Net leg of the default run")
    ap.add_argument("--unet-steps", type=int, default=9, help="timed steps of the secondary U-Net leg (2 warm-up steps)")
    ap.add_argument("--include-upload", action="store_true",
                    help="also time the host->device copy of every frame (pinned host buffers, copied by the worker that "
                         "then processes the frame, so uploads overlap other workers' kernels); NOT the headline value")
    ap.add_argument("--inflight", type=int, default=4,
                    help="frames in flight per GPU: host threads, each with its own HIP stream and workspaces (the library "
                         "is re-entrant per thread, like the reference's Qt workers); frames are independent units")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 16 if args.workload == "movie" else 64

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    Y, X, Z = args.size
    C = 2

    # CPU baseline first: its child processes are started before this process initialises the GPU
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and args.workload != "movie":
        wl = "classical" if args.workload in ("auto", "unet") else args.workload
        Ys, Xs = min(Y, 1408), min(X, 1408)
        scale = (Y * X) / float(Ys * Xs)
        host_cores = os.cpu_count() or 1
        try:
            host_cores = len(os.sched_getaffinity(0))
        except AttributeError:
            pass
        one = cpu_baseline((Ys, Xs), Z, wl, 1)[0]
        nproc = max(1, min(host_cores, 16))
        many = cpu_baseline((Ys, Xs), Z, wl, nproc) if nproc > 1 else [one]
        cpu = {"value": 1.0 / (one * scale), "unit": "frames/s", "cores": 1, "kind": "port", "host_cores": host_cores,
               "sample": "%dx%dx%d crop (1/%g of a frame) through the C/numpy oracle (%s path), %.1f s on one core, scaled "
                         "by pixel count" % (Ys, Xs, Z, scale, wl, one),
               "n_process": {"processes": nproc, "value": nproc / (max(many) * scale), "unit": "frames/s",
                             "sample": "the same crop in %d processes at once (one frame each), slowest %.1f s" % (nproc, max(many))}}

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    # TIP_BENCH_FORCE_DIST=1 (with torchrun --nproc-per-node 1) walks the RCCL process-group path on a one-GPU box
    use_dist = world > 1 or (os.environ.get("TIP_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from tissue_image_processing_amd import _lib, synthetic
    from tissue_image_processing_amd.pipeline import FramePipeline
    _lib.init(local_rank)
    if args.workload == "movie":
        return bench_movie(args, rank, local_rank, world, dist, torch)
    lib = _lib.lib()
    workload = args.workload
    if workload == "auto":
        workload = "classical"

    # synthetic frames, resident in HBM before the timed region (two distinct frames per rank, alternated)
    import threading
    st = synthetic.make_stack(Z, Y, X, seed=100 + rank)
    st_flip = np.ascontiguousarray(st[:, :, :, ::-1])
    steps_lock = threading.Lock()

    class Worker(object):
        """One frame in flight: own thread, own HIP stream / workspace pool (tip_init per thread), own resident buffers."""

        def __init__(self, wid, workload):
            self.wid = wid
            self.workload = workload
            self.todo = threading.Semaphore(0)
            self.done = threading.Semaphore(0)
            self.jobs = []
            self.report = {}
            self.error = None
            self.unet_ms = []
            self.thread = threading.Thread(target=self.run, daemon=True)
            self.thread.start()
            self.wait()  # wait for setup

        def run(self):
            try:
                self.run_inner()
            except BaseException as e:      # a dead worker must fail the run, not leave the main thread waiting
                self.error = e
                self.done.release()

        def run_inner(self):
            _lib.init(local_rank)
            self.pipe = FramePipeline(C, Z, Y, X, reference_channel=0, airyscan=False, use_torch=(self.workload == "unet"))
            self.predictor = None
            if self.workload == "unet":
                from tissue_image_processing_amd.prediction_local import SegmentationPredictor
                self.predictor = SegmentationPredictor(None, (2, X, Y), device=local_rank)  # random-init (none ship upstream)
            self.frames = [self.pipe.upload_stack(st), self.pipe.upload_stack(st_flip)]
            if self.predictor is not None:
                # random-init weights give an all-or-nothing class map; shift the head bias so that the thresholded map
                # (pl.py:168) covers half of the first frame: the tail then floods a boundary image with real structure
                self.pipe.project(self.frames[0])
                self.pipe.sync()
                pj = self.pipe._proj_t
                padded, _ = self.predictor.prepare_image(torch.stack([pj[1].T, pj[0].T]))
                self.predictor.model.calibrate_head(padded, 0.5)
            self.host = None
            if args.include_upload:   # pinned host copies of the two frames; the device buffers are re-filled every step
                self.host = [torch.from_numpy(st).pin_memory(), torch.from_numpy(st_flip).pin_memory()]
            self.done.release()
            while True:
                self.todo.acquire()
                job = self.jobs.pop(0)
                if job is None:
                    self.frames = self.pipe = self.predictor = None
                    self.done.release()
                    return
                kind, arg = job
                if kind == "steps":      # arg: shared iterator of step indices (workers pull, so no thread idles early)
                    while True:
                        with steps_lock:
                            i = next(arg, None)
                        if i is None:
                            break
                        self.step(i)
                    self.pipe.sync()
                    if self.workload == "unet":
                        torch.cuda.synchronize()
                elif kind == "prof":
                    if arg == "on":
                        _lib.prof_reset()
                        _lib.prof_enable(True)
                        if self.predictor is not None:
                            self.predictor.forward_ms = []
                    else:
                        _lib.prof_enable(False)
                        self.report = _lib.prof_report()
                        if self.predictor is not None:
                            self.unet_ms = list(self.predictor.forward_ms or [])
                            self.predictor.forward_ms = None
                self.done.release()

        def step(self, i):
            pipe = self.pipe
            if self.host is not None:
                h = self.host[i % 2]
                _lib.check(lib.tip_memcpy_h2d(_lib.dptr(self.frames[i % 2].ptr), ctypes.c_void_p(h.data_ptr()),
                                              ctypes.c_size_t(h.numel() * 2)))
            pipe.project(self.frames[i % 2])
            if self.workload == "classical":
                pipe.segment(0)
                pipe.cell_tables()
            elif self.workload == "unet":
                lab, _ = pipe.segment_unet(self.predictor)
                pipe.cell_tables(labels_ptr=lab.data_ptr(), shape=(X, Y))

        def submit(self, job):
            self.jobs.append(job)
            self.todo.release()

        def wait(self):
            self.done.acquire()
            if self.error is not None:
                raise RuntimeError("bench worker %d failed: %r" % (self.wid, self.error))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run_leg(workload, nthreads, steps, warmup):
        """warmup untimed steps, then exactly `steps` timed steps between barriers; then an isolated pass (ONE frame in
        flight) whose per-kernel HIP-event durations feed the roofline."""
        workers = [Worker(w, workload) for w in range(nthreads)]

        def run_steps(n, ws=workers):
            it = iter(range(n))
            for w in ws:
                w.submit(("steps", it))
            for w in ws:
                w.wait()

        def all_workers(job):
            for w in workers:
                w.submit(job)
            for w in workers:
                w.wait()

        def progress(msg):          # (stderr: a long U-Net leg must not look hung to whoever watches the run)
            if rank == 0:
                print("bench[%s]: %s" % (workload, msg), file=sys.stderr, flush=True)

        progress("warm-up (%d frames in flight)" % nthreads)
        run_steps(max(warmup, nthreads))
        barrier()
        progress("timed region: %d steps" % steps)
        # The timed region runs the product path as a caller would: no per-kernel events (two hipEventRecord per launch cost
        # the four-frame pipeline 3-5 % of its rate: 275 against 284 frames/s at 64 steps).  The same K steps are then repeated
        # with the events on -- same threads, same frames in flight -- for `roofline_timed_region`.
        t0 = time.perf_counter()
        run_steps(steps)
        barrier()
        elapsed = time.perf_counter() - t0
        progress("%.1f frames/s; the same steps again with per-kernel events" % (steps / elapsed))
        all_workers(("prof", "on"))
        barrier()
        t1 = time.perf_counter()
        run_steps(steps)
        barrier()
        elapsed_events = time.perf_counter() - t1
        all_workers(("prof", "off"))
        timed_reports = [dict(w.report) for w in workers]
        unet_ms = [m for w in workers for m in w.unet_ms]
        # isolated pass: with several frames in flight kernels of different frames share the chip and every per-kernel
        # duration is inflated
        iso_steps = min(steps, 5)
        progress("isolated pass: %d steps, one frame in flight" % iso_steps)
        workers[0].submit(("prof", "on")); workers[0].wait()
        run_steps(iso_steps, workers[:1])
        workers[0].submit(("prof", "off")); workers[0].wait()
        iso_report = dict(workers[0].report)
        iso_unet_ms = list(workers[0].unet_ms)
        all_workers(None)
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return dict(elapsed=elapsed, elapsed_events=elapsed_events, timed=timed_reports, iso=iso_report, iso_steps=iso_steps, unet_ms=unet_ms,
                    iso_unet_ms=iso_unet_ms, nthreads=nthreads, steps=steps, warmup=warmup)

    def merge(reports):
        rep = {}
        for r in reports:
            for k, (cnt_k, ms_k) in r.items():
                c0, m0 = rep.get(k, (0, 0.0))
                rep[k] = (c0 + cnt_k, m0 + ms_k)
        return rep

    def roofline_of(rep, nframes):
        """The dominant kernel = the one with the largest time PER FRAME (all its launches of a frame added up).  Its
        algorithmic bytes are per-frame figures, so achieved = bytes per frame / its time per frame -- for a
        kernel launched once per frame that is bytes / launch duration; for the watershed's tile kernel (launched ~20 times
        per frame over the same 50 MB job) it is NOT bytes / one launch."""
        total_kernel_ms = sum(v[1] for v in rep.values())
        name, (cnt, ms) = max(rep.items(), key=lambda kv: kv[1][1])
        per_frame_s = ms / nframes / 1e3
        launches_per_frame = cnt / float(nframes)
        ab = algorithmic_bytes(name, C, Z, Y, X)
        tr = pmc_traffic(name)
        roof = {"kernel": name, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                "traffic": tr * launches_per_frame if tr else None, "traffic_per_launch": tr,
                "avg_launch_ms": ms / cnt, "launches_per_frame": launches_per_frame, "ms_per_frame": ms / nframes,
                "share_of_kernel_time": ms / total_kernel_ms if total_kernel_ms else None}
        if ab:
            roof["achieved"] = ab / per_frame_s / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
            roof["algorithmic_bytes"] = ab
            if tr:
                roof["traffic_ratio"] = tr * launches_per_frame / ab
        ops = algorithmic_dp_ops(name, Z, Y, X)
        if ops:
            tf = ops / per_frame_s / 1e12
            roof["valu_fp64"] = {"achieved": tf, "peak": FP64_VALU_PEAK_TF / 2, "unit": "Tinstr-lanes/s",
                                 "frac": tf / (FP64_VALU_PEAK_TF / 2)}
        return roof

    def kernel_table(rep, nsteps):
        out_k = {}
        for k, v in rep.items():
            out_k[k] = {"n_per_step": round(v[0] / nsteps, 2), "ms_per_step": round(v[1] / nsteps, 4)}
            kb = algorithmic_bytes(k, C, Z, Y, X)
            if kb:
                gbs = kb / (v[1] / nsteps / 1e3) / 1e9     # per-frame bytes / per-frame time of this kernel
                out_k[k]["hbm_GBps"] = round(gbs, 1)
                out_k[k]["hbm_frac"] = round(gbs / HBM_PEAK_GBS, 4)
        return out_k

    def unet_summary(leg):
        """MFMA roofline of the U-Net forward pass: dense 3x3-conv flops of pl.py:31-72 at the padded size / the forward
        pass's duration (torch CUDA events on torch's stream around the network only)."""
        from tissue_image_processing_amd.prediction_local import _UNet, find_desired_shape
        hp, wp = find_desired_shape(X, Y)
        flops = _UNet.flops(None, hp, wp)
        ms = leg["iso_unet_ms"] or leg["unet_ms"]
        fwd_ms = float(np.median(ms)) if ms else None
        dt = os.environ.get("TISSUE_HIP_UNET_DTYPE", "fp32")
        peak = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0}[dt]
        r = {"bound": "mfma", "dtype": dt, "flops_per_frame": flops, "peak": peak, "unit": "TFLOP/s",
             "forward_ms": fwd_ms, "achieved": None, "frac": None,
             "measured": "torch CUDA events around the network's forward pass (MIOpen convolutions), median of %d" % len(ms)}
        if fwd_ms:
            r["achieved"] = flops / (fwd_ms / 1e3) / 1e12
            r["frac"] = r["achieved"] / peak
        return r

    wl_names = {
        "projection": "surface_projection",
        "classical": "surface_projection+watershed_segmentation+cell_tables",
        "unet": "surface_projection+unet_segmentation(%s,random-init,head bias calibrated to 50%% foreground)+threshold/closing/watershed tail+cell_tables"
                % os.environ.get("TISSUE_HIP_UNET_DTYPE", "fp32")}

    # U-Net variant: three frames in flight -- the network saturates the chip on its own, but the tail's watershed has a
    # sequential host stage (the heap-order recurrence of mode B) and latency-bound generations that the other frames'
    # convolutions hide (measured: 5.10 frames/s with two, 5.29 with three)
    unet_threads = max(1, min(int(os.environ.get("TIP_BENCH_UNET_INFLIGHT", "3")), args.inflight))
    nthreads = max(1, min(args.inflight, args.steps)) if workload != "unet" else min(unet_threads, args.steps)
    leg = run_leg(workload, nthreads, args.steps, args.warmup)
    unet_leg = None
    if workload == "classical" and not args.no_unet_leg and not args.include_upload:
        unet_leg = run_leg("unet", min(unet_threads, max(1, args.unet_steps)), max(1, args.unet_steps), max(2, unet_threads))
    del st, st_flip

    if rank == 0:
        elapsed = leg["elapsed"]
        roof = roofline_of(leg["iso"], leg["iso_steps"])
        roof["measured"] = ("HIP events on the library stream, isolated pass of %d steps with ONE frame in flight run right "
                            "after the timed region (kernels of concurrent frames share the chip in the timed region)"
                            % leg["iso_steps"])
        roof_timed = roofline_of(merge(leg["timed"]), args.steps)
        roof_timed["measured"] = ("HIP events on the library streams over a repeat of the timed region's %d steps with the same %d frames "
                                  "in flight (%.1f frames/s with the events on; the timed region itself runs without them)"
                                  % (args.steps, leg["nthreads"], world * args.steps / leg["elapsed_events"]))
        kernels = kernel_table(leg["iso"], leg["iso_steps"])
        # the heaviest ARITHMETIC kernels (the sigma-30 score passes) are bound by the FP32 matrix pipe, not by HBM: the
        # banded-Toeplitz MFMA tiles issue 2 * (32 + 2 * 120) flop per voxel and pass (241 of the 272 products of an output
        # are non-zero taps; "achieved" counts the issued flops, "useful" the 2 * 241 of a direct correlation); the dense
        # FP32 MFMA peak equals the FP32 vector peak (256 flop / clk / CU)
        valu = {}
        for kname in ("score_fast_y", "score_fast_x"):
            if kname in leg["iso"] and leg["iso"][kname][0]:
                cnt_k, ms_k = leg["iso"][kname]
                sec = ms_k / cnt_k / 1e3
                tf = Z * Y * X * 544.0 / sec / 1e12
                valu[kname] = {"bound": "mfma_fp32", "achieved": tf, "useful": Z * Y * X * 482.0 / sec / 1e12,
                               "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP32_VALU_PEAK_TF,
                               "avg_launch_ms": ms_k / cnt_k,
                               "note": "default build (TIP_FAST_CFG unset): matrix-core kernels; with a VALU configuration the issued-flop figure does not apply"}
        out = {
            "metric": "frames/sec end-to-end (2048^2, z=30)", "value": world * args.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if workload != "unet" else "f32",
            "data": "synthetic",
            "config": {"workload": "%dx%dx%d_c%d_u16:%s" % (Y, X, Z, C, wl_names[workload]),
                       "value_is": workload, "frames_per_step": world, "frames_in_flight_per_gpu": leg["nthreads"],
                       "includes_h2d_upload": bool(args.include_upload),
                       "parallelism": "frame-sharded dp%d, no data-path collective" % world},
            "roofline": roof, "roofline_timed_region": roof_timed, "roofline_valu": valu, "kernels": kernels,
        }
        if workload == "unet":
            out["roofline_unet"] = unet_summary(leg)
        if unet_leg is not None:
            ue = unet_leg["elapsed"]
            out["unet"] = {
                "workload": "%dx%dx%d_c%d_u16:%s" % (Y, X, Z, C, wl_names["unet"]),
                "value": world * unet_leg["steps"] / ue, "unit": "frames/s", "steps": unet_leg["steps"],
                "warmup": unet_leg["warmup"], "ms_per_step": 1e3 * ue / unet_leg["steps"], "dtype": "f32",
                "frames_in_flight_per_gpu": unet_leg["nthreads"], "roofline": unet_summary(unet_leg),
                "kernels": kernel_table(unet_leg["iso"], unet_leg["iso_steps"]),
                "note": "BASELINE config 3 as written; timed right after the classical leg in the same process"}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
