#!/usr/bin/env python3
"""bench.py -- end-to-end frames/s of the per-frame confocal-stack hot path on MI355X.

One "step" = one pass of the hot path over one synthetic 2048x2048x30 (C=2, uint16) frame per GPU that is already resident
in HBM.  One process per GPU; frames are independent units, so N GPUs process N frames per step with no data-path
collective (weak scaling).  Rank 0 prints ONE JSON line.

`value` (--workload auto) is BASELINE.json's config 3 as written: surface projection (sp.py:17-85) -> U-Net segmentation
(pl.py:124-199: prepare, network, threshold / closing / erosion / boundary, two-valued watershed) -> cell tables
(ti.py:880-909).  The classical variant of the same frame (projection -> watershed_segmentation (bim.py:446-476) -> cell
tables) is timed in the same process and reported under "classical" (`--workload classical` makes it the headline).
Both legs also report `value_with_pcie`: the same steps with every frame uploaded from pinned host memory and the
projection + label map fetched back, i.e. what the drop-in functions (host arrays in and out) sustain.

`python bench.py --gpus N` without a launcher environment starts N ranks itself (one process per GPU, RCCL process group
on 127.0.0.1) and fails if fewer than N devices are visible; under torchrun it is one of the ranks.
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


def cpu_baseline_worker(Ys, Xs, Z, workload, threads, seg=0):
    """One CPU-baseline sample: the C/numpy oracle (a port of the reference path, single thread like scipy.ndimage /
    skimage).  The projection runs on a Ys x Xs frame (the single-process record: the WHOLE frame), the segmentation stages on
    its top-left seg x seg corner (0: all of it), reported as measured -- the caller scales by pixel count and says so.  For the
    U-Net workload the network itself is the SAME float32 network run by torch on the host cores (`threads` of them) on a 512^2
    crop -- TensorFlow-CPU's role in the reference -- and the tail is the oracle's restatement of pl.py:167-194 (101 closings,
    as upstream).  Prints the seconds of every stage."""
    from oracle import oracle as orc
    from tissue_image_processing_amd import synthetic
    st = synthetic.make_stack(Z, Ys, Xs, seed=1234)
    t0 = time.perf_counter()
    proj, zmap = orc.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
    t_proj = time.perf_counter() - t0
    t_seg = t_fwd = 0.0
    fwd_px = 0
    sy, sx = (min(seg, Ys), min(seg, Xs)) if seg else (Ys, Xs)
    if workload == "classical":
        crop = np.ascontiguousarray(proj[0][:sy, :sx])
        t1 = time.perf_counter()
        lab = orc.watershed_segmentation(crop, 0.03, 3, 3)
        orc.frame_cellinfo(lab)
        t_seg = (time.perf_counter() - t1) * (Ys * Xs) / float(sy * sx)    # scaled to the projection's frame
    elif workload == "unet":
        import torch
        from tissue_image_processing_amd.prediction_local import _UNet
        torch.set_num_threads(max(1, threads))
        net = _UNet(2, "cpu", dtype=torch.float32)
        n = 512
        x = torch.from_numpy(np.stack([proj[1][:n, :n].T, proj[0][:n, :n].T])[None].astype(np.float32))
        x = x / max(float(x.max()), 1.0)
        net.calibrate_head(x, 0.5)
        t1 = time.perf_counter()
        p = net.forward(x)
        t_fwd = time.perf_counter() - t1
        fwd_px = n * n
        t1 = time.perf_counter()
        p0 = np.ascontiguousarray(p[0, 0].numpy())                        # float32, as Keras hands it to pl.py:168's `> 0.1`
        lab, hc = orc.closing_tail(p0)[:2]
        orc.frame_cellinfo(lab)
        t_seg = (time.perf_counter() - t1) * (Ys * Xs) / float(fwd_px)     # the tail ran on the network's crop
    keep = os.environ.get("TIP_BENCH_PARITY_OUT")
    if keep:          # what this sample computed, for the parent's `label_parity` record (the GPU path on the same frame, compared after the timed legs)
        extra = {"p0": p0, "hc": hc} if workload == "unet" else {}
        try:
            np.savez(keep, proj=proj, zmap=zmap, lab=lab, **extra)
        except OSError as e:          # (no parity record then; the baseline figure does not depend on it)
            print("bench.py: could not keep the cpu_baseline outputs: %r" % (e,), file=sys.stderr)
    print("CPU_BASELINE_SECONDS %.6f %.6f %.6f %d" % (t_proj, t_seg, t_fwd, fwd_px))


def cpu_baseline(sample_yx, Z, workload, nproc, threads=1, seg=0, keep=None):
    """Runs the sample in `nproc` child processes at once (started BEFORE this process touches the GPU) and returns the
    list of per-process (projection s, segmentation s, network s, network pixels): nproc = 1 is the single-process figure,
    nproc = host cores the embarrassingly parallel "N frames on N processes" one (scipy.ndimage / skimage are
    single-threaded)."""
    import subprocess
    Ys, Xs = sample_yx
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(Ys), str(Xs), str(Z), workload, str(threads), str(seg)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS=str(threads))
    env.pop("TIP_BENCH_PARITY_OUT", None)
    if keep and nproc == 1:
        env["TIP_BENCH_PARITY_OUT"] = keep
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True) for _ in range(nproc)]
    secs = []
    for p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("cpu baseline worker failed")
        f = [l for l in out.splitlines() if l.startswith("CPU_BASELINE_SECONDS")][0].split()
        secs.append((float(f[1]), float(f[2]), float(f[3]), int(f[4])))
    return secs


def cpu_baseline_record(Y, X, Z, workload):
    """cpu_baseline object of the JSON line for `workload` (classical / projection / unet): the projection on ONE WHOLE frame
    (~16 s on one core at 2048^2 x 30), the segmentation + tables on its 1024^2 corner and (U-Net workload) the network on a
    512^2 crop, both scaled by pixel count -- a bounded sample, stated in `sample`."""
    SEG = 1024
    host_cores = os.cpu_count() or 1
    try:
        host_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    threads = max(1, min(host_cores, 32)) if workload == "unet" else 1

    def frame_seconds(rec, scale):
        t_proj, t_seg, t_fwd, fwd_px = rec
        return (t_proj + t_seg) * scale + (t_fwd * (Y * X) / float(fwd_px) if fwd_px else 0.0)

    import tempfile
    try:
        keep = os.path.join(tempfile.mkdtemp(prefix="tip_bench_"), "cpu_%s.npz" % workload)
    except OSError:
        keep = None
    one = cpu_baseline((Y, X), Z, workload, 1, threads, seg=SEG, keep=keep)[0]
    sec = frame_seconds(one, 1.0)
    seg_scale = (Y * X) / float(min(SEG, Y) * min(SEG, X))
    rec = {"value": 1.0 / sec, "unit": "frames/s", "cores": threads, "kind": "port", "host_cores": host_cores, "workload": workload,
           "sample": "one whole %dx%dx%d frame through the C/numpy oracle's projection (%.1f s on one core, unscaled); segmentation + "
                     "tables of the %s path on its %dx%d corner, scaled x%g by pixel count to %.1f s" % (Y, X, Z, one[0], workload, min(SEG, Y), min(SEG, X), seg_scale, one[1])}
    rec["_outputs"] = keep          # (internal: consumed by label_parity_record, not printed)
    if workload == "unet":
        rec["sample"] += ("; network = the same float32 U-Net through torch-CPU on %d threads, 512^2 crop %.1f s scaled x%g "
                          "(%.2f TFLOP/s)" % (threads, one[2], (Y * X) / float(one[3]),
                                              19.8 * one[3] / (2048.0 * 2048.0) / max(one[2], 1e-9)))
        rec["stages_s_per_frame"] = {"projection": one[0], "tail_and_tables": one[1],
                                     "network": one[2] * (Y * X) / float(one[3])}
    else:
        Ys, Xs = min(Y, 1024), min(X, 1024)
        scale = (Y * X) / float(Ys * Xs)
        nproc = max(1, min(host_cores, 16))
        many = cpu_baseline((Ys, Xs), Z, workload, nproc)
        slowest = max(frame_seconds(m, scale) for m in many)
        rec["n_process"] = {"processes": nproc, "value": nproc / slowest, "unit": "frames/s",
                            "sample": "a %dx%dx%d crop (1/%g of a frame) in %d processes at once, scaled by pixel count: slowest %.1f s per frame"
                                      % (Ys, Xs, Z, scale, nproc, slowest)}
    return rec


def _label_iou(test, ref):
    """mean over the reference's labels of the IoU with the test label that covers most of it (1.0 for identical maps)"""
    if np.array_equal(test, ref):
        return 1.0
    ious = []
    for l in np.unique(ref):
        if l == 0:
            continue
        m = ref == l
        cand = np.bincount(test[m].clip(min=0))
        cand[0] = 0
        if cand.sum() == 0:
            ious.append(0.0)
            continue
        k = cand.argmax()
        ious.append((m & (test == k)).sum() / float((m | (test == k)).sum()))
    return float(np.mean(ious)) if ious else 1.0


def label_parity_record(cpu, Y, X, Z, device_index):
    """BASELINE.json's metric has a second half, "label IoU vs ref": the GPU path on the SAME synthetic frame the cpu_baseline leg's oracle
    processed (seed 1234), compared with what that leg computed -- projection and z-map bit for bit, the classical label map of the 1024^2
    corner and the U-Net tail's label map / HC map on the oracle's probability crop (bit-exact flag + IoU).  Runs after the timed legs.
    The network itself (U2) has no reference to compare with (no TensorFlow, no trained weights): stated, not measured."""
    from tissue_image_processing_amd import synthetic, surface_projection as sp, basic_image_manipulations as bim
    out = {"reference": "the cpu_baseline leg's outputs (C/numpy oracle, itself pinned by reference-generated goldens) on the same synthetic "
                        "%dx%dx%d frame (seed 1234)" % (Y, X, Z)}
    st = synthetic.make_stack(Z, Y, X, seed=1234)
    proj, zmap = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
    recs = [(cpu.get("workload"), cpu.get("_outputs"))] + ([("classical", cpu["classical"].get("_outputs"))] if isinstance(cpu.get("classical"), dict) else [])
    for wl, path in recs:
        if not path or not os.path.exists(path):
            continue
        z = np.load(path)
        if "projection_bit_exact" not in out:
            out["projection_bit_exact"] = bool(np.array_equal(proj, z["proj"]))
            out["projection_max_abs_diff"] = float(np.abs(proj - z["proj"]).max())
            out["zmap_bit_exact"] = bool(np.array_equal(zmap, z["zmap"]))
        ref = z["lab"]
        if wl == "classical":
            sy, sx = ref.shape
            lab = bim.watershed_segmentation(np.ascontiguousarray(z["proj"][0][:sy, :sx]), 0.03, 3, 3)
            out["classical_labels"] = {"bit_exact": bool(np.array_equal(lab, ref)), "iou": _label_iou(lab, ref), "labels": int(ref.max()),
                                       "pixels": int(ref.size), "what": "watershed_segmentation(0.03, 3, 3) of the projection's %dx%d corner" % (sy, sx)}
        elif wl == "unet":
            import torch
            from tissue_image_processing_amd.prediction_local import SegmentationPredictor
            p0 = torch.from_numpy(z["p0"]).to(torch.device("cuda", device_index))
            pred = SegmentationPredictor(None, (2,) + tuple(z["p0"].shape), device=device_index)
            lab, hc = pred.segment_probability(p0)
            out["unet_tail_labels"] = {"bit_exact": bool(np.array_equal(lab, ref)), "iou": _label_iou(lab, ref), "hc_bit_exact": bool(np.array_equal(hc, z["hc"])),
                                       "labels": int(ref.max()), "pixels": int(ref.size),
                                       "what": "threshold / 101 closings / erosion / boundary / watershed (pl.py:167-194) on the %dx%d probability map of the "
                                               "cpu_baseline leg's float32 network" % tuple(z["p0"].shape)}
            out["unet_network"] = "unpinned: no TensorFlow and no trained weights here or upstream; kernels vs the float64 evaluation of the same layers: tests"
        try:
            os.remove(path)
            os.rmdir(os.path.dirname(path))
        except OSError:
            pass
    return out


# ---- --gpus N without a launcher: this process starts the N ranks ------------------------------------------------------------
def launch_ranks(args, argv):
    """Starts N = --gpus child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, as torchrun would)
    BEFORE this process touches the GPU, and exits non-zero when fewer devices are visible.  Rank 0's stdout (the ONE JSON
    line) is this process's stdout; the exit code is the worst child's."""
    import socket
    import subprocess
    n = args.gpus
    if os.environ.get("TIP_BENCH_STUB") != "1":
        import torch
        have = torch.cuda.device_count()        # (counting devices does not initialise the GPU)
        if have < n:
            print("bench.py: --gpus %d asked for, %d HIP device(s) visible" % (n, have), file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # poll all of them: when one rank dies (device missing, import error) the others would sit in init_process_group / a barrier
    # until the collective timeout -- they are terminated and the failing rank's code is returned
    worst = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and worst == 0:
                worst = rc
                for q in live:
                    q.terminate()
        if live:
            time.sleep(0.05)
    return worst


def stub_rank(args, rank, world):
    """TIP_BENCH_STUB=1: the launcher / process-group / barrier / max-over-ranks plumbing with no GPU behind it (gloo on
    the CPU, a step is a short sleep) -- what the CPU test of `--gpus N` runs.  Labelled as such in the line it prints."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=2))
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002)
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    seen = torch.tensor([1.0])
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"metric": "frames/sec end-to-end (2048^2, z=30)", "value": world * args.steps / float(el), "unit": "frames/s",
                          "n_gpus": world, "ranks_seen": int(seen.item()), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * float(el) / args.steps, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "stub (no GPU: launcher plumbing only)",
                          "config": {"workload": "stub"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


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
                                    torch.device("cuda", local_rank) if world > 1 else "cpu", estimate_drift=True,
                                    block_frames=max(1, args.inflight))     # rounds: the exchange of one overlaps the next one's kernels
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    backend.close()
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
        return cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]),
                                   int(sys.argv[7]) if len(sys.argv) > 7 else 0)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps of the headline leg (default: 30 for the U-Net workload, 64 for classical / projection, "
                         "16 for --workload movie, whose every frame is a distinct synthetic stack kept in pinned memory)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 6 for the U-Net workload, else 8)")
    ap.add_argument("--size", type=int, nargs=3, default=[2048, 2048, 30], metavar=("Y", "X", "Z"))
    ap.add_argument("--workload", default="auto", choices=["auto", "projection", "classical", "unet", "movie"],
                    help="auto = unet = BASELINE config 3 as written (the classical variant rides along as a secondary leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary-leg", "--no-unet-leg", dest="no_secondary", action="store_true",
                    help="skip the secondary leg (classical under the U-Net headline and vice versa)")
    ap.add_argument("--secondary-steps", type=int, default=None, help="timed steps of the secondary leg (default 64 classical / 9 U-Net)")
    ap.add_argument("--no-pcie-leg", action="store_true", help="skip the value_with_pcie repeat (host upload + output fetch per frame)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="frames in flight per GPU: host threads, each with its own HIP stream and workspaces (the library "
                         "is re-entrant per thread, like the reference's Qt workers); frames are independent units")
    args = ap.parse_args()
    workload = "unet" if args.workload == "auto" else args.workload
    if args.steps is None:
        args.steps = {"movie": 16, "unet": 30}.get(workload, 64)     # (1.6 s of timed region: run-to-run spread of a 16-step region was 3 %)
    if args.warmup is None:
        args.warmup = 6 if workload == "unet" else 8
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and "RANK" in os.environ and args.gpus > 1:
        print("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world), file=sys.stderr)
        return 2
    if os.environ.get("TIP_BENCH_STUB") == "1":
        return stub_rank(args, rank, world)
    Y, X, Z = args.size
    C = 2

    # CPU baseline first: its child processes are started before this process initialises the GPU
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and workload != "movie":
        cpu = cpu_baseline_record(Y, X, Z, workload)
        if workload == "unet" and not args.no_secondary:
            cpu["classical"] = cpu_baseline_record(Y, X, Z, "classical")

    # MIOpen's exhaustive find ran its naive reference convolutions for ~3 minutes per process (profiles/r02e_*): the fast
    # find mode picks the same implicit-GEMM solvers in seconds
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
    import torch
    import torch.distributed as dist
    if torch.cuda.device_count() <= local_rank:
        print("bench.py: rank %d needs device %d, %d visible" % (rank, local_rank, torch.cuda.device_count()), file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    # TIP_BENCH_FORCE_DIST=1 (with torchrun --nproc-per-node 1) walks the RCCL process-group path on a one-GPU box
    use_dist = world > 1 or (os.environ.get("TIP_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout carries the ONE JSON line, so
        # file descriptor 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            import datetime
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(minutes=5))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from tissue_image_processing_amd import _lib, synthetic
    from tissue_image_processing_amd.pipeline import FramePipeline
    _lib.init(local_rank)
    if workload == "movie":
        return bench_movie(args, rank, local_rank, world, dist, torch)
    lib = _lib.lib()
    from tissue_image_processing_amd import prediction_local as plm

    # synthetic frames, resident in HBM before the timed region (two distinct frames per rank, alternated)
    import threading
    st = synthetic.make_stack(Z, Y, X, seed=100 + rank)
    st_flip = np.ascontiguousarray(st[:, :, :, ::-1])
    host_frames = [torch.from_numpy(st).pin_memory(), torch.from_numpy(st_flip).pin_memory()]   # the PCIe leg uploads these
    steps_lock = threading.Lock()

    class Worker(object):
        """One frame in flight: own thread, own HIP stream / workspace pool (tip_init per thread), own torch stream for the
        network, own resident buffers."""

        def __init__(self, wid, workload):
            self.wid = wid
            self.workload = workload
            self.todo = threading.Semaphore(0)
            self.done = threading.Semaphore(0)
            self.jobs = []
            self.report = {}
            self.error = None
            self.unet_ms = []
            self.pcie = False
            self.diag = []           # U-Net leg: (watershed flags, markers) of every step since the last reset
            self.thread = threading.Thread(target=self.run, daemon=True)
            self.thread.start()
            self.wait()  # wait for setup

        def run(self):
            try:
                if self.workload == "unet":      # the threads do not share torch's default stream: no thread waits for another's network
                    with torch.cuda.stream(torch.cuda.Stream(device=local_rank)):
                        self.run_inner()
                else:
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
            # pinned landing buffers of the PCIe leg: the projection (C, Y, X) float64 and the label map int32
            self.out_proj = torch.empty((C, Y, X), dtype=torch.float64).pin_memory()
            self.out_lab = torch.empty((Y, X), dtype=torch.int32).pin_memory()
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
                        torch.cuda.current_stream().synchronize()
                elif kind == "pcie":
                    self.pcie = bool(arg)
                elif kind == "diag":
                    self.diag = []
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
            if self.pcie:
                h = host_frames[i % 2]
                _lib.check(lib.tip_memcpy_h2d(_lib.dptr(self.frames[i % 2].ptr), ctypes.c_void_p(h.data_ptr()),
                                              ctypes.c_size_t(h.numel() * 2)))
            pipe.project(self.frames[i % 2])
            lab_ptr = None
            if self.workload == "classical":
                pipe.segment(0)
                pipe.cell_tables()
                lab_ptr = pipe.d_labels.ptr
            elif self.workload == "unet":
                lab, _ = pipe.segment_unet(self.predictor)
                if getattr(self.predictor.model, "last_mode", None) != unet_mode:       # a silent fall-back to MIOpen must not be timed under this label
                    raise RuntimeError("the U-Net step ran in mode %r, the bench line says %r" % (getattr(self.predictor.model, "last_mode", None), unet_mode))
                self.diag.append((int(self.predictor.last_flags), int(self.predictor.last_markers)))
                pipe.cell_tables(labels_ptr=lab.data_ptr(), shape=(X, Y))
                lab_ptr = lab.data_ptr()
            if self.pcie:        # what the drop-in functions hand back: the (C, Y, X) float64 projection and the int32 label map
                _lib.check(lib.tip_memcpy_d2h(ctypes.c_void_p(self.out_proj.data_ptr()), _lib.dptr(pipe.d_proj.ptr),
                                              ctypes.c_size_t(C * Y * X * 8)))
                if lab_ptr is not None:
                    _lib.check(lib.tip_memcpy_d2h(ctypes.c_void_p(self.out_lab.data_ptr()), _lib.dptr(lab_ptr), ctypes.c_size_t(Y * X * 4)))

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

    def diag_summary(workers):
        """U-Net leg: what the watershed of every step since the last reset reported.  Every boundary image must be
        two-valued (mode B, exact); a serial stage or a tie flag here would mean a corrupted intermediate."""
        d = [x for w in workers for x in w.diag]
        if not d:
            return None
        fl = [f for f, _ in d]
        return {"steps": len(d), "flags_seen": sorted(set(f & 0xff for f in fl)),
                "all_two_valued": all((f & _lib.WS_FLAG_TWO_VALUED) and not (f & (_lib.WS_FLAG_SERIAL_EXACT | _lib.WS_FLAG_SERIAL_FINISH)) for f in fl),
                "serial_finish_pixels": int(sum(f >> _lib.WS_FLAG_COUNT_SHIFT for f in fl)),
                "markers_min": int(min(m for _, m in d)), "markers_max": int(max(m for _, m in d))}

    def run_leg(workload, nthreads, steps, warmup, pcie_steps):
        """warmup untimed steps, then exactly `steps` timed steps between barriers; the same steps again with per-kernel
        events; an isolated pass (ONE frame in flight) whose per-kernel HIP-event durations feed the roofline; and the
        PCIe-inclusive repeat."""
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

        diag = {}
        progress("warm-up (%d frames in flight)" % nthreads)
        run_steps(max(warmup, nthreads))
        barrier()
        diag["warmup"] = diag_summary(workers)
        all_workers(("diag", None))
        progress("timed region: %d steps" % steps)
        # The timed region runs the product path as a caller would: no per-kernel events (two hipEventRecord per launch cost
        # the four-frame pipeline 3-5 % of its rate).  The same K steps are then repeated with the events on -- same threads,
        # same frames in flight -- for `roofline_timed_region`.
        t0 = time.perf_counter()
        run_steps(steps)
        barrier()
        elapsed = time.perf_counter() - t0
        diag["timed"] = diag_summary(workers)
        all_workers(("diag", None))
        progress("%.2f frames/s; the same steps again with per-kernel events" % (steps / elapsed))
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
        diag["instrumented"] = diag_summary(workers)
        all_workers(("diag", None))
        pcie_elapsed = None
        if pcie_steps:
            progress("PCIe-inclusive repeat: %d steps (upload of every frame from pinned memory, projection + label map fetched)" % pcie_steps)
            all_workers(("pcie", True))
            run_steps(nthreads)
            barrier()
            t2 = time.perf_counter()
            run_steps(pcie_steps)
            barrier()
            pcie_elapsed = time.perf_counter() - t2
            all_workers(("pcie", False))
            diag["pcie"] = diag_summary(workers)
        all_workers(None)
        if use_dist:
            t = torch.tensor([elapsed, pcie_elapsed or 0.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0].item())
            pcie_elapsed = float(t[1].item()) if pcie_steps else None
        return dict(workload=workload, elapsed=elapsed, elapsed_events=elapsed_events, timed=timed_reports, iso=iso_report,
                    iso_steps=iso_steps, unet_ms=unet_ms, iso_unet_ms=iso_unet_ms, nthreads=nthreads, steps=steps, warmup=warmup,
                    pcie_steps=pcie_steps, pcie_elapsed=pcie_elapsed, diag=diag)

    def merge(reports):
        rep = {}
        for r in reports:
            for k, (cnt_k, ms_k) in r.items():
                c0, m0 = rep.get(k, (0, 0.0))
                rep[k] = (c0 + cnt_k, m0 + ms_k)
        return rep

    def roofline_of(rep, nframes):
        """The dominant LIBRARY kernel = the one with the largest time PER FRAME (all its launches of a frame added up).  Its
        algorithmic bytes are per-frame figures, so achieved = bytes per frame / its time per frame -- for a
        kernel launched once per frame that is bytes / launch duration; for the watershed's tile kernel (launched ~16 times
        per frame over the same 50 MB job) it is NOT bytes / one launch."""
        if not rep:
            return None
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

    class _FlopsShape(object):          # (flops() only needs the widths)
        filters, bottleneck = plm._FILTERS, 1024

    def unet_roofline(leg):
        """MFMA roofline of the U-Net forward pass, the dominant kernel group of config 3: dense conv flops of pl.py:31-72 at
        the padded size / the forward pass's duration (events on the stream the network is launched on, around the network
        only, one frame in flight)."""
        hp, wp = plm.find_desired_shape(X, Y)
        flops = plm._UNet.flops(_FlopsShape(), hp, wp)
        ms = leg["iso_unet_ms"] or leg["unet_ms"]
        fwd_ms = float(np.median(ms)) if ms else None
        info = plm.unet_arithmetic()
        r = {"kernel": "unet_forward", "bound": "mfma", "dtype": info["dtype"], "arithmetic": info["arithmetic"],
             "flops_per_frame": flops, "peak": info["peak_tflops"], "unit": "TFLOP/s", "forward_ms": fwd_ms,
             "achieved": None, "frac": None, "traffic": None,
             "measured": "events around the network's forward pass on its stream, isolated pass (one frame in flight), median of %d" % len(ms)}
        # HBM bytes of the network's kernels from the committed counter passes (FETCH_SIZE x 2 + WRITE_SIZE over one forward pass)
        import glob
        tr_paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_unet_pmc_traffic.json")))
        if tr_paths and (hp, wp) == (2048, 2048) and info.get("issued_flops_factor", 1) == 3:
            recs = json.load(open(tr_paths[-1]))
            r["traffic_note"] = ("rocprofv3 --pmc counters of a separate, committed run of this network (profiles/%s), not of this run: "
                                 "a cross-reference" % os.path.basename(tr_paths[-1]))
            tot = sum((v["fetch_MB_per_call_x2corrected"] + v["write_MB_per_call"]) * v["calls"] for k, v in recs.items() if k.startswith("k_unet_"))
            # forward passes in that counter run: the first layer is launched once per pass (the head is fused into the last convolution)
            per_pass = [v["calls"] for k, v in recs.items() if k.startswith("k_unet_conv_first")] or [v["calls"] for k, v in recs.items() if k.startswith("k_unet_head")]
            passes = max(1, min(per_pass)) if per_pass else 1
            r["traffic"] = tot * 1e6 / passes
            r["algorithmic_bytes"] = plm.unet_algorithmic_bytes(hp, wp)
            r["traffic_ratio"] = r["traffic"] / r["algorithmic_bytes"]
            r["traffic_source"] = os.path.basename(tr_paths[-1])
        if fwd_ms:
            r["achieved"] = flops / (fwd_ms / 1e3) / 1e12
            r["frac"] = r["achieved"] / info["peak_tflops"]
            if info.get("issued_flops_factor", 1) != 1:
                r["issued_tflops"] = r["achieved"] * info["issued_flops_factor"]
                r["issued_frac_of_bf16_peak"] = r["issued_tflops"] / info["issued_peak_tflops"]
                r["note"] = ("`peak` is the nominal dense 16-bit MFMA peak / the products per term; back-to-back 32x32x16 MFMAs from registers "
                             "with random operand bits sustain 1.70-1.75 PFLOP/s in fp16 and 1.85-1.94 in bf16 (68-78 % of 2.5) at a "
                             "power-limited 1.7-1.9 GHz on this chip (tools/ubench/mfma_f16_rate.hip, profiles/r04a_mfma_*_rate*.txt); "
                             "the board reads 1400 W, its cap, for the whole forward pass (tools/power_probe.py, profiles/r04n_unet_power.txt)")
        return r

    def unet_modes_record():
        """The network's forward pass in every arithmetic mode, timed in THIS run on one frame in flight (events on torch's stream,
        median of 3 after one warm-up): the headline mode next to the 16-significand-bit bf16x3 of round 3 and the six-product
        bf16x6 -- the float32-equivalent cross-reference of the headline's figure."""
        from tissue_image_processing_amd.prediction_local import SegmentationPredictor
        _lib.init(local_rank)
        pred = SegmentationPredictor(None, (2, X, Y), device=local_rank)
        xin = torch.rand((1, 2) + tuple(plm.find_desired_shape(X, Y)), device=torch.device("cuda", local_rank))
        keep = os.environ.get("TISSUE_HIP_UNET_ARITH")
        out = {}
        try:
            for mode in ("f16x3", "bf16x3", "bf16x6"):        # (MIOpen's float32 route: 181 ms when it has the device to itself, profiles/r02*)
                os.environ["TISSUE_HIP_UNET_ARITH"] = mode
                pred.model.forward(xin)
                ms = []
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    pred.model.forward(xin)
                    e1.record()
                    e1.synchronize()
                    ms.append(e0.elapsed_time(e1))
                info = dict(plm.unet_arithmetic())
                out[mode] = {"forward_ms": float(np.median(ms)), "ran_as": pred.model.last_mode,
                             "float32_equivalent": bool(info.get("float32_equivalent", mode == "miopen")),
                             "useful_tflops": pred.model.flops(int(xin.shape[2]), int(xin.shape[3])) / (float(np.median(ms)) / 1e3) / 1e12}
        finally:
            if keep is None:
                os.environ.pop("TISSUE_HIP_UNET_ARITH", None)
            else:
                os.environ["TISSUE_HIP_UNET_ARITH"] = keep
        return out

    unet_mode = plm._unet_mode()
    wl_names = {
        "projection": "surface_projection",
        "classical": "surface_projection+watershed_segmentation+cell_tables",
        "unet": "surface_projection+unet_segmentation(%s,random-init,head bias calibrated to 50%% foreground)+threshold/closing/watershed tail+cell_tables"
                % plm.unet_arithmetic()["dtype"]}

    # U-Net variant: three frames in flight -- the network saturates the chip on its own, but the tail's watershed has
    # latency-bound generations and a short host stage that the other frames' convolutions hide
    unet_threads = max(1, min(int(os.environ.get("TIP_BENCH_UNET_INFLIGHT", "3")), args.inflight))

    def threads_for(wl, steps):
        return max(1, min(unet_threads if wl == "unet" else args.inflight, steps))

    pcie_steps = 0 if args.no_pcie_leg else None
    main_pcie = pcie_steps if pcie_steps is not None else (6 if workload == "unet" else 24)
    leg = run_leg(workload, threads_for(workload, args.steps), args.steps, args.warmup, main_pcie)
    second = None
    if not args.no_secondary and workload in ("unet", "classical"):
        wl2 = "classical" if workload == "unet" else "unet"
        st2 = args.secondary_steps or (64 if wl2 == "classical" else 9)
        second = run_leg(wl2, threads_for(wl2, st2), st2, 8 if wl2 == "classical" else 3,
                         pcie_steps if pcie_steps is not None else (24 if wl2 == "classical" else 6))
    del st, st_flip
    unet_modes = None
    if rank == 0 and workload == "unet" and unet_mode != "miopen" and os.environ.get("TIP_BENCH_NO_MODES") != "1":
        unet_modes = unet_modes_record()

    def leg_object(lg):
        """Everything one leg measured, as a JSON object."""
        el = lg["elapsed"]
        wl = lg["workload"]
        o = {"workload": "%dx%dx%d_c%d_u16:%s" % (Y, X, Z, C, wl_names[wl]), "value": world * lg["steps"] / el, "unit": "frames/s",
             "steps": lg["steps"], "warmup": lg["warmup"], "ms_per_step": 1e3 * el / lg["steps"],
             "dtype": "f64" if wl != "unet" else plm.unet_arithmetic()["dtype"], "frames_in_flight_per_gpu": lg["nthreads"],
             "includes_h2d_upload": False, "kernels": kernel_table(lg["iso"], lg["iso_steps"])}
        if lg["pcie_elapsed"]:
            o["value_with_pcie"] = world * lg["pcie_steps"] / lg["pcie_elapsed"]
            o["with_pcie"] = {"value": o["value_with_pcie"], "unit": "frames/s", "steps": lg["pcie_steps"], "includes_h2d_upload": True,
                              "d2h_outputs": "projection (C,Y,X) float64 + label map int32 into pinned host buffers",
                              "h2d_bytes_per_frame": C * Z * Y * X * 2, "d2h_bytes_per_frame": C * Y * X * 8 + Y * X * 4}
        hb = roofline_of(lg["iso"], lg["iso_steps"])
        if hb:
            hb["measured"] = ("HIP events on the library stream, isolated pass of %d steps with ONE frame in flight run right after the "
                              "timed region (kernels of concurrent frames share the chip in the timed region)" % lg["iso_steps"])
        rt = roofline_of(merge(lg["timed"]), lg["steps"])
        if rt:
            rt["measured"] = ("HIP events on the library streams over a repeat of the timed region's %d steps with the same %d frames in "
                              "flight (%.1f frames/s with the events on; the timed region itself runs without them)"
                              % (lg["steps"], lg["nthreads"], world * lg["steps"] / lg["elapsed_events"]))
        if wl == "unet":
            o["roofline"] = unet_roofline(lg)
            o["roofline_library_kernel"] = hb
            o["watershed_diagnostics"] = lg["diag"]
        else:
            o["roofline"] = hb
        o["roofline_timed_region"] = rt
        # the heaviest ARITHMETIC kernels of the projection (the sigma-30 score passes) are bound by the FP32 matrix pipe, not
        # by HBM: the banded-Toeplitz MFMA tiles issue 2 * (32 + 2 * 120) flop per voxel and pass, of which the 2 * 241 of a
        # direct correlation are useful (the rest multiplies the band's zeros); `frac` is the USEFUL rate over the dense FP32
        # MFMA peak (= the FP32 vector peak, 256 flop / clk / CU)
        valu = {}
        cfg = (os.environ.get("TIP_FAST_CFG") or "5,5").split(",")
        for kname, which in (("score_fast_y", cfg[0]), ("score_fast_x", cfg[-1])):
            if kname in lg["iso"] and lg["iso"][kname][0]:
                cnt_k, ms_k = lg["iso"][kname]
                sec = ms_k / cnt_k / 1e3
                useful = Z * Y * X * 482.0 / sec / 1e12
                if which == "5":
                    # the fp16 tiles (csrc/tip_corr_f16.h): 17 K-steps of 16 positions x 3 piece products per output = 1632 flop issued
                    # for the 482 of a direct correlation; the pass moves 2 V 4 bytes -- with the matrix work sixteen times cheaper it
                    # is the memory system's turn, so both fractions are given
                    issued = Z * Y * X * 1632.0 / sec / 1e12
                    valu[kname] = {"bound": "mfma_f16 / hbm", "achieved": useful, "issued": issued, "peak": 2500.0, "unit": "TFLOP/s",
                                   "frac": issued / 2500.0, "hbm_GBps": 2 * Z * Y * X * 4 / sec / 1e9,
                                   "hbm_frac": 2 * Z * Y * X * 4 / sec / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": ms_k / cnt_k,
                                   "arithmetic": "float32 samples and taps split into two fp16 pieces, three products per term, float32 accumulation; "
                                                 "`frac` = issued fp16 MFMA flops / dense peak"}
                else:
                    valu[kname] = {"bound": "mfma_fp32", "achieved": useful, "issued": Z * Y * X * 544.0 / sec / 1e12,
                                   "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": useful / FP32_VALU_PEAK_TF, "avg_launch_ms": ms_k / cnt_k}
        o["roofline_valu"] = valu
        return o

    if rank == 0:
        head = leg_object(leg)
        out = {
            "metric": "frames/sec end-to-end (2048^2, z=30)", "value": head["value"], "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic",
            "config": {"workload": head["workload"], "value_is": workload,
                       "unet_arithmetic": plm.unet_arithmetic()["mode"] if workload == "unet" and unet_mode != "miopen" else None,
                       "parity": ("U2 (the trained network, pl.py:31-72 + model.load_weights) is UNPINNED: no TensorFlow and no trained weights in "
                                  "the container, the reference ships none -- random-init weights; the kernels are checked against the float64 "
                                  "evaluation of the same layers (1e-5), every other stage against reference-generated goldens, bit for bit")
                       if workload == "unet" else "every stage pinned by reference-generated goldens, bit for bit",
                       "baseline_config": ("configs[2]: 2048x2048 z=30 single frame, full pipeline (projection -> filter -> U-Net seg -> "
                                           "watershed/CCL)") if workload == "unet" else "the classical variant of configs[2] (no network)",
                       "frames_per_step": world, "frames_in_flight_per_gpu": leg["nthreads"], "includes_h2d_upload": False,
                       "network_passes_of_frames_in_flight": (None if workload != "unet" else
                                                              "concurrent (TISSUE_HIP_UNET_SERIAL=0)" if os.environ.get("TISSUE_HIP_UNET_SERIAL", "1") == "0"
                                                              else "one after the other on the device, other frames' projections / tails beside them"),
                       "parallelism": "frame-sharded dp%d, no data-path collective" % world},
            "value_with_pcie": head.get("value_with_pcie"), "with_pcie": head.get("with_pcie"),
            "roofline": head["roofline"], "roofline_timed_region": head["roofline_timed_region"], "roofline_valu": head["roofline_valu"],
            "kernels": head["kernels"],
        }
        for k in ("roofline_library_kernel", "watershed_diagnostics"):
            if k in head:
                out[k] = head[k]
        if second is not None:
            out[second["workload"]] = dict(leg_object(second), note="secondary leg, timed right after the headline leg in the same process")
        if cpu is not None:
            try:
                out["label_parity"] = label_parity_record(cpu, Y, X, Z, local_rank)
            except Exception as e:      # the parity record must not cost the run its line
                out["label_parity"] = {"error": repr(e)}
            for c in (cpu, cpu.get("classical")):
                if isinstance(c, dict):
                    c.pop("_outputs", None)
            out["cpu_baseline"] = cpu
        if unet_modes is not None:
            out["unet_arithmetic_modes"] = unet_modes
        print(json.dumps(out), flush=True)
        wd = leg["diag"] if workload == "unet" else (second["diag"] if second is not None and second["workload"] == "unet" else None)
        if wd and not all(v is None or v["all_two_valued"] for v in wd.values()):
            print("bench.py: a U-Net step's boundary image was not flooded by the two-valued mode: %r" % (wd,), file=sys.stderr)
            if use_dist:
                dist.destroy_process_group()
            return 3
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
