"""CPU: `python bench.py --gpus N` without a launcher environment starts N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* as torchrun would set them) and rank 0 prints the one JSON line with n_gpus = N.  TIP_BENCH_STUB=1 swaps the GPU
work for a sleep and RCCL for gloo, so the launcher, the process group, the barriers and the max-over-ranks reduction run
here; without the stub the launcher must refuse when fewer than N devices are visible."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, **env):
    e = dict(os.environ, **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_starts_two_ranks():
    r = run(["--gpus", "2", "--steps", "5", "--warmup", "1"], TIP_BENCH_STUB="1")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                              # ONE line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["steps"] == 5 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["data"].startswith("stub")


def test_gpus_n_fails_loudly_without_the_devices():
    import torch
    have = torch.cuda.device_count()
    r = run(["--gpus", str(have + 2), "--steps", "1", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert "device" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
