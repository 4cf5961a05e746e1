"""Builds libtissue_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

-ffp-contract=off is part of the numerical contract: scipy's x86-64 wheels round the multiply and the add of
`tmp += (a+b)*w` separately, and bit parity with them needs the same on the device (and in host-side tap code).

Every csrc/*.hip is compiled to its own object (in parallel, only when it or a header changed) and the objects are linked;
the objects live under csrc/_obj/ (git-ignored).
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libtissue_hip.so")
CFLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-fvisibility=hidden",
          "-std=c++17", "-Wno-unused-result"]
LDFLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared", "-fvisibility=hidden"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "tissue_hip.h")]


def _obj_of(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    return _stale(LIB, sources() + _headers() + [os.path.abspath(__file__)])


def _hipcc():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def build(force=False, verbose=False, extra_flags=(), out=None):
    """extra_flags / out: diagnostic builds (tools/unet_trace.sh: -DUC_TRACE into a library of another name)."""
    out = out or LIB
    diag = bool(extra_flags) or out != LIB
    if not force and not diag and not needs_build():
        return LIB
    hipcc = _hipcc()
    objdir = OBJ if not diag else OBJ + "_" + os.path.basename(out)
    os.makedirs(objdir, exist_ok=True)
    hdrs = _headers() + [os.path.abspath(__file__)]
    jobs = []
    for s in sources():
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        if force or diag or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + CFLAGS + list(extra_flags) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in sources()]
    run([hipcc] + LDFLAGS + objs + ["-o", out])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
