"""ctypes binding of libtissue_hip.so (C-ABI declared in include/tissue_hip.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU is visible, every operator
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TISSUE_HIP_LIB: another build of the same library (diagnostic builds, tools/unet_trace.sh); never a different implementation
LIB_PATH = os.environ.get("TISSUE_HIP_LIB") or os.path.join(_HERE, "libtissue_hip.so")
_lib = None
_lock = threading.Lock()
_tls = threading.local()

c_int, c_long, c_double, c_void_p, c_size_t = ctypes.c_int, ctypes.c_long, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t
c_i64 = ctypes.c_int64

TIP_ERR_ARG, TIP_ERR_INDEX, TIP_ERR_OVERFLOW = -2, -4, -6


class TissueHipError(RuntimeError):
    pass


def load():
    """Loads the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise TissueHipError(
                    "libtissue_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `python -m tissue_image_processing_amd.build`. There is no CPU fallback." % LIB_PATH)
            _preload_torch_hip_runtime()
            _lib = ctypes.CDLL(LIB_PATH)
            _lib.tip_prof_report.restype = c_int
    return _lib


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so with the same soname as /opt/rocm's, so a process gets
    whichever copy is loaded first.  When this library came first and torch was imported afterwards (the U-Net path),
    torch saw no device.  If torch is installed its copy is therefore loaded first, without importing torch -- the same
    state as in a process that imported torch before this package (bench.py, the drivers)."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def device_for_thread():
    return getattr(_tls, "device", None)


def init(device=None):
    """Binds the calling thread to a GPU (default: env TISSUE_HIP_DEVICE, LOCAL_RANK, else 0)."""
    lib = load()
    if device is None:
        device = int(os.environ.get("TISSUE_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if lib.tip_device_count() <= 0:
        raise TissueHipError("no HIP device visible: the tissue_image_processing_amd operators need an MI355X "
                             "(there is no CPU fallback)")
    check(lib.tip_init(int(device)))
    _tls.device = int(device)
    return lib


def lib():
    l = load()
    if device_for_thread() is None:
        init()
    return l


def last_error():
    buf = ctypes.create_string_buffer(1024)
    load().tip_last_error(buf, c_size_t(1024))
    return buf.value.decode("utf-8", "replace")


def check(rc):
    if rc == 0:
        return
    msg = last_error()
    if rc == TIP_ERR_ARG:
        raise ValueError(msg)
    if rc == TIP_ERR_INDEX:
        raise IndexError(msg)
    raise TissueHipError("tissue_hip error %d: %s" % (rc, msg))


def ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def dptr(addr):
    return c_void_p(int(addr))


class DeviceBuffer:
    """Owned device allocation (tip_malloc/tip_free)."""

    def __init__(self, nbytes):
        p = c_void_p()
        check(lib().tip_malloc(ctypes.byref(p), c_size_t(int(nbytes))))
        self.ptr = p.value
        self.nbytes = int(nbytes)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib().tip_memcpy_h2d(dptr(self.ptr), ptr(arr), c_size_t(arr.nbytes)))
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib().tip_memcpy_d2h(ptr(out), dptr(self.ptr), c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.ptr:
            load().tip_free(dptr(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


WS_FLAG_TIES, WS_FLAG_TWO_VALUED, WS_FLAG_SERIAL_EXACT, WS_FLAG_SERIAL_FINISH, WS_FLAG_COUNT_SHIFT = 1, 2, 4, 8, 8


def set_tuning(name, value=None):
    """tip_set_tuning: one of the TIP_* hooks of include/tissue_hip.h; value None restores the default.  The library reads
    the environment once, when it is first used -- later changes of os.environ do not reach it, this call does."""
    l = load()
    rc = l.tip_set_tuning(name.encode(), None if value is None else str(value).encode())
    if rc != 0:
        raise ValueError("unknown tuning name %r" % (name,))


class tuning:
    """Context manager: with _lib.tuning(TIP_WS_TIES="fast"): ..."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            set_tuning(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            set_tuning(k, os.environ.get(k))
        return False


def prof_enable(on=True):
    check(lib().tip_prof_enable(1 if on else 0))


def prof_reset():
    check(lib().tip_prof_reset())


def prof_report():
    """{kernel_name: (count, total_ms)} measured with HIP events on the library's stream."""
    n = lib().tip_prof_report(None, c_size_t(0))
    buf = ctypes.create_string_buffer(n + 16)
    lib().tip_prof_report(buf, c_size_t(n + 16))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.rsplit(" ", 2)
        out[name] = (int(cnt), float(ms))
    return out
