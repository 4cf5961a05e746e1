"""phase_cross_correlation on MI355X (tip_phase_correlation): the drop-in for the skimage call behind
Tissue.update_drift / calculate_refine_drift (ti.py:1941-2035) and bim.calculate_drift (bim.py:522-536)."""
import ctypes

import numpy as np

from . import _lib


def phase_cross_correlation(reference_image, moving_image, upsample_factor=1, space="real", return_error=True):
    """Returns (shifts, error, phasediff) like skimage 0.18; error and phasediff are not computed (None): the reference
    discards them at every call site (ti.py:1976, 2029; bim.py:533-535)."""
    if space.lower() != "real":
        raise NotImplementedError("only space='real' (the reference's usage)")
    a = np.asarray(reference_image)
    b = np.asarray(moving_image)
    if a.shape != b.shape:
        raise ValueError("images must be same shape")
    if a.ndim != 2:
        raise NotImplementedError("2-D frames only")
    if a.dtype != b.dtype:
        a = a.astype(np.float64)
        b = b.astype(np.float64)
    if a.dtype == np.uint16:
        dt = 3
    elif a.dtype == np.float32:
        dt = 0
    else:
        a = a.astype(np.float64)
        b = b.astype(np.float64)
        dt = 1
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    ny, nx = a.shape
    for n in (ny, nx):
        if n < 2 or n > 4096:
            raise NotImplementedError("MI355X phase correlation takes extents in [2, 4096] (got %dx%d)" % (ny, nx))
    out = (ctypes.c_int64 * 4)()
    _lib.check(_lib.lib().tip_phase_correlation(_lib.ptr(a), _lib.ptr(b), dt, ny, nx, int(upsample_factor), out))
    return _finish_shifts(out, ny, nx, upsample_factor), None, None


def _finish_shifts(out, ny, nx, upsample_factor):
    """skimage's closing arithmetic on the two integer peaks the library returns (whole-pixel peak, upsampled-DFT peak)."""
    shape = np.array([ny, nx])
    shifts = np.array([out[0], out[1]], dtype=np.float64)
    midpoints = np.array([np.fix(s / 2) for s in shape])
    shifts[shifts > midpoints] -= shape[shifts > midpoints]
    if upsample_factor > 1:
        uf = float(upsample_factor)
        shifts = np.round(shifts * uf) / uf
        dftshift = np.fix(np.ceil(uf * 1.5) / 2.0)
        maxima = np.array([out[2], out[3]], dtype=np.float64) - dftshift
        shifts = shifts + maxima / uf
    return shifts


def phase_cross_correlation_dev(ref_ptr, mov_ptr, ny, nx, upsample_factor=100, dtype="float64"):
    """The same on two device-resident (ny, nx) planes (device addresses); returns the shift array."""
    dt = {"float32": 0, "float64": 1, "uint16": 3}[dtype]
    out = (ctypes.c_int64 * 4)()
    _lib.check(_lib.lib().tip_phase_correlation_dev(_lib.dptr(ref_ptr), _lib.dptr(mov_ptr), dt, int(ny), int(nx),
                                                    int(upsample_factor), out))
    return _finish_shifts(out, ny, nx, upsample_factor)
