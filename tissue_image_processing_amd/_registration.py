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


# ---- local drifts (ti.py:2149-2175): one refined drift per window of a frame pair ----------------------------------------
def local_drift_windows(shape, step_size=100, window_size=700):
    """The (row0, row1, col0, col1) windows upstream slides over a frame: starts every step_size pixels while
    start < extent - window_size; a window that could not be followed by another whole one runs to the frame's edge."""
    H, W = shape
    out = []
    for r0 in range(0, H - window_size, step_size):
        r1 = H if r0 + step_size + window_size > H else r0 + window_size
        for c0 in range(0, W - window_size, step_size):
            c1 = W if c0 + step_size + window_size > W else c0 + window_size
            out.append((r0, r1, c0, c1))
    return out


def _overlap(n, shift):
    """Offsets (into the previous window, into the current window) and length of the overlap calculate_refine_drift crops
    along one axis for a floored coarse shift (ti.py:1945-1973)."""
    if shift > 0:
        return shift, 0, n - shift
    if shift < 0:
        return 0, -shift, n + shift
    return 0, 0, n


def local_drifts(first_image, second_image, initial_shift_x=0, initial_shift_y=0, step_size=100, window_size=700):
    """[(window, shift_x, shift_y)] in upstream's loop order: Tissue.calculate_refine_drift on every window of the pair
    (ti.py:2152-2166).  Both frames are uploaded once; every window is cut on the device (tip_memcpy2d_d2d) and goes
    through the device phase correlation (upsample factor 100)."""
    a = np.asarray(first_image)
    b = np.asarray(second_image)
    if a.shape != b.shape or a.ndim != 2:
        raise ValueError("local_drifts takes two 2-D frames of one shape")
    if a.dtype != b.dtype or a.dtype not in (np.uint16, np.float32, np.float64):
        a = a.astype(np.float64)
        b = b.astype(np.float64)
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    name = {np.dtype(np.uint16): "uint16", np.dtype(np.float32): "float32", np.dtype(np.float64): "float64"}[a.dtype]
    es = a.dtype.itemsize
    H, W = a.shape
    windows = local_drift_windows((H, W), step_size, window_size)
    if not windows:
        return []
    rx, ry = int(np.floor(initial_shift_x)), int(np.floor(initial_shift_y))
    lib = _lib.lib()
    da, db = _lib.DeviceBuffer(a.nbytes).upload(a), _lib.DeviceBuffer(b.nbytes).upload(b)
    big = max((r1 - r0) * (c1 - c0) for r0, r1, c0, c1 in windows) * es
    wa, wb = _lib.DeviceBuffer(big), _lib.DeviceBuffer(big)
    out = []
    try:
        for (r0, r1, c0, c1) in windows:
            pr, cr, ny = _overlap(r1 - r0, rx)
            pc, cc, nx = _overlap(c1 - c0, ry)
            if ny < 2 or nx < 2:
                raise NotImplementedError("MI355X phase correlation takes extents in [2, 4096] (got %dx%d)" % (ny, nx))
            for dst, src, ro, co in ((wa, da, r0 + pr, c0 + pc), (wb, db, r0 + cr, c0 + cc)):
                _lib.check(lib.tip_memcpy2d_d2d(_lib.dptr(dst.ptr), ctypes.c_size_t(nx * es), _lib.dptr(src.ptr + (ro * W + co) * es),
                                                ctypes.c_size_t(W * es), ctypes.c_size_t(nx * es), ctypes.c_size_t(ny)))
            sh = phase_cross_correlation_dev(wa.ptr, wb.ptr, ny, nx, 100, dtype=name)
            out.append(((r0, r1, c0, c1), rx + sh[-2], ry + sh[-1]))
    finally:
        for buf in (da, db, wa, wb):
            buf.free()
    return out


def sample_local_drift(drifts, rows, cols):
    """local_shifts_x / local_shifts_y of ti.py:2149-2168 at the pixels (rows, cols): the mean of the shifts of the
    windows that contain the pixel, added up in upstream's loop order (NaN where no window does: 0 / 0 upstream)."""
    rows = np.asarray(rows)
    cols = np.asarray(cols)
    sx = np.zeros(rows.shape)
    sy = np.zeros(rows.shape)
    cnt = np.zeros(rows.shape)
    for (r0, r1, c0, c1), dx, dy in drifts:
        inside = (rows >= r0) & (rows < r1) & (cols >= c0) & (cols < c1)
        sx[inside] += dx
        sy[inside] += dy
        cnt[inside] += 1
    with np.errstate(invalid="ignore", divide="ignore"):
        return sx / cnt, sy / cnt
