"""Drop-in for the array operators of the reference's basic_image_manipulations.py (bim.py), running on MI355X.

Same function names, argument meaning, return dtypes and error behaviour as the reference:
    put_channel_axis_first(image, axes)                              bim.py:199-231   (pure view logic, host)
    blur_image(image, std)                                           bim.py:373-390   -> tip_gaussian3d_w
    watershed_segmentation(image, imgthresh, stdeviation, blocksize) bim.py:446-476   -> tip_watershed_segmentation
File I/O (read_tiff, save_tiff, ...) is out of scope.
"""
import ctypes

import numpy as np

from . import _lib

UINT8_MAXVAL = 255
UINT16_MAXVAL = 65535


def _gaussian_kernel1d(sigma, radius):
    """Taps exactly as scipy's Python layer builds them (scipy/ndimage/filters.py:_gaussian_kernel1d, order 0):
    host numpy, like scipy itself, so that a process that also imports scipy sees identical taps."""
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    return phi_x


def gaussian_taps(sigma, truncate=4.0):
    sd = float(sigma)
    lw = int(truncate * sd + 0.5)
    return np.ascontiguousarray(_gaussian_kernel1d(sd, lw)[::-1], dtype=np.float64)


def put_channel_axis_first(image, axes):
    """bim.py:199-231: (array, order) with the channel axis moved to the front and the remaining axes in the
    reference's canonical order C, [T], [Z], X, Y.  The reference only reorders when "C" is present and not already
    first; otherwise the array comes back untouched with the identity order."""
    where = {name: axes.find(name) for name in "CTZXY"}
    if where["C"] <= 0:
        return image, tuple(np.arange(len(axes)))
    order = tuple(where[name] for name in "CTZXY" if name in "CXY" or where[name] >= 0)
    return np.transpose(image, axes=order), order


def _normalize_sigma(std, ndim):
    sig = np.ravel(np.asarray(std, dtype=np.float64))
    if sig.size == 1:
        sig = np.repeat(sig, ndim)
    if sig.size != ndim:
        # scipy's _ni_support._normalize_sequence
        raise RuntimeError("sequence argument must have length equal to input rank")
    return sig


def blur_image(image, std):
    """bim.py:373-390: scipy.ndimage.gaussian_filter(image, std, mode='nearest'); same shape and dtype out.

    float32 / float64 arrays of rank 1..3 run on the GPU with scipy's exact arithmetic (double accumulation in
    scipy's tap order, rounding to the array dtype after each axis).  Integer images follow scipy's rule
    "output dtype == input dtype": they are filtered in float64 and truncated on store, as scipy's C core does.
    """
    image = np.asarray(image)
    if image.ndim < 1 or image.ndim > 3:
        raise ValueError("blur_image on MI355X supports rank 1..3 arrays (got rank %d)" % image.ndim)
    sig = _normalize_sigma(std, image.ndim)
    if np.issubdtype(image.dtype, np.integer) or image.dtype == bool:
        # scipy keeps the input dtype: every axis pass accumulates in double and the C core casts the result back to
        # the integer type (truncation toward zero) before the next axis sees it
        cur = image.astype(np.float64)
        for ax in range(image.ndim):
            if sig[ax] > 1e-15:
                one = np.zeros(image.ndim)
                one[ax] = sig[ax]
                cur = np.trunc(blur_image(cur, tuple(one)))
        return cur.astype(image.dtype)
    src = image
    if image.dtype == np.float32:
        dtype = 0
    elif image.dtype == np.float64:
        dtype = 1
    else:
        raise TypeError("blur_image on MI355X supports float32/float64/integer images (got %s)" % image.dtype)
    src = np.ascontiguousarray(src)
    if src.size == 0:
        return src.copy()
    shape3 = (1,) * (3 - src.ndim) + src.shape
    taps = [None, None, None]
    for ax in range(src.ndim):
        if sig[ax] > 1e-15:
            taps[ax + 3 - src.ndim] = gaussian_taps(sig[ax])
    for t in taps:
        if t is not None and t.size > 255:
            raise ValueError("blur_image on MI355X supports sigma <= 31.8 (radius <= 127)")
    out = np.empty_like(src)
    lib = _lib.lib()
    args = []
    for t in taps:
        args += [_lib.ptr(t), ctypes.c_int(0 if t is None else t.size)]
    _lib.check(lib.tip_gaussian3d_w(_lib.ptr(src), _lib.ptr(out), dtype, shape3[0], shape3[1], shape3[2], *args))
    return out


def watershed_segmentation(image, imgthresh, stdeviation, blocksize):
    """bim.py:446-476 (the 4-argument definition, which shadows the 3-argument one at bim.py:417-443).

    threshold_local(generic max) -> zero below threshold -> blur_image(stdeviation) -> skimage watershed with
    watershed_line=True.  Returns int32 labels, 0 on watershed lines.
    """
    from . import _segmentation
    return _segmentation.watershed_segmentation(image, imgthresh, stdeviation, blocksize)


def calculate_drift(first_image, second_image, sub_pixel_precision=True):
    """bim.py:522-536: global 2-D drift between two frames by phase cross-correlation (shift[-2:])."""
    from ._registration import phase_cross_correlation
    if sub_pixel_precision:
        shift, error, diffphase = phase_cross_correlation(first_image, second_image, upsample_factor=100)
    else:
        shift, error, diffphase = phase_cross_correlation(first_image, second_image)
    return shift[-2:]
