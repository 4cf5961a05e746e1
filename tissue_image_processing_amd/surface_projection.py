"""Drop-in for time_point_surface_projection of the reference's surface_projection.py (sp.py:17-85) on MI355X.

Signature, defaults, return types and error behaviour follow the reference; the arithmetic runs in
libtissue_hip.so (tip_project_u16 / tip_project_u16_binned).  Covered: bin_size == 1 (what every BASELINE config and
movie_surface_projection's default use) and bin_size > 1 with methods 'max_averages', 'max_std', 'multi_channel'
(sp.py:39-65); build_manifold (the serial spiral of sp.py:87-165) is not.
"""

_METHODS = {"max_averages": 0, "max_std": 1, "multi_channel": 2}
import ctypes

import numpy as np

from . import _lib
from .basic_image_manipulations import put_channel_axis_first, gaussian_taps


def time_point_surface_projection(time_point, axes, reference_channel, min_z=0, max_z=0,
                                  method='max_averages', bin_size=1, airyscan=True, z_map=False, atoh_shift=0,
                                  build_manifold=False):
    if bin_size > 1 and method not in ("max_averages", "max_std", "multi_channel"):
        raise TypeError("exceptions must derive from BaseException")  # sp.py:53 raises a str
    if build_manifold:
        raise NotImplementedError("MI355X path covers build_manifold=False (SURVEY.md 8f rank 3)")
    if bin_size > 128:
        raise NotImplementedError("MI355X path covers bin_size <= 128")
    if axes.find("T") >= 0:
        time_point = time_point.reshape(time_point.shape[1:])
        image, _ = put_channel_axis_first(time_point, axes[1:])
    else:
        image, _ = put_channel_axis_first(time_point, axes)
    if axes.find("C") < 0 or image.ndim != 4:
        # sp.py:32 indexes image[reference_channel] on a (Z,Y,X) array and then blurs the 2-D slice with a
        # 3-tuple sigma -> scipy's RuntimeError (golden: tests/golden/projection.npz c_error)
        raise RuntimeError("sequence argument must have length equal to input rank")
    image = np.asarray(image)
    if image.dtype != np.uint16:
        if np.issubdtype(image.dtype, np.integer) and image.size and image.min() >= 0 and image.max() <= 65535:
            image = image.astype(np.uint16)
        else:
            raise TypeError("MI355X projection takes uint16 stacks (microscope data); got %s" % image.dtype)
    image = np.ascontiguousarray(image)
    C, Z, Y, X = image.shape
    if not (-C <= reference_channel < C):
        raise IndexError("index %d is out of bounds for axis 0 with size %d" % (reference_channel, C))
    reference_channel %= C
    zlo, zhi = (min_z, min(max_z, Z)) if max_z > 0 else (0, Z)
    if zlo < 0:
        zlo = max(0, Z + zlo)
    if zhi <= zlo:
        raise ValueError("attempt to get argmax of an empty sequence")  # numpy's error for an empty z slice
    t05, t1, t2, t30 = gaussian_taps(0.5), gaussian_taps(1.0), gaussian_taps(2.0), gaussian_taps(30.0)
    proj = np.empty((C, Y, X), np.float64)
    zmap = np.empty((Y, X), np.int64)
    lib = _lib.lib()
    if bin_size > 1:
        rc = lib.tip_project_u16_binned(_lib.ptr(image), C, Z, Y, X, int(zlo), int(zhi), int(min_z), int(reference_channel),
                                        _METHODS[method], int(bin_size), 1 if airyscan else 0, int(atoh_shift),
                                        _lib.ptr(t05), _lib.ptr(t1), _lib.ptr(t2), _lib.ptr(t30), _lib.ptr(proj),
                                        _lib.ptr(zmap))
    else:
        rc = lib.tip_project_u16(_lib.ptr(image), C, Z, Y, X, int(zlo), int(zhi), int(min_z), int(reference_channel),
                                 1 if airyscan else 0, int(atoh_shift), _lib.ptr(t05), _lib.ptr(t1), _lib.ptr(t2),
                                 _lib.ptr(t30), _lib.ptr(proj), _lib.ptr(zmap))
    _lib.check(rc)
    if z_map:
        return proj, zmap
    return proj
