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


# ---- display / export array operations ------------------------------------------------------------
def _as_float_like_skimage(image):
    """skimage.util.img_as_float as skimage.filters.gaussian applies it (unsigned integers scaled by 1 / dtype max in
    float64, floats untouched)."""
    image = np.asarray(image)
    if image.dtype.kind == "u":
        return np.multiply(image, 1.0 / np.iinfo(image.dtype).max, dtype=np.float64)
    if image.dtype.kind == "f":
        return image
    raise TypeError("band_pass_filter on MI355X takes unsigned integer or float images (got %s)" % image.dtype)


def band_pass_filter(image, lowsigma, highsigma):
    """bim.py:393-414: skimage.filters.difference_of_gaussians = gaussian(low) - gaussian(high), edges replicated.
    Both blurs run on the device with scipy's exact tap arithmetic (blur_image)."""
    f = _as_float_like_skimage(image)
    if np.any(np.asarray(highsigma, dtype=float) < np.asarray(lowsigma, dtype=float)):
        raise ValueError("high_sigma must be equal to or larger thanlow_sigma for all axes")   # skimage's wording
    return blur_image(f, lowsigma) - blur_image(f, highsigma)


def _scoreatpercentile(values, per):
    """scipy.stats.scoreatpercentile (default 'fraction' interpolation): the two neighbouring order statistics come from
    the device (radix select, no sort), scipy's weighting from the host."""
    from . import _segmentation as seg
    flat = np.ascontiguousarray(values, dtype=np.float64).reshape(1, -1)
    idx = per / 100.0 * (flat.size - 1)
    i = int(idx)
    lo, hi = seg.label_order_stats(None, flat, 1, np.array([i], np.int64))
    if i == idx:
        return lo[0]
    w = np.array([(i + 1) - idx, idx - i], float)
    return np.add.reduce(np.array([lo[0], hi[0]]) * w) / w.sum()


def set_channel_brightness(image, max_possible_val, method='bestFit', clearExtreamPrecentage=1, minimum_pixel_val=0):
    """bim.py:299-348: saturate the extreme percentiles, shift / scale to [0, 1] (+ 1 / max_possible_val).  `image` is a
    float64 channel and is clipped in place like upstream; skimage's adjust_gamma with gamma 1 is the identity."""
    if clearExtreamPrecentage > 0:
        new_maximum = _scoreatpercentile(image, 100 - clearExtreamPrecentage)
        new_minimum = _scoreatpercentile(image, clearExtreamPrecentage)
        if minimum_pixel_val > 0:
            new_minimum = max(new_minimum, minimum_pixel_val)
        image[image > new_maximum] = new_maximum
    else:
        new_minimum = minimum_pixel_val
    if method in ('minMax', 'bestFit'):
        image = image - new_minimum
        image = image / np.max(image)
        image = image + 1 / max_possible_val
        image[image < 0] = 0
    return image


def set_brightness(image, axes, metadata={}, method='bestFit', clearExtreamPrecentage=1, minVal=0, maxVal=0):
    """bim.py:233-297: per-channel brightness adjustment to floats in [0, 1]; with metadata, (image, adjusted copy of it)."""
    from copy import deepcopy
    kind = np.asarray(image).dtype
    top = maxVal if maxVal else (255 if kind == np.uint8 else 65535 if kind == np.uint16 else 1)
    adjusted = np.array(image, dtype=np.float64)
    floor = metadata['min'] if (metadata and 'min' in metadata) else max(minVal, 0)
    if axes.find("C") >= 0:
        adjusted, order = put_channel_axis_first(adjusted, axes)
        for channel in range(adjusted.shape[0]):
            adjusted[channel] = set_channel_brightness(adjusted[channel], top, method, clearExtreamPrecentage, floor)
        adjusted = np.transpose(adjusted, axes=np.argsort(order))
    else:
        adjusted = set_channel_brightness(adjusted, top, method, clearExtreamPrecentage, floor)
    if not metadata:
        return adjusted
    meta = deepcopy(metadata)
    if 'min' in meta:
        meta['min'] = 0
    if 'max' in meta:
        meta['max'] = top
    if 'Ranges' in meta:
        meta['Ranges'] = (0, top) * int(len(meta['Ranges']) // 2)
    return adjusted, meta


def tiff_normalise(image, data_type=""):
    """The conversion save_tiff applies before writing (bim.py:183-186): images that are not already of the requested
    unsigned type are scaled so that their maximum becomes the type's maximum, and rounded."""
    image = np.asarray(image)
    if data_type and image.dtype != data_type and data_type in ('uint8', 'uint16'):
        top = 255 if data_type == 'uint8' else 65535
        image = np.round((image / np.max(image)) * top).astype(data_type)
    return image


def save_tiff(path, image, metadata=None, axes="", data_type=""):
    """bim.py:160-188.  Upstream hands the array to aicsimageio's OME-TIFF writer; here a self-contained baseline TIFF
    writer stores every (Y, X) plane of the normalised array as one page (little-endian, uncompressed, min-is-black) with
    the axes string and shape in the first page's ImageDescription -- what ImageJ / tifffile / the reference's read_tiff
    open as a stack.  OME-XML metadata is not written."""
    import struct
    image = tiff_normalise(image, data_type)
    if image.dtype == np.float64:
        image = image.astype(np.float32)
    if image.dtype not in (np.uint8, np.uint16, np.float32, np.int32):
        raise TypeError("save_tiff writes uint8 / uint16 / int32 / float32 pages (got %s)" % image.dtype)
    if image.ndim < 2:
        raise ValueError("save_tiff needs at least a 2-D image")
    planes = np.ascontiguousarray(image).reshape((-1,) + image.shape[-2:]).astype(image.dtype.newbyteorder("<"))
    rows, cols = planes.shape[1:]
    bits = planes.dtype.itemsize * 8
    fmt = 3 if planes.dtype.kind == "f" else (2 if planes.dtype.kind == "i" else 1)
    desc = ("axes=%s shape=%s" % (axes, "x".join(str(v) for v in image.shape))).encode("ascii") + b"\0"
    if planes.nbytes + planes.shape[0] * 256 + len(desc) >= 2 ** 32:
        raise ValueError("save_tiff: classic TIFF holds less than 4 GiB; split the movie")
    with open(path, "wb") as fh:
        fh.write(struct.pack("<2sHI", b"II", 42, 8))
        offset = 8
        for k in range(planes.shape[0]):
            entries = [(256, 4, 1, cols), (257, 4, 1, rows), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1),
                       (277, 3, 1, 1), (278, 4, 1, rows), (279, 4, 1, rows * cols * planes.dtype.itemsize), (339, 3, 1, fmt)]
            extra = desc if k == 0 else b""
            ifd_size = 2 + 12 * (len(entries) + 1 + (1 if extra else 0)) + 4
            data_at = offset + ifd_size + len(extra)
            entries.append((273, 4, 1, data_at))
            if extra:
                entries.append((270, 2, len(extra), offset + ifd_size))
            entries.sort()
            nxt = data_at + rows * cols * planes.dtype.itemsize if k + 1 < planes.shape[0] else 0
            fh.write(struct.pack("<H", len(entries)))
            for tag, typ, cnt, val in entries:
                fh.write(struct.pack("<HHII", tag, typ, cnt, val))
            fh.write(struct.pack("<I", nxt))
            fh.write(extra)
            fh.write(planes[k].tobytes())
            offset = nxt
    return
