"""Drop-in for the array operators of the reference's basic_image_manipulations.py (bim.py), running on MI355X.

Same function names, argument meaning, return dtypes and error behaviour as the reference:
    put_channel_axis_first(image, axes)                              bim.py:199-231   (pure view logic, host)
    blur_image(image, std)                                           bim.py:373-390   -> tip_gaussian3d_w
    watershed_segmentation(image, imgthresh, stdeviation, blocksize) bim.py:446-476   -> tip_watershed_segmentation
    read_image_in_chunks / get_image_dimensions / concatenate_time_points / read_tiff / save_tiff
                                                                     bim.py:28-188, 478-495 (host I/O around the GPU path;
                                                                     image sources: arrays, .npy, TIFF stacks, aicsimageio objects)
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

    float32 / float64 arrays of any rank run on the GPU with scipy's exact arithmetic (double accumulation in
    scipy's tap order, rounding to the array dtype after each axis).  Integer images follow scipy's rule
    "output dtype == input dtype": they are filtered in float64 and truncated on store, as scipy's C core does.
    """
    image = np.asarray(image)
    if image.ndim < 1:
        raise ValueError("blur_image needs an array of rank >= 1")
    sig = _normalize_sigma(std, image.ndim)
    if image.ndim > 3:
        # scipy filters axis after axis, rounding to the array dtype in between: for any rank that is one pass per axis over
        # the (leading, axis, trailing) view of the array, each of them a rank-3 call below
        cur = np.ascontiguousarray(image)
        for ax in range(image.ndim):
            if sig[ax] > 1e-15:
                lead = int(np.prod(image.shape[:ax], dtype=np.int64))
                trail = int(np.prod(image.shape[ax + 1:], dtype=np.int64))
                cur = blur_image(cur.reshape(lead, image.shape[ax], trail), (0.0, sig[ax], 0.0)).reshape(image.shape)
        return cur if cur is not image else image.copy()
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


def stretch_for_display(disp_img, min_percent, max_percent):
    """The display stretch of gui.py:445-452 / 468-475 (display_frame: the zo and atoh planes of the composite): both level
    percentiles of the plane (np.percentile, linear), a maximum equal to the minimum is moved up by one, then
    255 * max(plane - lo, 0) / (hi - lo) clipped at 255, in float64.  The two order statistics -- the only dense-array work --
    come from the device radix select (tip_label_order_stats_f64); the arithmetic on them is numpy's."""
    from . import _segmentation as seg
    img = np.asarray(disp_img)
    lo = seg.percentile_frame(img, float(min_percent))
    hi = seg.percentile_frame(img, float(max_percent))
    if hi == lo:
        hi += 1
    out = img - lo
    np.putmask(out, out < 0, 0)
    out = 255 * out / (hi - lo)
    np.putmask(out, out > 255, 255)
    return out


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
    open as a stack.  `metadata` with a to_xml() method (or a string) is appended to that description; other objects are
    reported once as not written."""
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
    if metadata is not None:
        # upstream hands the OME metadata to aicsimageio's writer; here its XML (if it has one) rides in the description
        xml = metadata.to_xml() if hasattr(metadata, "to_xml") else (metadata if isinstance(metadata, (str, bytes)) else None)
        if xml is None:
            import warnings
            warnings.warn("save_tiff: metadata of type %s is not written (no OME-XML writer here)" % type(metadata).__name__)
        else:
            desc = desc[:-1] + b"\n" + (xml if isinstance(xml, bytes) else xml.encode("utf-8", "replace")) + b"\0"
    desc += b"\0" * (len(desc) & 1)                       # TIFF 6.0: every IFD starts on a word boundary
    plane_bytes = rows * cols * planes.dtype.itemsize
    pad = plane_bytes & 1                                   # (odd-sized uint8 planes)
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
            nxt = data_at + plane_bytes + pad if k + 1 < planes.shape[0] else 0
            fh.write(struct.pack("<H", len(entries)))
            for tag, typ, cnt, val in entries:
                fh.write(struct.pack("<HHII", tag, typ, cnt, val))
            fh.write(struct.pack("<I", nxt))
            fh.write(extra)
            fh.write(planes[k].tobytes())
            fh.write(b"\0" * pad)
            offset = nxt
    return


# ---- image sources and the chunk iterator (bim.py:28-159, 478-495) -------------------------------------------------------
# The reference opens everything through aicsimageio (absent in this image).  Here an image source is anything that can
# say its (T, C, Z, Y, X) extents and hand out 5-D sub-blocks:
#   * a numpy array / memmap of rank 5 (TCZYX), or a list / tuple of them (one per scene = microscope position);
#   * a path to a .npy file (memory-mapped) or to an uncompressed TIFF stack (the reader below; axes from save_tiff's
#     or ImageJ's ImageDescription);
#   * an object with aicsimageio's interface (set_scene, dims, get_image_dask_data) -- a real AICSImage where that
#     package is installed; any other path is handed to aicsimageio and fails loudly without it.
class ImageDims(tuple):
    """(T, C, Z, Y, X) with aicsimageio-style attribute access (dims.T, dims.C, ...)."""
    __slots__ = ()
    _names = "TCZYX"

    def __new__(cls, t, c, z, y, x):
        return tuple.__new__(cls, (int(t), int(c), int(z), int(y), int(x)))

    def __getattr__(self, name):
        i = ImageDims._names.find(name)
        if i < 0 or len(name) != 1:
            raise AttributeError(name)
        return self[i]


class _ArraySource(object):
    def __init__(self, scenes):
        self.scenes = scenes
        self.scene = 0

    def set_scene(self, series):
        if not (0 <= series < len(self.scenes)):
            raise IndexError("scene %d of %d" % (series, len(self.scenes)))
        self.scene = series

    @property
    def dims(self):
        return ImageDims(*self.scenes[self.scene].shape)

    def block(self, t, c, z, y, x):
        return np.asarray(self.scenes[self.scene][t, c, z, y, x])


class _AicsSource(object):
    def __init__(self, img):
        self.img = img
        self.data = None

    def set_scene(self, series):
        self.img.set_scene(series)
        self.data = None

    @property
    def dims(self):
        d = self.img.dims
        return ImageDims(d.T, d.C, d.Z, d.Y, d.X)

    def block(self, t, c, z, y, x):
        if self.data is None:
            self.data = self.img.get_image_dask_data()
        chunk = self.data[t, c, z, y, x]
        return np.asarray(chunk.compute() if hasattr(chunk, "compute") else chunk)


def _as_tczyx(arr, axes):
    """View of `arr` (axes named by `axes`, a subset of TCZYX in any order) as a 5-D TCZYX array."""
    arr = np.asarray(arr) if not isinstance(arr, np.memmap) else arr
    axes = axes.upper()
    if len(axes) != arr.ndim or any(a not in "TCZYX" for a in axes) or len(set(axes)) != len(axes):
        raise ValueError("axes %r do not describe an array of rank %d" % (axes, arr.ndim))
    order = [axes.index(a) for a in "TCZYX" if a in axes]
    view = np.transpose(arr, order)
    shape, k = [], 0
    for a in "TCZYX":
        if a in axes:
            shape.append(view.shape[k])
            k += 1
        else:
            shape.append(1)
    return view.reshape(shape)


def open_image(source, series=0):
    """The image-source object for `source` (see above), positioned on scene `series`."""
    import os
    if hasattr(source, "block") and hasattr(source, "dims"):
        src = source
    elif hasattr(source, "get_image_dask_data"):
        src = _AicsSource(source)
    elif isinstance(source, (list, tuple)):
        src = _ArraySource([_as_tczyx(a, "TCZYX"[5 - np.ndim(a):]) for a in source])
    elif isinstance(source, np.ndarray):
        src = _ArraySource([_as_tczyx(source, "TCZYX"[5 - source.ndim:])])
    elif isinstance(source, (str, bytes, os.PathLike)):
        path = os.fspath(source)
        ext = os.path.splitext(path)[1].lower()
        if ext == ".npy":
            arr = np.load(path, mmap_mode="r")
            src = _ArraySource([_as_tczyx(arr, "TCZYX"[5 - arr.ndim:])])
        elif ext in (".tif", ".tiff"):
            image, axes, _, _ = read_tiff(path)
            src = _ArraySource([_as_tczyx(image, axes if axes else "TCZYX"[5 - image.ndim:])])
        else:
            try:
                from aicsimageio import AICSImage
                from aicsimageio.readers import bioformats_reader
            except ImportError as e:
                raise _lib.TissueHipError("reading %r needs aicsimageio (with bioformats), which is not installed: %s; "
                                          "convert the movie to .npy / TIFF or pass an array" % (path, e))
            src = _AicsSource(AICSImage(path, reader=bioformats_reader.BioformatsReader))
    else:
        raise TypeError("cannot open %r as an image" % (type(source),))
    src.set_scene(series)
    return src


def get_image_dimensions(path, series=0):
    """bim.py:80-83."""
    return open_image(path, series).dims


def read_image_in_chunks(path, series=0, dx=0, dy=0, dz=0, dc=0, dt=0, apply_function=None, output=None,
                         **apply_function_params):
    """bim.py:89-159: generator over the (dt, dc, dz, dy, dx) blocks of scene `series`, x fastest, then y, z, c, t.
    Without `apply_function` the blocks themselves are yielded; with it, its result per block -- and, when `output`
    is given, each result is also reshaped into the block's place in `output` (a 5-D TCZYX array, or a list of them when
    the function returns a tuple; an output axis shorter than the image is clipped, which is how the projection's
    z axis of length 1 takes a whole z range).  `path` is any image source of open_image()."""
    src = open_image(path, series)
    ext = src.dims                                    # (T, C, Z, Y, X)
    step = [d if d else e for d, e in zip((dt, dc, dz, dy, dx), ext)]
    starts = [range(0, e, s) for e, s in zip(ext, step)]
    for t in starts[0]:
        for c in starts[1]:
            for z in starts[2]:
                for y in starts[3]:
                    for x in starts[4]:
                        lo = (t, c, z, y, x)
                        hi = tuple(min(a + s, e) for a, s, e in zip(lo, step, ext))
                        chunk = src.block(*[slice(a, b) for a, b in zip(lo, hi)])
                        if apply_function is None:
                            yield chunk
                            continue
                        result = apply_function(chunk, **apply_function_params)
                        if output is None:
                            continue                  # (bim.py:121-146 yields nothing in this case)
                        single = not isinstance(result, tuple)
                        results = [result] if single else list(result)
                        outputs = [output] if single else output
                        for res, out in zip(results, outputs):
                            a = [min(p, n) for p, n in zip(lo, out.shape)]
                            b = [min(q, n) for q, n in zip(hi, out.shape)]
                            box = tuple(slice(p, q) for p, q in zip(a, b))
                            out[box] = np.reshape(res, [q - p for p, q in zip(a, b)])
                        yield result


def _tiff_axes_from_description(desc, npages, rows, cols):
    """(axes, shape) from the first page's ImageDescription: save_tiff's "axes=... shape=...", ImageJ's hyperstack
    keys, else a plain page stack."""
    import re
    m = re.search(r"axes=([A-Za-z]*) shape=([0-9x]+)", desc)
    if m and m.group(2):
        shape = tuple(int(v) for v in m.group(2).split("x"))
        if int(np.prod(shape)) == npages * rows * cols:
            return m.group(1).upper(), shape
    if desc.startswith("ImageJ="):
        keys = dict(line.split("=", 1) for line in desc.splitlines() if "=" in line)
        t, z, c = int(keys.get("frames", 1)), int(keys.get("slices", 1)), int(keys.get("channels", 1))
        if t * z * c == npages:
            axes, shape = "", ()
            for name, n in (("T", t), ("Z", z), ("C", c)):
                if n > 1:
                    axes += name
                    shape += (n,)
            return axes + "YX", shape + (rows, cols)
    if npages > 1:
        return "QYX", (npages, rows, cols)           # (tifffile's name for an unknown page axis)
    return "YX", (rows, cols)


def read_tiff(path):
    """bim.py:28-51: (image, axes, shape, metadata) of a TIFF file.  Self-contained reader for what this package and
    ImageJ write -- classic TIFF and BigTIFF, either byte order, uncompressed strips, one sample per pixel, 8 / 16 / 32-bit
    unsigned, signed or float pages of one size; anything else raises."""
    import struct
    with open(path, "rb") as fh:
        buf = fh.read()
    if len(buf) < 8 or buf[:2] not in (b"II", b"MM"):
        raise ValueError("%s is not a TIFF file" % path)
    bo = "<" if buf[:2] == b"II" else ">"
    magic, = struct.unpack(bo + "H", buf[2:4])
    if magic == 42:                       # classic: 4-byte offsets, 12-byte entries, 2-byte entry count
        big, ifd = False, struct.unpack(bo + "I", buf[4:8])[0]
    elif magic == 43:                     # BigTIFF: 8-byte offsets, 20-byte entries, 8-byte entry count
        big, ifd = True, struct.unpack(bo + "Q", buf[8:16])[0]
    else:
        raise ValueError("%s: not a TIFF / BigTIFF file (magic %d)" % (path, magic))
    sizes = {1: 1, 2: 1, 3: 2, 4: 4, 6: 1, 8: 2, 9: 4, 16: 8, 17: 8}
    codes = {1: "B", 3: "H", 4: "I", 6: "b", 8: "h", 9: "i", 16: "Q", 17: "q"}
    cnt_fmt, ent_size, inline, off_fmt = ("Q", 20, 8, "Q") if big else ("H", 12, 4, "I")
    pages, desc = [], ""
    while ifd:
        hdr = struct.calcsize(cnt_fmt)
        n, = struct.unpack(bo + cnt_fmt, buf[ifd:ifd + hdr])
        tags = {}
        for k in range(n):
            e = buf[ifd + hdr + ent_size * k: ifd + hdr + ent_size * (k + 1)]
            tag, typ = struct.unpack(bo + "HH", e[:4])
            cnt, = struct.unpack(bo + off_fmt, e[4:4 + inline])
            raw = e[4 + inline:]
            if typ not in sizes:
                continue
            nbytes = sizes[typ] * cnt
            data = raw[:nbytes] if nbytes <= inline else buf[struct.unpack(bo + off_fmt, raw)[0]:][:nbytes]
            tags[tag] = data if typ == 2 else struct.unpack(bo + "%d%s" % (cnt, codes[typ]), data)
        ifd, = struct.unpack(bo + off_fmt, buf[ifd + hdr + ent_size * n: ifd + hdr + ent_size * n + inline])
        if tags.get(259, (1,))[0] != 1 or tags.get(277, (1,))[0] != 1:
            raise ValueError("%s: compressed or multi-sample pages are not read here" % path)
        cols, rows, bits = tags[256][0], tags[257][0], tags.get(258, (1,))[0]
        kind = {1: "u", 2: "i", 3: "f"}[tags.get(339, (1,))[0]]
        dtype = np.dtype("%s%s%d" % (bo, kind, bits // 8))
        offs, counts = tags[273], tags.get(279, (rows * cols * dtype.itemsize,))
        raw = b"".join(buf[o:o + c] for o, c in zip(offs, counts))
        pages.append(np.frombuffer(raw, dtype=dtype, count=rows * cols).reshape(rows, cols))
        if not desc and 270 in tags:
            desc = tags[270].split(b"\0")[0].decode("latin-1")
    if not pages:
        raise ValueError("%s holds no image" % path)
    if any(p.shape != pages[0].shape or p.dtype != pages[0].dtype for p in pages):
        raise ValueError("%s: pages of different size or type" % path)
    rows, cols = pages[0].shape
    axes, shape = _tiff_axes_from_description(desc, len(pages), rows, cols)
    image = np.stack(pages).astype(pages[0].dtype.newbyteorder("=")).reshape(shape)
    metadata = dict(line.split("=", 1) for line in desc.splitlines() if "=" in line) if desc.startswith("ImageJ=") else None
    return image, axes, image.shape, metadata


def concatenate_time_points(files):
    """bim.py:478-495: the per-movie projection files (.npy, time first) as one uint16 movie.  A later movie with fewer
    channels (axes 1 .. ndim-3) than the first is zero-padded at the FRONT of that axis.  (The reference also means to
    resize frames of a different (Y, X) size but passes skimage a nested tuple and raises there; so does this.)"""
    movies = []
    for f in files:
        img = np.load(f).astype("uint16")
        if movies:
            first = movies[0]
            for dim in range(1, img.ndim - 2):
                missing = first.shape[dim] - img.shape[dim]
                if missing > 0:
                    pad = [(0, 0)] * img.ndim
                    pad[dim] = (missing, 0)
                    img = np.pad(img, pad_width=pad, constant_values=0)
            if img.shape[-2:] != first.shape[-2:]:
                raise ValueError("concatenate_time_points: frame size %s differs from the first movie's %s"
                                 % (img.shape[-2:], first.shape[-2:]))
        movies.append(img)
    return np.concatenate(movies, axis=0)


# ---- the remaining small operators and readers of bim.py -------------------------------------------------------------------
def binary_image(image, axes, thresholds):
    """bim.py:350-369: per channel, values above the threshold become 1, then values below it become 0 -- two passes in that
    order, so for a threshold above 1 the freshly written ones are zeroed again (as upstream) and values EQUAL to the
    threshold keep their value; `thresholds` is a scalar or one value per channel (the first one when the image has no
    channel axis, or the channel axis leads).  An elementwise pass over a display image: host numpy."""
    adjusted = np.copy(image)
    per_channel = hasattr(thresholds, "__len__")
    if axes.find("C") > 0:
        moved, order = put_channel_axis_first(adjusted, axes)
        for channel in range(moved.shape[0]):
            thr = thresholds[channel] if per_channel else thresholds
            plane = moved[channel]
            np.putmask(plane, plane > thr, 1)
            np.putmask(plane, plane < thr, 0)
        return np.transpose(moved, axes=np.argsort(order))
    thr = thresholds[0] if per_channel else thresholds
    np.putmask(adjusted, adjusted > thr, 1)
    np.putmask(adjusted, adjusted < thr, 0)
    return adjusted


def _whole(src):
    T, C, Z, Y, X = src.dims
    return src.block(slice(0, T), slice(0, C), slice(0, Z), slice(0, Y), slice(0, X))


def read_whole_image(path):
    """bim.py:54-57: (data TCZYX, dims, metadata) of the first scene."""
    src = open_image(path)
    return _whole(src), src.dims, getattr(getattr(src, "img", None), "metadata", None)


def read_virtual_image(path):
    """bim.py:59-62: upstream returns the lazy (dask) array; here the source object itself plays that role: index it
    through `.block(t, c, z, y, x)` slices, nothing is read before."""
    src = open_image(path)
    return src, src.dims, getattr(getattr(src, "img", None), "metadata", None)


def read_part_of_image(path, x_range, y_range, z_range, c_range, t_range, dims_order="TCZXY"):
    """bim.py:64-77: a sub-block of the first scene.  Kept as upstream wrote it: the z slice ends at c_range[1] (not
    z_range[1]), and a `dims_order` other than the default transposes the TCZYX block by the index of each letter of
    "TCZXY" in `dims_order`."""
    src = open_image(path)
    data = src.block(slice(t_range[0], t_range[1]), slice(c_range[0], c_range[1]), slice(z_range[0], c_range[1]),
                     slice(y_range[0], y_range[1]), slice(x_range[0], x_range[1]))
    default = "TCZXY"
    if default != dims_order:
        data = np.transpose(data, [dims_order.index(default[i]) for i in range(len(default))])
    return data, src.dims, getattr(getattr(src, "img", None), "metadata", None)


def extract_all_frames_from_a_scene(path, scene_index, max_frames=None):
    """bim.py:497-509: generator over the time points of one scene, each as a (Z, C, Y, X) array (upstream asks its reader for
    "TZCYX" order), at most `max_frames` of them."""
    src = open_image(path, scene_index)
    T, C, Z, Y, X = src.dims
    for t in range(min(T, max_frames) if max_frames else T):
        frame = src.block(slice(t, t + 1), slice(0, C), slice(0, Z), slice(0, Y), slice(0, X))[0]
        yield np.transpose(frame, (1, 0, 2, 3))
    return 0


class _BigTiffAppender(object):
    """Streaming BigTIFF writer (magic 43, 8-byte offsets): every (Y, X) plane appended becomes one page -- uncompressed,
    min-is-black, one strip -- so a movie larger than memory (or than classic TIFF's 4 GiB) can be written frame by frame."""

    def __init__(self, path):
        import struct
        self._s = struct
        self.fh = open(path, "wb")
        self.fh.write(struct.pack("<2sHHHQ", b"II", 43, 8, 0, 0))       # the first-IFD offset (bytes 8..15) is patched later
        self.link = 8                                                  # file position of the pointer to the next IFD

    def write_plane(self, plane):
        s = self._s
        plane = np.ascontiguousarray(plane)
        if plane.ndim != 2 or plane.dtype not in (np.uint8, np.uint16, np.float32, np.int32):
            raise TypeError("BigTIFF pages are 2-D uint8 / uint16 / int32 / float32 planes (got %s %s)" % (plane.dtype, plane.shape))
        rows, cols = plane.shape
        raw = plane.astype(plane.dtype.newbyteorder("<")).tobytes()
        at = self.fh.seek(0, 2)
        at += (-at) % 8
        self.fh.seek(at)
        self.fh.write(raw)
        ifd = at + len(raw)
        ifd += (-ifd) % 8
        fmt = 3 if plane.dtype.kind == "f" else (2 if plane.dtype.kind == "i" else 1)
        entries = [(256, 4, 1, cols), (257, 4, 1, rows), (258, 3, 1, plane.dtype.itemsize * 8), (259, 3, 1, 1), (262, 3, 1, 1),
                   (273, 16, 1, at), (277, 3, 1, 1), (278, 4, 1, rows), (279, 16, 1, len(raw)), (339, 3, 1, fmt)]
        self.fh.seek(ifd)
        self.fh.write(s.pack("<Q", len(entries)))
        for tag, typ, cnt, val in entries:
            self.fh.write(s.pack("<HHQQ", tag, typ, cnt, val))
        nxt = self.fh.tell()
        self.fh.write(s.pack("<Q", 0))
        self.fh.seek(self.link)
        self.fh.write(s.pack("<Q", ifd))
        self.link = nxt

    def close(self):
        self.fh.close()


def virtually_concatenate_time_points(files, position_indices, output_path="output.tif"):
    """bim.py:511-520: stream every time point of scene `position_indices[i] - 1` of `files[i]` into one BigTIFF, one
    (Z, C, Y, X) frame at a time (upstream: tifffile.TiffWriter(bigtiff=True).write per frame): pages in frame, z, channel
    order."""
    out = _BigTiffAppender(output_path)
    try:
        for path, scene in zip(files, position_indices):
            for frame in extract_all_frames_from_a_scene(path, scene - 1):
                for plane in frame.reshape((-1,) + frame.shape[-2:]):
                    out.write_plane(plane)
    finally:
        out.close()
