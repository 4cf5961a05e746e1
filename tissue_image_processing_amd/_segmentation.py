"""Device pipelines behind basic_image_manipulations.watershed_segmentation and the labelling helpers."""
import ctypes

import numpy as np

from . import _lib


def _taps(sigma):
    from .basic_image_manipulations import gaussian_taps
    return gaussian_taps(sigma) if float(sigma) > 1e-15 else None


def watershed(image, watershed_line=True, return_flags=False):
    """skimage.segmentation.watershed(image, markers=None, connectivity=1, watershed_line=True) (bim.py:475, pl.py:194).

    flags: bit0 = value ties between non-marker neighbours (serial push-age order not reproduced bit for bit),
    bit1 = two-valued image handled by the generation-synchronous BFS, bits 2.. = global-minimum fallback steps."""
    img = np.ascontiguousarray(image, dtype=np.float64)
    if img.ndim != 2:
        raise ValueError("watershed on MI355X takes 2-D images")
    labels = np.empty(img.shape, np.int32)
    flags = ctypes.c_int32(0)
    _lib.check(_lib.lib().tip_watershed_f64(_lib.ptr(img), _lib.ptr(labels), img.shape[0], img.shape[1],
                                            1 if watershed_line else 0, ctypes.byref(flags)))
    return (labels, flags.value) if return_flags else labels


def watershed_segmentation(image, imgthresh, stdeviation, blocksize, return_flags=False):
    """bim.py:446-476."""
    image = np.asarray(image)
    if image.ndim != 2:
        raise ValueError("watershed_segmentation takes a 2-D image")
    if image.dtype not in (np.float32, np.float64) and not np.issubdtype(image.dtype, np.integer):
        raise TypeError("watershed_segmentation on MI355X takes float or integer images (got %s)" % image.dtype)
    if blocksize % 2 == 0:
        blocksize += 1
    lib = _lib.lib()
    Y, X = image.shape
    labels = np.empty((Y, X), np.int32)
    flags = ctypes.c_int32(0)
    if image.dtype == np.float64:
        img = np.ascontiguousarray(image)
        d_img = _lib.DeviceBuffer(img.nbytes).upload(img)
        d_lab = _lib.DeviceBuffer(labels.nbytes)
        taps = _taps(stdeviation)
        _lib.check(lib.tip_watershed_segmentation_f64_dev(
            _lib.dptr(d_img.ptr), _lib.dptr(d_lab.ptr), Y, X, ctypes.c_double(imgthresh), _lib.ptr(taps),
            0 if taps is None else taps.size, int(blocksize), ctypes.byref(flags)))
        _lib.check(lib.tip_sync())
        labels = d_lab.download((Y, X), np.int32)
        d_img.free()
        d_lab.free()
    else:
        # float32 / integer image: the reference keeps the image dtype through thresholding and blurring
        # (bim.py:463-474); integer-valued landscapes have value ties, see the `flags` bit0 note in watershed()
        from .basic_image_manipulations import blur_image
        img64 = np.ascontiguousarray(image, dtype=np.float64)
        mx = np.empty_like(img64)
        _lib.check(lib.tip_rankfilter2d(_lib.ptr(img64), _lib.ptr(mx), 1, Y, X, blocksize, blocksize, 0, 1, 1))
        seg = np.copy(image)
        seg[seg < imgthresh * mx] = 0
        blurred = blur_image(seg, stdeviation)
        labels, fl = watershed(blurred, True, return_flags=True)
        flags.value = fl
    return (labels, flags.value) if return_flags else labels


def label(input, background=None, return_num=False, connectivity=None):
    """skimage.measure.label for 2-D integer images with connectivity 1 (ti.py:2922, 3470)."""
    a = np.asarray(input)
    if a.ndim != 2:
        raise ValueError("label on MI355X takes 2-D images")
    if connectivity != 1:
        raise NotImplementedError("label on MI355X implements connectivity=1 (what the reference passes)")
    if a.dtype == bool:
        a = a.astype(np.int32)
    if not np.issubdtype(a.dtype, np.integer):
        raise TypeError("label takes integer images")
    if a.size and (a.min() < -2 ** 31 or a.max() > 2 ** 31 - 1):
        raise ValueError("label values must fit int32")
    a32 = np.ascontiguousarray(a, dtype=np.int32)
    bg = 0 if background is None else int(background)
    out = np.empty(a32.shape, np.int32)
    n = ctypes.c_int32(0)
    _lib.check(_lib.lib().tip_label4_i32(_lib.ptr(a32), ctypes.c_int32(bg), _lib.ptr(out), a32.shape[0], a32.shape[1],
                                         ctypes.byref(n)))
    out = out.astype(np.int64)  # skimage returns the platform integer
    return (out, n.value) if return_num else out


def rank_filter(a, size, footprint_kind=0, mode="reflect", is_max=True):
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("2-D images only")
    if a.dtype == np.float64:
        dt = 1
    elif a.dtype == np.int32:
        dt = 2
    else:
        raise TypeError("rank filters take float64 or int32 images (got %s)" % a.dtype)
    ky, kx = (size, size) if np.isscalar(size) else size
    src = np.ascontiguousarray(a)
    out = np.empty_like(src)
    _lib.check(_lib.lib().tip_rankfilter2d(_lib.ptr(src), _lib.ptr(out), dt, src.shape[0], src.shape[1], int(ky), int(kx),
                                           int(footprint_kind), {"constant": 0, "reflect": 1}[mode], 1 if is_max else 0))
    return out


def maximum_filter(a, size=None, footprint=None, mode="reflect"):
    """scipy.ndimage.maximum_filter for the reference's call shapes (rectangles, and the 3x3 cross footprint)."""
    if footprint is not None:
        return rank_filter(a, 3, 1, mode, True)
    return rank_filter(a, size, 0, mode, True)


def minimum_filter(a, size=None, footprint=None, mode="reflect"):
    if footprint is not None:
        return rank_filter(a, 3, 1, mode, False)
    return rank_filter(a, size, 0, mode, False)


def regionprops_arrays(labels, intensity=None, n=None):
    """Per-label reductions (tip_regionprops_i32) -> dict of arrays over labels 1..n."""
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    if n is None:
        n = int(labels.max()) if labels.size else 0
    area = np.zeros(n, np.int64)
    bbox = np.zeros((n, 4), np.int64)
    sy = np.zeros(n, np.int64)
    sx = np.zeros(n, np.int64)
    pc = np.zeros((n, 3), np.int64)
    inten = None if intensity is None else np.ascontiguousarray(intensity, dtype=np.float64)
    isum = None if intensity is None else np.zeros(n, np.float64)
    if n > 0:
        _lib.check(_lib.lib().tip_regionprops_i32(_lib.ptr(labels), _lib.ptr(inten), labels.shape[0], labels.shape[1], n,
                                                  _lib.ptr(area), _lib.ptr(bbox), _lib.ptr(sy), _lib.ptr(sx), _lib.ptr(pc),
                                                  _lib.ptr(isum)))
    with np.errstate(invalid="ignore", divide="ignore"):
        cy = sy / area
        cx = sx / area
    sq2 = np.sqrt(2.0)
    perim = pc[:, 0] * 1.0 + pc[:, 1] * sq2 + pc[:, 2] * ((1 + sq2) / 2)
    out = dict(label=np.arange(1, n + 1), area=area, bbox=bbox, cy=cy, cx=cx, perimeter=perim)
    if isum is not None:
        with np.errstate(invalid="ignore", divide="ignore"):
            out["intensity_mean"] = isum / area
    return out


def neighbor_pairs(labels, cap=None):
    """Unique (hi, lo) pairs: a pixel labelled lo > 0 whose zero-padded 5x5 maximum is hi (ti.py:1822-1835)."""
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    grow = cap is None   # default capacity: planar label maps have ~3 pairs per cell; noisy maps get a bigger table
    if cap is None:
        cap = max(1024, 16 * (int(labels.max()) + 1))
    while True:
        pairs = np.empty((cap, 2), np.int32)
        n = ctypes.c_int64(0)
        rc = _lib.lib().tip_neighbor_pairs_i32(_lib.ptr(labels), labels.shape[0], labels.shape[1], _lib.ptr(pairs),
                                               ctypes.c_int64(cap), ctypes.byref(n))
        if rc == _lib.TIP_ERR_OVERFLOW and grow and cap < 8 * labels.size:
            cap *= 8
            continue
        _lib.check(rc)
        break
    p = pairs[:n.value].astype(np.int64)
    if p.size:
        p = p[np.lexsort((p[:, 1], p[:, 0]))]
    return p


def contact_pairs(labels, cap=None):
    """{(hi, lo): pixels} for the contact-length rule of ti.py:1844-1872: pixels whose cross-footprint maximum of the labels
    is hi and whose cross-footprint minimum (zeros replaced by max + 1) is lo, hi > lo >= 1.  One device pass."""
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    big = int(labels.max()) + 1
    grow = cap is None
    if cap is None:
        cap = max(4096, 16 * big)
    while True:
        pairs = np.empty((cap, 2), np.int32)
        counts = np.empty(cap, np.int64)
        n = ctypes.c_int64(0)
        rc = _lib.lib().tip_contact_pairs_i32(_lib.ptr(labels), labels.shape[0], labels.shape[1], big, _lib.ptr(pairs),
                                              _lib.ptr(counts), ctypes.c_int64(cap), ctypes.byref(n))
        if rc == _lib.TIP_ERR_OVERFLOW and grow and cap < 8 * labels.size:
            cap *= 8
            continue
        _lib.check(rc)
        break
    k = int(n.value)
    return {(int(h), int(l)): int(c) for (h, l), c in zip(pairs[:k].tolist(), counts[:k].tolist())}


def label_order_stats(labels, img, nlab, ranks):
    """(lo, hi): per label l+1 the values of 0-based ranks ranks[l] and ranks[l] + 1 among img's pixels of that label
    (hi == lo when there is no next one); ranks[l] < 0 skips the label.  labels None: one rank over the whole frame."""
    img = np.ascontiguousarray(img, dtype=np.float64)
    ranks = np.ascontiguousarray(ranks, dtype=np.int64)
    lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int32)
    lo = np.empty(nlab, np.float64)
    hi = np.empty(nlab, np.float64)
    _lib.check(_lib.lib().tip_label_order_stats_f64(_lib.ptr(lab), _lib.ptr(img), img.shape[0], img.shape[1], int(nlab),
                                                    _lib.ptr(ranks), _lib.ptr(lo), _lib.ptr(hi)))
    return lo, hi


def _lerp_percentile(lo, hi, gamma):
    """numpy's 'linear' percentile from the two neighbouring order statistics (function_base._lerp)."""
    diff = hi - lo
    return np.where(gamma >= 0.5, hi - diff * (1 - gamma), lo + diff * gamma)


def percentile_per_label(labels, img, nlab, counts, q):
    """np.percentile(img[labels == l + 1], q) for every label with counts[l] > 0 (others: nan), exact."""
    counts = np.asarray(counts, dtype=np.int64)
    present = counts > 0
    virt = (counts - 1) * (q / 100.0)
    prev = np.clip(np.floor(virt).astype(np.int64), 0, np.maximum(counts - 1, 0))
    gamma = virt - np.floor(virt)
    lo, hi = label_order_stats(labels, img, nlab, np.where(present, prev, -1))
    hi = np.where(prev + 1 <= counts - 1, hi, lo)
    return np.where(present, _lerp_percentile(lo, hi, gamma), np.nan)


def percentile_frame(img, q):
    """np.percentile(img, q) over all pixels, exact (radix select on the device, numpy's interpolation on the host)."""
    img = np.asarray(img)
    n = img.size
    virt = (n - 1) * (q / 100.0)
    prev = int(np.floor(virt))
    gamma = virt - np.floor(virt)
    lo, hi = label_order_stats(None, img.reshape(1, -1) if img.ndim != 2 else img, 1, np.array([prev], np.int64))
    hi = hi if prev + 1 <= n - 1 else lo
    return float(_lerp_percentile(lo, hi, gamma)[0])
