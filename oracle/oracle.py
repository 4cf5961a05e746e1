"""CPU ORACLE -- test infrastructure, NOT product code.

numpy + plain-C (oracle/tip_oracle.c) restatement of the reference's hot path
(SURVEY.md section 8a).  Every function cites the reference lines it follows.  It
is pinned by tests/golden/*.npz, which were produced by running the reference
itself under the build container's Anaconda interpreter
(tools/make_goldens.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libtip_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libtip_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_gaussian_weights.restype = ctypes.c_long
        _LIB.orc_label4_i32.restype = ctypes.c_long
        _LIB.orc_build_manifold_f32.restype = ctypes.c_long
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_MODES = {"nearest": 0, "reflect": 1, "constant": 2, "mirror": 3, "wrap": 4}


# ----------------------------------------------------------------------------- Gaussian (bim.py:373-390)
TAP_OVERRIDE = {}  # {sigma: taps}; tests inject the golden environment's taps here (np.exp differs by build)


def gaussian_kernel1d(sigma, truncate=4.0):
    """scipy/ndimage/filters.py:_gaussian_kernel1d (order 0) exactly as scipy's Python layer builds it.
    np.exp is not correctly rounded and differs between numpy builds in the last bit, so the taps belong to
    the environment; TAP_OVERRIDE lets the golden tests use the taps of the interpreter that made the goldens."""
    sd = float(sigma)
    if truncate == 4.0 and sd in TAP_OVERRIDE:
        return np.asarray(TAP_OVERRIDE[sd], dtype=np.float64)
    radius = int(truncate * sd + 0.5)
    sigma2 = sd * sd
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    return phi_x


def correlate1d(a, weights, axis, mode="nearest", cval=0.0):
    a = np.ascontiguousarray(a)
    assert a.dtype in (np.float32, np.float64)
    out = np.empty_like(a)
    shape3 = (1,) * (3 - a.ndim) + a.shape
    dims = (ctypes.c_long * 3)(*shape3)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    rc = lib().orc_correlate1d(_p(a), _p(out), 0 if a.dtype == np.float32 else 1, dims,
                               axis + (3 - a.ndim), _p(w), ctypes.c_long(w.size), _MODES[mode],
                               ctypes.c_double(cval))
    assert rc == 0
    return out


def gaussian_filter(a, sigma, mode="nearest", truncate=4.0):
    """scipy.ndimage.gaussian_filter as called by blur_image (bim.py:389): one correlate1d per axis in axis
    order, axes with sigma <= 1e-15 skipped, output dtype == input dtype (rounded after every axis)."""
    a = np.asarray(a)
    if a.dtype not in (np.float32, np.float64):
        raise TypeError("oracle gaussian_filter handles float32/float64 only")
    sig = np.ravel(np.asarray(sigma, dtype=np.float64))
    if sig.size == 1:
        sig = np.repeat(sig, a.ndim)
    if sig.size != a.ndim:
        raise RuntimeError("sequence argument must have length equal to input rank")
    out = a
    done = False
    for ax in range(a.ndim):
        if sig[ax] > 1e-15:
            out = correlate1d(out, gaussian_kernel1d(sig[ax], truncate)[::-1], ax, mode)
            done = True
    return out if done else a.copy()


def blur_image(image, std):
    """bim.py:373-390."""
    return gaussian_filter(image, std, mode="nearest")


# ----------------------------------------------------------------------------- display operations (bim.py:160-188, 233-414)
def img_as_float(image):
    """skimage.util.img_as_float as skimage.filters.gaussian applies it: unsigned integers are scaled by 1 / max of the
    dtype in float64, floats pass through."""
    image = np.asarray(image)
    if image.dtype.kind == "u":
        return np.multiply(image, 1.0 / np.iinfo(image.dtype).max, dtype=np.float64)
    if image.dtype.kind == "f":
        return image
    raise NotImplementedError("oracle: unsigned integer or float images")


def band_pass_filter(image, lowsigma, highsigma):
    """bim.py:393-414 -> skimage.filters.difference_of_gaussians: gaussian(low) - gaussian(high), mode 'nearest'."""
    f = img_as_float(image)
    return gaussian_filter(f, lowsigma) - gaussian_filter(f, highsigma)


def scoreatpercentile(a, per):
    """scipy.stats.scoreatpercentile(a, per) ('fraction'): weights (j - idx, idx - i) on the two neighbouring order
    statistics, divided by their sum."""
    v = np.sort(np.ravel(a))
    idx = per / 100.0 * (v.size - 1)
    i = int(idx)
    if i == idx:
        return v[i] * 1.0 / 1.0
    w = np.array([(i + 1) - idx, idx - i], float)
    return np.add.reduce(v[i:i + 2] * w) / w.sum()


def set_channel_brightness(image, max_possible_val, method="bestFit", clearExtreamPrecentage=1, minimum_pixel_val=0):
    """bim.py:299-348 (float64 channel, modified in place like upstream).  adjust_gamma with gamma 1 is the identity."""
    if clearExtreamPrecentage > 0:
        new_maximum = scoreatpercentile(image, 100 - clearExtreamPrecentage)
        new_minimum = scoreatpercentile(image, clearExtreamPrecentage)
        if minimum_pixel_val > 0:
            new_minimum = max(new_minimum, minimum_pixel_val)
        image[image > new_maximum] = new_maximum
    else:
        new_minimum = minimum_pixel_val
    if method in ("minMax", "bestFit"):
        image = image - new_minimum
        image = image / np.max(image)
        image = image + 1 / max_possible_val
        image[image < 0] = 0
    return image


def set_brightness(image, axes, metadata=None, method="bestFit", clearExtreamPrecentage=1, minVal=0, maxVal=0):
    """bim.py:233-297."""
    data_type = image.dtype
    max_possible_val = maxVal if maxVal else (255 if data_type == "uint8" else 65535 if data_type == "uint16" else 1)
    adjusted = np.copy(image).astype("double")
    minimum_pixel_val = max(minVal, 0)
    if metadata and "min" in metadata:
        minimum_pixel_val = metadata["min"]
    if axes.find("C") >= 0:
        adjusted, order = put_channel_axis_first(adjusted, axes)
        for ch in range(adjusted.shape[0]):
            adjusted[ch] = set_channel_brightness(adjusted[ch], max_possible_val, method, clearExtreamPrecentage,
                                                  minimum_pixel_val)
        adjusted = np.transpose(adjusted, axes=np.argsort(order))
    else:
        adjusted = set_channel_brightness(adjusted, max_possible_val, method, clearExtreamPrecentage, minimum_pixel_val)
    return adjusted


def stretch_for_display(disp_img, min_percent, max_percent):
    """gui.py:445-452 (display_frame): plane -> 255 * clip((plane - p_lo) / (p_hi - p_lo)), levels from np.percentile."""
    img = np.asarray(disp_img)
    lo, hi = percentile_linear(img, min_percent), percentile_linear(img, max_percent)
    if hi == lo:
        hi += 1
    out = img - lo
    np.putmask(out, out < 0, 0)
    out = 255 * out / (hi - lo)
    np.putmask(out, out > 255, 255)
    return out


def tiff_normalise(image, data_type):
    """The conversion save_tiff applies before writing (bim.py:183-186)."""
    if data_type and image.dtype != data_type and data_type in ("uint8", "uint16"):
        top = 255 if data_type == "uint8" else 65535
        return np.round((image / np.max(image)) * top).astype(data_type)
    return image


# ----------------------------------------------------------------------------- percentile (sp.py:33-36)
def percentile_linear(a, q):
    """np.percentile(a, q) for a 1-D float array with numpy 1.26.4 arithmetic (the oracle interpreter):
    virtual index (n-1)*q' with q' = q/100, previous/next order statistics, and
    _lerp: a + (b-a)*t, replaced by b - (b-a)*(1-t) when t >= 0.5; (b-a) is formed in the array dtype,
    the rest in float64.  Returns a float64 scalar."""
    a = np.sort(np.ravel(a))
    n = a.size
    quant = np.true_divide(q, 100)
    virtual = (n - 1) * quant  # numpy's 'linear' method: get_virtual_index = (n - 1) * quantiles
    prev = int(np.floor(virtual))
    gamma = np.float64(virtual - prev)
    prev = min(max(prev, 0), n - 1)
    nxt = min(prev + 1, n - 1)
    lo, hi = a[prev], a[nxt]
    diff = np.float64(hi - lo)  # subtraction in the array dtype, then widened
    res = np.float64(lo) + diff * gamma
    if gamma >= 0.5:
        res = np.float64(hi) - diff * (1 - gamma)
    return np.float64(res)


# ----------------------------------------------------------------------------- projection (sp.py:17-85)
def put_channel_axis_first(image, axes):
    """bim.py:199-231 (note: only transposes when the C axis index is > 0; order is C,(T),(Z),X,Y)."""
    c = axes.find("C")
    if c > 0:
        t, x, y, z = axes.find("T"), axes.find("X"), axes.find("Y"), axes.find("Z")
        order = (x, y)
        if z >= 0:
            order = (z,) + order
        if t >= 0:
            order = (t,) + order
        order = (c,) + order
        return np.transpose(image, axes=order), order
    return image, tuple(np.arange(len(axes)))


def _row_sum_f32(cols):
    """numpy's float32 add.reduce along a contiguous axis (the inner loop of block_reduce's np.mean / np.var):
    pairwise_sum (numpy/core/src/umath/loops_utils.h.src) over all n elements -- a plain running sum below 8 elements,
    else eight running partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the tail added one by one
    (blocks of at most 128 elements).  Pinned empirically against numpy 1.26.4 and 2.2.6.  `cols`: the elements."""
    m = len(cols)
    if m > 128:
        raise NotImplementedError("oracle: block sizes up to 128")
    if m < 8:
        res = cols[0].copy()
        for x in cols[1:]:
            res = res + x
        return res
    r = [cols[j].copy() for j in range(8)]
    i = 8
    while i < m - (m % 8):
        for j in range(8):
            r[j] = r[j] + cols[i + j]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < m:
        res = res + cols[i]
        i += 1
    return res


def _blocks(a, b):
    """skimage.measure.block_reduce's view: (Z, Y, X) float32 zero-padded to multiples of b -> (Z, Yb, b, Xb, b)."""
    a = np.asarray(a, np.float32)
    Z, Y, X = a.shape
    Yp, Xp = -(-Y // b) * b, -(-X // b) * b
    pad = np.zeros((Z, Yp, Xp), np.float32)
    pad[:, :Y, :X] = a
    return pad.reshape(Z, Yp // b, b, Xp // b, b)


def _block_sum(blk):
    """np.add.reduce of a (1, b, b) block in numpy's nditer order: every row of the block (contiguous in x) is reduced by
    the inner loop, the row sums are accumulated one after the other (pinned empirically against numpy 1.26.4)."""
    b = blk.shape[2]
    acc = None
    for r in range(b):
        row = _row_sum_f32([blk[:, :, r, :, c] for c in range(b)])
        acc = row if acc is None else acc + row
    return acc


def block_mean(a, b):
    """block_reduce(a, (1, b, b), func=np.mean) (sp.py:41,50), float32."""
    blk = _blocks(a, b)
    return _block_sum(blk) / np.float32(b * b)


def block_var(a, b):
    """block_reduce(a, (1, b, b), func=np.var) (sp.py:43,49): numpy's _var in float32 -- mean, squared deviations, mean."""
    blk = _blocks(a, b)
    mean = _block_sum(blk) / np.float32(b * b)
    dev = blk - mean[:, :, None, :, None]
    return _block_sum(dev * dev) / np.float32(b * b)


def _mirror_axis(n_in, n_out):
    """Per output index along one axis: the two input indices and weights of skimage.transform.resize(order=1,
    mode='reflect') -> scipy.ndimage.map_coordinates(order=1, mode='mirror') (skimage/transform/_warps.py:95-195):
    coordinate factor*(i+0.5)-0.5 in float64, mirrored at the ends, weights (1 - t, 1 - (1 - t))."""
    factor = np.float64(n_in) / np.float64(n_out)
    c = factor * (np.arange(n_out) + 0.5) - 0.5
    if n_in <= 1:
        z = np.zeros(n_out, np.int64)
        return z, z, np.ones(n_out), np.zeros(n_out)
    c = np.where(c < 0, -c, c)                      # |c| < 0.5 below zero: mirror about 0
    i0 = np.floor(c).astype(np.int64)
    t = c - np.floor(c)
    i1 = i0 + 1
    i1 = np.where(i1 >= n_in, 2 * n_in - 2 - i1, i1)  # beyond the last sample: mirror about n-1
    w0 = 1.0 - t
    w1 = 1.0 - w0
    return i0, i1, w0, w1


def resize_linear(a, out_yx):
    """skimage.transform.resize(a.astype('float32'), (Z, Y, X)) for a (Z, Yb, Xb) array (sp.py:58): the z factor is 1,
    so every plane is interpolated bilinearly; scipy accumulates the four corner terms in float64 in the order
    (y0,x0), (y0,x1), (y1,x0), (y1,x1), each as (v * wy) * wx, and rounds to float32."""
    a = np.asarray(a, np.float32).astype(np.float64)
    Y, X = out_yx
    y0, y1, wy0, wy1 = _mirror_axis(a.shape[1], Y)
    x0, x1, wx0, wx1 = _mirror_axis(a.shape[2], X)
    wy0, wy1 = wy0[None, :, None], wy1[None, :, None]
    wx0, wx1 = wx0[None, None, :], wx1[None, None, :]
    r0, r1 = a[:, y0, :], a[:, y1, :]
    t = (r0[:, :, x0] * wy0) * wx0
    t = t + (r0[:, :, x1] * wy0) * wx1
    t = t + (r1[:, :, x0] * wy1) * wx0
    t = t + (r1[:, :, x1] * wy1) * wx1
    return t.astype(np.float32)


def resize_warp2d(a, out_yx):
    """skimage.transform.resize(a.astype('float32'), (Y, X)) for a 2-D array (sp.py:64-65: the plane maps of build_manifold with
    bin_size > 1).  2-D arrays take skimage's bilinear warp (_warps_cy, a binary): source coordinate scale * i + (scale / 2 -
    1 / 2) evaluated in float32, corners floor / ceil with numpy 'reflect' borders (index -1 -> 1), (1 - dc) v00 + dc v01 for
    top and bottom, (1 - dr) top + dr bottom in double, float32 out.  Agrees with skimage to ~1e-6 (upstream's affine matrix
    comes out of a least-squares estimate); np.round of it equals the reference's on every golden, .5 ties included."""
    a = np.asarray(a, np.float32)
    Y, X = out_yx

    def axis(n_in, n_out):
        s = np.float64(n_in) / np.float64(n_out)
        f = np.float32(s) * np.arange(n_out, dtype=np.float32) + np.float32(0.5 * s - 0.5)
        lo, hi = np.floor(f).astype(np.int64), np.ceil(f).astype(np.int64)
        d = (f - lo.astype(np.float32)).astype(np.float64)

        def mirror(c):
            if n_in == 1:
                return np.zeros_like(c)
            p = 2 * (n_in - 1)
            c = np.abs(c) % p
            return np.where(c > n_in - 1, p - c, c)
        return mirror(lo), mirror(hi), d
    y0, y1, dr = axis(a.shape[0], Y)
    x0, x1, dc = axis(a.shape[1], X)
    v = a.astype(np.float64)
    dr, dc = dr[:, None], dc[None, :]
    top = (1.0 - dc) * v[y0][:, x0] + dc * v[y0][:, x1]
    bot = (1.0 - dc) * v[y1][:, x0] + dc * v[y1][:, x1]
    return ((1.0 - dr) * top + dr * bot).astype(np.float32)


def build_continues_manifold(score):
    """sp.py:87-165: the spiral z-map (C restatement orc_build_manifold_f32; int64 like upstream's astype(int))."""
    s = np.ascontiguousarray(score, dtype=np.float32)
    out = np.empty(s.shape[1:], np.int64)
    rc = lib().orc_build_manifold_f32(_p(s), ctypes.c_long(s.shape[0]), ctypes.c_long(s.shape[1]), ctypes.c_long(s.shape[2]), _p(out))
    if rc:
        raise TypeError("unsupported operand type(s) for -: 'NoneType' and 'int'")   # what upstream raises there
    return out


def time_point_surface_projection(time_point, axes, reference_channel, min_z=0, max_z=0,
                                  method="max_averages", bin_size=1, airyscan=True, z_map=False,
                                  atoh_shift=0, build_manifold=False, clip_from=None):
    """sp.py:17-85.  clip_from (not in the reference): the array whose non-zero 95th percentile
    clips the reference channel instead of the channel's own -- a spatial tile passes the whole frame's channel."""
    if axes.find("T") >= 0:
        time_point = time_point.reshape(time_point.shape[1:])
        image, _ = put_channel_axis_first(time_point, axes[1:])
    else:
        image, _ = put_channel_axis_first(time_point, axes)
    image = image.astype("float32")
    if airyscan:
        image -= 10000
        image[image < 0] = 0
    if max_z > 0:
        image = image[:, min_z:max_z, :, :]
    ch = np.copy(image[reference_channel])
    src = ch if clip_from is None else np.asarray(clip_from, np.float32)
    nz = src[src > 0]
    if nz.size > 0:
        p95 = percentile_linear(nz, 95)
        # numpy 1.x value-based casting: the float64 scalar is compared/assigned as float32
        p95_32 = np.float32(p95)
        ch[ch > p95_32] = p95_32
    ch = blur_image(ch, (0.5, 1, 1))
    z_size, y_size, x_size = image.shape[-3:]
    if bin_size > 1:
        if method == "max_averages":
            score = block_mean(blur_image(ch, (0.5, 30, 30)), bin_size)
        elif method == "max_std":
            score = block_var(ch, bin_size)
        elif method == "multi_channel":
            atoh = np.copy(image[(reference_channel + 1) % image.shape[0]])
            a95 = np.float32(percentile_linear(atoh, 95))          # all voxels here, zeros included (sp.py:46)
            atoh[atoh > a95] = a95
            atoh = blur_image(atoh, (0.5, 1, 1))
            score = block_mean(blur_image(atoh, (0.5, 30, 30)), bin_size) * block_var(ch, bin_size)
        else:
            raise TypeError("exceptions must derive from BaseException")   # `raise "No such method"` (sp.py:53)
        if not build_manifold:
            score = resize_linear(score, (y_size, x_size))
    else:
        score = blur_image(ch, (0.5, 30, 30))
    chosen_z = build_continues_manifold(score) if build_manifold else min_z + np.argmax(score, axis=0)   # (sp.py:56-61)
    chosen_z_atoh = np.copy(chosen_z) if atoh_shift == 0 else np.clip(chosen_z + atoh_shift, 0, score.shape[0])
    if chosen_z.shape != (y_size, x_size):          # the spiral ran on the binned score (sp.py:63-65)
        chosen_z = np.round(resize_warp2d(chosen_z.astype("float32"), (y_size, x_size))).astype("int")
        chosen_z_atoh = np.round(resize_warp2d(chosen_z_atoh.astype("float32"), (y_size, x_size))).astype("int")
    mask = np.zeros((z_size, y_size * x_size), np.float32)
    mask_atoh = np.zeros((z_size, y_size * x_size), np.float32)
    mask[chosen_z.ravel(), np.arange(x_size * y_size)] = 1
    mask_atoh[chosen_z_atoh.ravel(), np.arange(x_size * y_size)] = 1
    mask = blur_image(mask.reshape((z_size, y_size, x_size)), (1, 2, 2))
    mask_atoh = blur_image(mask_atoh.reshape((z_size, y_size, x_size)), (1, 2, 2))
    if axes.find("C") >= 0:
        channels = image.shape[0]
        projection = np.zeros((channels, y_size, x_size))
        for c in range(channels):
            m = mask if c == reference_channel else mask_atoh
            projection[c] = np.max(image[c] * m, axis=0)
    else:
        projection = np.max(image * mask, axis=0)
    return (projection, chosen_z) if z_map else projection


# ----------------------------------------------------------------------------- rank filters
def _minmax(a, size=None, footprint=None, mode="reflect", is_max=True):
    a = np.ascontiguousarray(a)
    if footprint is not None:
        fp = np.ascontiguousarray(np.asarray(footprint) != 0, dtype=np.uint8)
        ky, kx = fp.shape
        fpp = _p(fp)
    else:
        ky, kx = (size, size) if np.isscalar(size) else size
        fp, fpp = None, None
    if a.dtype == np.float64:
        fn = lib().orc_minmax2d_f64
    elif a.dtype == np.int32:
        fn = lib().orc_minmax2d_i32
    else:
        raise TypeError(a.dtype)
    out = np.empty_like(a)
    rc = fn(_p(a), _p(out), ctypes.c_long(a.shape[0]), ctypes.c_long(a.shape[1]), ctypes.c_long(ky),
            ctypes.c_long(kx), fpp, _MODES[mode], 1 if is_max else 0)
    assert rc == 0
    return out


def maximum_filter(a, size=None, footprint=None, mode="reflect"):
    """scipy.ndimage.maximum_filter (ti.py:1822 5x5 constant; ti.py:2081,2969 3x3 constant; ti.py:4079 cross)."""
    return _minmax(a, size, footprint, mode, True)


def minimum_filter(a, size=None, footprint=None, mode="reflect"):
    """scipy.ndimage.minimum_filter (ti.py:4083 cross footprint, constant)."""
    return _minmax(a, size, footprint, mode, False)


def dilation(a, k):
    """skimage.morphology.dilation with a k x k ones footprint (pl.py:170,173,193) = max filter, reflect."""
    return _minmax(np.asarray(a, np.float64), k, None, "reflect", True)


def erosion(a, k):
    """skimage.morphology.erosion with a k x k ones footprint (pl.py:171,174,191) = min filter, reflect."""
    return _minmax(np.asarray(a, np.float64), k, None, "reflect", False)


def threshold_local_generic_max(image, imgthresh, blocksize):
    """bim.py:464-472: threshold_local(method='generic', param=imgthresh*max) == imgthresh * max filter
    (block x block, reflect) evaluated in float64 (skimage/filters/thresholding.py:210-213)."""
    if blocksize % 2 == 0:
        blocksize += 1
    return imgthresh * _minmax(np.asarray(image, np.float64), blocksize, None, "reflect", True)


# ----------------------------------------------------------------------------- labelling / watershed
def label4(a, background=0):
    """skimage.measure.label(a, connectivity=1, background=bg) (ti.py:2922,3470)."""
    a = np.ascontiguousarray(a, dtype=np.int32)
    out = np.empty_like(a)
    n = lib().orc_label4_i32(_p(a), ctypes.c_int32(background), _p(out), ctypes.c_long(a.shape[0]),
                             ctypes.c_long(a.shape[1]))
    return out, int(n)


def local_minima(img):
    """skimage.morphology.local_minima(img, connectivity=1) (called by watershed, _watershed.py:75)."""
    img = np.ascontiguousarray(img, dtype=np.float64)
    out = np.empty(img.shape, np.uint8)
    rc = lib().orc_local_minima_f64(_p(img), _p(out), ctypes.c_long(img.shape[0]), ctypes.c_long(img.shape[1]))
    assert rc == 0
    return out


def watershed(img, markers=None, watershed_line=True):
    """skimage.segmentation.watershed(img, markers=None, connectivity=1, watershed_line=...) (bim.py:475, pl.py:194)."""
    img = np.ascontiguousarray(img, dtype=np.float64)
    if markers is None:
        markers, _ = label4(local_minima(img).astype(np.int32), 0)
    lab = np.ascontiguousarray(markers, dtype=np.int32).copy()
    rc = lib().orc_watershed_f64(_p(img), _p(lab), ctypes.c_long(img.shape[0]), ctypes.c_long(img.shape[1]),
                                 1 if watershed_line else 0)
    assert rc == 0
    return lab


def equal_key_pop_order(c):
    """Order in which skimage's heap pops M equal-keyed markers when marker i pushes c[i] larger entries as it pops
    (the marker phase of watershed() on a two-valued image, pl.py:194): order[t] = raster rank of the t-th popped marker."""
    c = np.ascontiguousarray(c, dtype=np.uint8)
    order = np.empty(c.size, np.int32)
    rc = lib().orc_equal_key_pop_order(_p(c), ctypes.c_long(c.size), _p(order))
    assert rc == 0
    return order


def watershed_segmentation(image, imgthresh, stdeviation, blocksize):
    """bim.py:446-476 (the 4-argument definition that shadows bim.py:417-443)."""
    seg = np.copy(image)
    thr = threshold_local_generic_max(seg, imgthresh, blocksize)
    seg[seg < thr] = 0
    blurred = blur_image(seg, stdeviation)
    return watershed(blurred, watershed_line=True)


# ----------------------------------------------------------------------------- cell tables (ti.py:880-909, 1815-1842)
_SQ2 = np.sqrt(2.0)


def regionprops(labels, intensity=None):
    """skimage.measure.regionprops_table(labels, properties=[label, area, perimeter, centroid, bbox]) (ti.py:891)
    as a dict of arrays over labels 1..max (absent labels have area 0)."""
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    n = int(labels.max()) if labels.size else 0
    area = np.zeros(n, np.int64)
    bbox = np.zeros((n, 4), np.int64)
    sy = np.zeros(n, np.int64)
    sx = np.zeros(n, np.int64)
    pc = np.zeros((n, 3), np.int64)
    isum = np.zeros(n, np.float64) if intensity is not None else None
    inten = np.ascontiguousarray(intensity, dtype=np.float64) if intensity is not None else None
    rc = lib().orc_regionprops_i32(_p(labels), _p(inten) if inten is not None else None,
                                   ctypes.c_long(labels.shape[0]), ctypes.c_long(labels.shape[1]), ctypes.c_long(n),
                                   _p(area), _p(bbox), _p(sy), _p(sx), _p(pc), _p(isum) if isum is not None else None)
    assert rc == 0
    with np.errstate(invalid="ignore", divide="ignore"):
        cy = sy / area
        cx = sx / area
    perim = pc[:, 0] * 1.0 + pc[:, 1] * _SQ2 + pc[:, 2] * ((1 + _SQ2) / 2)
    res = dict(label=np.arange(1, n + 1), area=area, bbox=bbox, cy=cy, cx=cx, perimeter=perim)
    if isum is not None:
        with np.errstate(invalid="ignore", divide="ignore"):
            res["intensity_mean"] = isum / area
    return res


def neighbor_pairs(labels):
    """Unique (hi, lo) label pairs such that a pixel labelled lo>0 has hi as the max of its 5x5 window
    (zero padded) -- the relation find_neighbors evaluates with labels[dilated == i] (ti.py:1822-1835)."""
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    dil = maximum_filter(labels, (5, 5), mode="constant")
    sel = (labels > 0) & (dil != labels)
    pairs = np.unique(np.stack([dil[sel], labels[sel]], axis=1), axis=0)
    return pairs.astype(np.int64)


def frame_cellinfo(labels, min_cell_area=0.1, max_cell_area=10):
    """Tissue.calculate_frame_cellinfo + find_neighbors (ti.py:880-909, 1815-1842) as plain arrays.
    Returns dict with area, perimeter, label, cx, cy, bbox columns, valid, neighbors (list of sets), n_neighbors."""
    n = int(np.max(labels))
    rp = regionprops(labels)
    present = rp["area"] > 0
    area = np.where(present, rp["area"], 0)
    mean_area = np.mean(area)
    valid = np.logical_and(area < max_cell_area * mean_area, area > min_cell_area * mean_area).astype(int)
    pairs = neighbor_pairs(labels)
    neighbors = [set() for _ in range(n)]
    working = set((np.nonzero(valid)[0] + 1).tolist())
    for hi, lo in pairs:
        if int(hi) in working:
            neighbors[hi - 1].add(int(lo))
            neighbors[lo - 1].add(int(hi))
    n_nb = np.array([len(s) for s in neighbors])
    return dict(label=np.where(present, rp["label"], 0), area=area,
                perimeter=np.where(present, rp["perimeter"], 0.0),
                cx=np.where(present, rp["cx"], 0.0), cy=np.where(present, rp["cy"], 0.0),
                bbox=np.where(present[:, None], rp["bbox"], 0), valid=valid, neighbors=neighbors, n_neighbors=n_nb)


def update_labels(labels):
    """Tissue.update_labels (ti.py:2967-2970): negative pixels take the 3x3 zero-padded maximum."""
    labels = np.ascontiguousarray(labels, dtype=np.int32).copy()
    dil = maximum_filter(labels, (3, 3), mode="constant")
    labels[labels < 0] = dil[labels < 0]
    return labels


def contact_matrix(labels, neighbors):
    """calc_neighbors_contact_matrix / calculate_contact_length (ti.py:1844-1872, 4073-4094): for each cell and
    each of its neighbours, the number of pixels whose cross-footprint max is the larger label and whose
    cross-footprint min (zeros replaced by max+1) is the smaller one, inside bbox +-2."""
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    cross = np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]])
    mx = maximum_filter(labels, footprint=cross, mode="constant")
    lc = labels.copy()
    lc[lc == 0] = labels.max() + 1
    mn = minimum_filter(lc, footprint=cross, mode="constant")
    n = len(neighbors)
    rp = regionprops(labels)
    out = np.zeros((n, n))
    for i in range(n):
        if rp["area"][i] == 0:
            r0 = c0 = 0
            r1 = c1 = 2
        else:
            b = rp["bbox"][i]
            r0, c0, r1, c1 = max(0, b[0] - 2), max(0, b[1] - 2), b[2] + 2, b[3] + 2
        mxr, mnr = mx[r0:r1, c0:c1], mn[r0:r1, c0:c1]
        for nb in neighbors[i]:
            hi, lo = max(i + 1, nb), min(i + 1, nb)
            out[i, nb - 1] = np.sum(np.logical_and(mxr == hi, mnr == lo))
    return out


def track_simple(labels_list, tables, drifts):
    """Tissue.track_cells_iterator with existing drifts (ti.py:2037-2113): previous centroids minus drift are
    looked up in the 3x3-max-filtered label map of the current frame; ids propagate, duplicates are resolved by
    np.unique first-occurrence, unmatched cells get fresh ids.  Returns the per-frame id arrays."""
    ids0 = tables[0]["label"].astype(np.int64).copy()
    unl = ids0 == 0
    last = ids0.max()
    ids0[unl] = np.arange(last + 1, last + unl.sum() + 1)
    out = [ids0]
    cx_prev = tables[0]["cx"].astype(np.float64).copy()
    cy_prev = tables[0]["cy"].astype(np.float64).copy()
    ids_prev = ids0
    for f in range(1, len(labels_list)):
        cx_prev = cx_prev - drifts[f][1]
        cy_prev = cy_prev - drifts[f][0]
        lab = maximum_filter(np.ascontiguousarray(labels_list[f], np.int32), (3, 3), mode="constant")
        n_cur = tables[f]["cx"].size
        ids = np.zeros(n_cur, np.int64)
        idx = -1 * np.ones(cy_prev.shape)
        yl = np.round(cy_prev).astype(int)
        xl = np.round(cx_prev).astype(int)
        ok = (0 <= yl) & (yl < lab.shape[0]) & (0 <= xl) & (xl < lab.shape[1])
        idx[ok] = lab[yl[ok], xl[ok]] - 1
        lp = ids_prev[idx >= 0]
        idx = idx[idx >= 0]
        _, loc = np.unique(lp, return_index=True)
        idx, lp = idx[loc], lp[loc]
        _, loc = np.unique(idx, return_index=True)
        idx, lp = idx[loc], lp[loc]
        ids[idx.astype(int)] = lp
        unl = ids == 0
        last = ids.max()
        ids[unl] = np.arange(last + 1, last + unl.sum() + 1)
        out.append(ids)
        cx_prev = tables[f]["cx"].astype(np.float64).copy()
        cy_prev = tables[f]["cy"].astype(np.float64).copy()
        ids_prev = ids
    return out


def tracking_labels(labels, ids):
    """Tissue.get_trackking_labels (ti.py:4021-4028): per-pixel label -> track id LUT gather."""
    lut = np.insert(np.asarray(ids), 0, 0)
    return lut[labels]


def closing_tail(p0, thr=0.1):
    """prediction_local.py:167-194 after the network: threshold, 5x5 closing x101, 7x7 erosion, boundary, watershed."""
    hcb = np.zeros(p0.shape)
    hcb[p0 > thr] = 255
    d = dilation(hcb, 5)
    e = erosion(d, 5)
    for _ in range(100):
        d = dilation(e, 5)
        e = erosion(d, 5)
    hc = erosion(e, 7)
    bound = e - hc
    boundary = dilation(bound, 5)
    return watershed(boundary, watershed_line=True), hc, boundary, e


# ----------------------------------------------------------------------------- drift (ti.py:1982-2035, bim.py:522-536)
def phase_cross_correlation(reference_image, moving_image, upsample_factor=1):
    """skimage.registration.phase_cross_correlation (0.18.3, space='real', no normalisation) restated with numpy.fft:
    whole-pixel peak of ifft2(F1 * conj(F2)), then the matrix-multiply upsampled DFT in a 1.5-pixel neighbourhood
    (skimage/registration/_phase_cross_correlation.py:11-76, 196-262).  Returns the shift vector only."""
    a = np.asarray(reference_image, dtype=np.float64)
    b = np.asarray(moving_image, dtype=np.float64)
    src, tgt = np.fft.fft2(a), np.fft.fft2(b)
    shape = src.shape
    prod = src * tgt.conj()
    cc = np.fft.ifft2(prod)
    maxima = np.unravel_index(np.argmax(np.abs(cc)), cc.shape)
    mid = np.array([np.fix(s / 2) for s in shape])
    shifts = np.stack(maxima).astype(np.float64)
    shifts[shifts > mid] -= np.array(shape)[shifts > mid]
    if upsample_factor > 1:
        shifts = np.round(shifts * upsample_factor) / upsample_factor
        region = int(np.ceil(upsample_factor * 1.5))
        dftshift = np.fix(region / 2.0)
        uf = float(upsample_factor)
        offs = dftshift - shifts * uf
        data = prod.conj()
        for (n_items, ax_off) in list(zip(shape, offs))[::-1]:
            kernel = (np.arange(region) - ax_off)[:, None] * np.fft.fftfreq(n_items, uf)
            kernel = np.exp(-1j * 2 * np.pi * kernel)
            data = np.tensordot(kernel, data, axes=(1, -1))
        cc2 = data.conj()
        mx = np.unravel_index(np.argmax(np.abs(cc2)), cc2.shape)
        shifts = shifts + (np.stack(mx).astype(np.float64) - dftshift) / uf
    for d in range(2):
        if shape[d] == 1:
            shifts[d] = 0
    return shifts


def update_drift(previous_img, current_img):
    """Tissue.update_drift without stage locations (ti.py:1982-2035): returns (shift_y, shift_x) = (refined[-1], refined[-2])."""
    r = phase_cross_correlation(previous_img, current_img, upsample_factor=100)
    return r[-1], r[-2]


# ---- overlays (ti.py:584-607, 2585-2645) ------------------------------------------------------------------------------------------
def disk_mask(shape, center, radius):
    """skimage.draw.disk(center, radius, shape=shape) as a boolean image (skimage/draw/draw.py: ellipse / _ellipse_in_shape: the
    pixels of the box ceil(center - r) .. floor(center + r), clipped to the image, with ((r - r0) / R)^2 + ((c - c0) / R)^2 < 1,
    coordinates relative to the box)."""
    Y, X = int(shape[0]), int(shape[1])
    r0, c0 = float(center[0]), float(center[1])
    ur, uc = max(int(np.ceil(r0 - radius)), 0), max(int(np.ceil(c0 - radius)), 0)
    lr, lc = min(int(np.floor(r0 + radius)), Y - 1), min(int(np.floor(c0 + radius)), X - 1)
    out = np.zeros((Y, X), bool)
    if lr < ur or lc < uc:
        return out
    rr = np.arange(0, lr - ur + 1, dtype=np.float64)[:, None] - (r0 - ur)
    cc = np.arange(0, lc - uc + 1, dtype=np.float64)[None, :] - (c0 - uc)
    out[ur:lr + 1, uc:lc + 1] = (rr / radius) ** 2 + ((-cc) / radius) ** 2 < 1
    return out


def draw_disks(shape, centers, radius, colors):
    """(3, Y, X) float64: discs painted in order, a later one over an earlier one"""
    out = np.zeros((3,) + tuple(shape), np.float64)
    for (cy, cx), col in zip(centers, colors):
        m = disk_mask(shape, (cy, cx), radius)
        for j in range(3):
            out[j][m] = col[j]
    return out


def line_pixels(r0, c0, r1, c1):
    """skimage.draw.line (skimage/draw/_draw.pyx `_line`, a binary in the container: the published integer Bresenham walk)"""
    r, c = int(r0), int(c0)
    dr, dc = abs(int(r1) - r), abs(int(c1) - c)
    sc = 1 if (int(c1) - c) > 0 else -1
    sr = 1 if (int(r1) - r) > 0 else -1
    steep = dr > dc
    if steep:
        c, r = r, c
        dc, dr = dr, dc
        sc, sr = sr, sc
    d = 2 * dr - dc
    pix = []
    for _ in range(dc):
        pix.append((c, r) if steep else (r, c))
        while d >= 0:
            r += sr
            d -= 2 * dc
        c += sc
        d += 2 * dr
    pix.append((int(r1), int(c1)))
    return pix


def draw_lines(shape, ends, color):
    img = np.zeros(tuple(shape), np.float64)
    for r0, c0, r1, c1 in ends:
        for r, c in line_pixels(r0, c0, r1, c1):
            img[r, c] = 1
    return np.tile(img, (3, 1, 1)) * np.asarray(color, np.float64).reshape(3, 1, 1)


def draw_cell_types(cell_types, type_index, pos_color=(1, 0, 1), neg_color=(1, 1, 0)):
    """ti.py:2585-2593 with is_positive_for_type's bit test (ti.py:164-176): scalar type index"""
    t = np.asarray(cell_types).astype(np.uint8)
    bit = np.uint8(1 << type_index)
    pos = ((t & bit) == bit) & (t != 255)
    neg = ~pos & (t != 255)
    return pos * np.asarray(pos_color, np.float64).reshape(3, 1, 1) + neg * np.asarray(neg_color, np.float64).reshape(3, 1, 1)


def draw_tracking(track, cycle=((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1))):
    """ti.py:2625-2635"""
    track = np.asarray(track)
    out = np.zeros((3,) + track.shape, np.float64)
    for i in range(len(cycle)):
        for j in range(3):
            out[j][track % len(cycle) == i] = cycle[i][j]
    for j in range(3):
        out[j][track == 0] = 0
    return out
