"""Drop-in for Segmentation/prediction_local.py (pl.py) on MI355X.

    find_desired_shape, normalize_channel                      pl.py:10-29
    build_unet_model / SegmentationPredictor                   pl.py:31-198

The U-Net's dense 3x3 convolutions run through PyTorch-ROCm (MIOpen -> MFMA), as BASELINE.json's north_star
prescribes for the weight/conv path; everything after the network (threshold, 5x5 closing, 7x7 erosion, boundary,
watershed: pl.py:167-194) runs in libtissue_hip.so on the same resident buffers.

Deliberate deviations from the reference (documented in DESIGN.md):
  * pl.py:96-135 writes nine debug TIFFs to hard-coded C:\\Users\\... paths on every call; not reproduced.
  * the 101-fold closing loop (pl.py:170-174) is applied once: grey closing with a flat footprint is idempotent
    (pinned by tests/golden/rank_filters.npz closed_101 == closed_once).
  * Keras .h5 checkpoints (gui.py:38-39, pl.py:76-88) are read by a self-contained HDF5 reader (_hdf5.py: no h5py /
    TensorFlow in the image); an .npz holding model.get_weights() in layer order is accepted too, and a path of None
    gives random initialisation.  No weights ship with the reference, so the TRAINED network's parity is unpinned.
"""
import ctypes
import os
import threading

import numpy as np

from . import _lib


def find_desired_shape(shape_y, shape_x):
    """pl.py:10-19: the smallest power of two >= each extent (the network halves the frame three times)."""
    if shape_y < 1 or shape_x < 1:
        raise UnboundLocalError("find_desired_shape: extents must be >= 1")   # what the reference's empty loop ends in
    return 1 << (int(shape_y) - 1).bit_length(), 1 << (int(shape_x) - 1).bit_length()


def normalize_channel(image):
    """pl.py:21-29: clip to the [1st, 99th] percentile and scale to [0, 1].

    The reference writes the two percentiles INTO a copy of the input, so the clip values take the input's dtype: for
    the uint16 planes the GUI hands over (gui.py:2059-2061) they are truncated to integers, for float32 they are rounded
    to float32 (and numpy 1.x, the reference's environment, then keeps the arithmetic in float32); float64 is plain."""
    image = np.asarray(image)
    per99 = np.percentile(image, 99)
    per1 = np.percentile(image, 1)
    kind = image.dtype
    clipped = np.where(image > per99, np.asarray(per99).astype(kind), image)
    clipped = np.where(image < per1, np.asarray(per1).astype(kind), clipped)
    if kind == np.float32:
        return (clipped - np.float32(per1)) / np.float32(per99 - per1)
    return (clipped - per1) / (per99 - per1)


_shared_runtime = {}


def shares_runtime_with_torch(t):
    """True when libtissue_hip.so and torch use ONE HIP runtime for tensor t's device: the library's runtime attributes
    t's storage to t.device.index.  PyTorch wheels bundle their own libamdhip64; with two runtimes in the process torch's
    pointers and stream handles mean nothing to the library, and the fused passes that take them must not be used (the torch
    expressions run instead).  Checked once per device."""
    idx = t.device.index if t.device.index is not None else 0
    ok = _shared_runtime.get(idx)
    if ok is None:
        ok = _lib.load().tip_pointer_device(_lib.dptr(t.data_ptr())) == idx
        _shared_runtime[idx] = ok
    return ok


_MODES = {  # mode -> (pieces per value, piece format of the C-ABI: 0 bf16, 1 fp16, products per term, dropped part of a term)
    "f16x3": (2, 1, 3, "7.2e-7"),
    "bf16x3": (2, 0, 3, "1.6e-5"),
    "bf16x6": (3, 0, 6, "9e-8"),
}
_F16_ACT_SCALE = 16.0      # fp16 pieces: activations are stored times 2^4 (saturate beyond |v| = 4094, absolute floor 2^-29)


def _unet_mode():
    """TISSUE_HIP_UNET_ARITH: how the hand-written implicit-GEMM convolutions feed the 16-bit matrix cores (float32 accumulation):
    'f16x3' (default): float32 operands split into two fp16 pieces of power-of-two-scaled values, three products per term --
        float32-equivalent (<= 3 x 2^-22 of a term is dropped) at three products;
    'bf16x3': two bf16 pieces, three products per term (<= 2^-15.9 of a term dropped: 16 significand bits);
    'bf16x6': three bf16 pieces, six products (float32-equivalent to 2^-23.4; twice the matrix work);
    'miopen': PyTorch-ROCm / MIOpen float32 convolutions."""
    m = os.environ.get("TISSUE_HIP_UNET_ARITH", "f16x3")
    if m not in _MODES and m != "miopen":
        raise ValueError("TISSUE_HIP_UNET_ARITH must be f16x3, bf16x3, bf16x6 or miopen")
    if os.environ.get("TISSUE_HIP_UNET_DTYPE", "fp32") != "fp32":
        m = "miopen"
    return m


def unet_arithmetic():
    """How the network's convolutions are computed in this process (bench.py reports it with the MFMA roofline)."""
    m = _unet_mode()
    if m == "miopen":
        dt = os.environ.get("TISSUE_HIP_UNET_DTYPE", "fp32")
        if dt == "fp32":
            return {"dtype": "f32", "arithmetic": "float32 convolutions through PyTorch-ROCm / MIOpen (fp32 matrix pipe)", "peak_tflops": 157.3}
        return {"dtype": dt, "arithmetic": "%s convolutions through PyTorch-ROCm / MIOpen" % dt, "peak_tflops": 2500.0}
    planes, fmt, prods, err = _MODES[m]
    piece = "fp16" if fmt else "bf16"
    return {"dtype": "f32(%s)" % m, "mode": m, "peak_tflops": 2500.0 / prods, "issued_flops_factor": prods, "issued_peak_tflops": 2500.0,
            "float32_equivalent": m != "bf16x3",
            "arithmetic": "float32 operands split into %d %s pieces, %d %s MFMA products per term (relative error per term <= %s), "
                          "float32 accumulation; hand-written implicit-GEMM kernels (csrc/tip_unet_conv.h); peak = dense 16-bit MFMA "
                          "peak / %d" % (planes, piece, prods, piece, err, prods)}


def unet_algorithmic_bytes(h, w):
    """Compulsory HBM bytes of one forward pass at float32 width (every layer's input read once, its output written once, the
    split weights read once): what `roofline.traffic` of the network's kernels is compared with."""
    total = 0
    c, hh, ww = 2, h, w
    layers = []
    for f in _FILTERS:
        layers += [(hh, ww, c, f), (hh, ww, f, f)]
        c, hh, ww = f, hh // 2, ww // 2
    layers += [(hh, ww, c, 1024), (hh, ww, 1024, 1024)]
    c = 1024
    for f in reversed(_FILTERS):
        total += 4 * (hh * ww * c + 4 * hh * ww * f) + 4 * 9 * c * f          # transposed convolution: in, out (2h x 2w), weights
        hh, ww = hh * 2, ww * 2
        layers += [(hh, ww, 2 * f, f), (hh, ww, f, f)]
        c = f
    for (lh, lw, ci, co) in layers:
        total += 4 * lh * lw * (ci + co) + 4 * 9 * ci * co
    total += 4 * hh * ww * (c + 2)                                            # head
    total += 4 * sum((h >> (i + 1)) * (w >> (i + 1)) * f for i, f in enumerate(_FILTERS))   # pooled maps (written by the conv epilogue)
    return total


class _ConvDesc(ctypes.Structure):
    """tip_unet_conv_desc of include/tissue_hip.h."""
    _fields_ = [("in0", ctypes.c_void_p), ("in1", ctypes.c_void_p), ("c0", ctypes.c_int), ("c1", ctypes.c_int), ("h", ctypes.c_int),
                ("w", ctypes.c_int), ("planes", ctypes.c_int), ("weights", ctypes.c_void_p), ("ntaps", ctypes.c_int),
                ("dy", ctypes.c_int * 9), ("dx", ctypes.c_int * 9), ("cout", ctypes.c_int), ("bias", ctypes.c_void_p),
                ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("out", ctypes.c_void_p), ("out_h", ctypes.c_int),
                ("out_w", ctypes.c_int), ("sy", ctypes.c_int), ("sx", ctypes.c_int), ("oy", ctypes.c_int), ("ox", ctypes.c_int),
                ("pool_out", ctypes.c_void_p), ("head_w", ctypes.c_void_p), ("head_b", ctypes.c_void_p), ("head_out", ctypes.c_void_p),
                ("format", ctypes.c_int), ("acc_scale", ctypes.c_float), ("tf", ctypes.c_int), ("nmask", ctypes.c_int * 9)]


_FILTERS = (128, 256, 512)
_BN_EPS = 1e-3  # Keras BatchNormalization default


def _percentile_linear_t(flat_sorted, q):
    """np.percentile(..., q) ('linear') on an ascending torch tensor, numpy's index / lerp arithmetic in float64."""
    n = flat_sorted.numel()
    quant = q / 100.0
    virt = (n - 1) * quant
    prev = int(np.floor(virt))
    gamma = virt - prev
    prev = min(max(prev, 0), n - 1)
    nxt = min(prev + 1, n - 1)
    lo = float(flat_sorted[prev])
    hi = float(flat_sorted[nxt])
    diff = hi - lo
    res = lo + diff * gamma
    if gamma >= 0.5:
        res = hi - diff * (1 - gamma)
    return res


_FORWARD_GATES = {}
_FORWARD_GATES_LOCK = threading.Lock()


def _forward_gate(device_index):
    with _FORWARD_GATES_LOCK:
        g = _FORWARD_GATES.get(device_index)
        if g is None:
            g = _FORWARD_GATES[device_index] = {"lock": threading.Lock(), "event": None, "stream": None}
        return g


_SIDE = threading.local()


def _side_streams(device):
    """three extra streams per (thread, device) for the transposed convolution's parity classes"""
    import torch
    d = getattr(_SIDE, "streams", None)
    if d is None:
        d = _SIDE.streams = {}
    if device.index not in d:
        d[device.index] = [torch.cuda.Stream(device=device) for _ in range(3)]
    return d[device.index]


class _UNet(object):
    """3-level U-Net of pl.py:31-72 as explicit torch functional calls (NCHW, channels_last memory format).

    Keras semantics kept: Conv2D(3, 'same') + ReLU THEN BatchNormalization (eps 1e-3, inference statistics);
    MaxPool2D(2); Dropout is identity at inference; Conv2DTranspose(n, 3, 2, 'same') == torch conv_transpose2d
    (stride 2, padding 0) cropped to the first 2N rows/cols (TF pads SAME with pad_before 0 / pad_after 1);
    concatenate([upsampled, skip]); Conv2D(2, 1) + softmax over channels.
    """

    def __init__(self, in_ch=2, device="cuda", dtype=None, weights=None, seed=0, filters=None, bottleneck=None):
        import torch
        self.torch = torch
        self.device = device
        if weights is not None:                   # the widths are the checkpoint's (12 arrays per double block: 2 x (kernel, bias, 4 BN vectors))
            weights = list(weights)
            if len(weights) != 12 * 7 + 2 * 3 + 2:
                raise ValueError("the U-Net of pl.py:31-72 has 92 weight arrays, the checkpoint holds %d" % len(weights))
            filters = tuple(int(weights[12 * i].shape[3]) for i in range(3))
            bottleneck = int(weights[36].shape[3])
        self.filters = tuple(filters) if filters is not None else _FILTERS
        self.bottleneck = int(bottleneck) if bottleneck is not None else 1024
        self.dtype = dtype or {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[
            os.environ.get("TISSUE_HIP_UNET_DTYPE", "fp32")]
        g = torch.Generator().manual_seed(seed)
        self.p = {}
        it = iter(weights) if weights is not None else None

        def expect(name, arr, shape):
            if tuple(arr.shape) != tuple(shape):
                raise ValueError("checkpoint does not fit the U-Net of pl.py:31-72: %s has shape %s, expected %s"
                                 % (name, tuple(arr.shape), tuple(shape)))
            return np.ascontiguousarray(arr, dtype=np.float32)

        def conv(name, cin, cout, k):
            if it is not None:
                kern, bias = next(it), next(it)  # Keras (kh, kw, in, out)
                kern, bias = expect(name + " kernel", kern, (k, k, cin, cout)), expect(name + " bias", bias, (cout,))
                w = torch.from_numpy(np.ascontiguousarray(kern)).permute(3, 2, 0, 1)
                b = torch.from_numpy(np.ascontiguousarray(bias))
            else:
                std = (2.0 / (cin * k * k)) ** 0.5  # he_normal
                w = torch.randn((cout, cin, k, k), generator=g) * std
                b = torch.zeros(cout)
            self.p[name + ".w"] = w.to(device=device, dtype=self.dtype).contiguous(memory_format=torch.channels_last)
            self.p[name + ".b"] = b.to(device=device, dtype=self.dtype)

        def bn(name, c):
            if it is not None:
                gamma, beta, mean, var = (torch.from_numpy(expect(name + " statistics", next(it), (c,))) for _ in range(4))
            else:
                gamma, beta, mean, var = torch.ones(c), torch.zeros(c), torch.zeros(c), torch.ones(c)
            scale = gamma.double() / torch.sqrt(var.double() + _BN_EPS)
            shift = beta.double() - mean.double() * scale
            self.p[name + ".s"] = scale.to(device=device, dtype=self.dtype).view(1, c, 1, 1)
            self.p[name + ".t"] = shift.to(device=device, dtype=self.dtype).view(1, c, 1, 1)

        def convT(name, cin, cout):
            if it is not None:
                kern, bias = next(it), next(it)  # Keras Conv2DTranspose kernel (kh, kw, out, in)
                kern, bias = expect(name + " kernel", kern, (3, 3, cout, cin)), expect(name + " bias", bias, (cout,))
                w = torch.from_numpy(np.ascontiguousarray(kern)).permute(3, 2, 0, 1)  # torch: (in, out, kh, kw)
                b = torch.from_numpy(np.ascontiguousarray(bias))
            else:
                std = (1.0 / (cin * 9)) ** 0.5
                w = torch.randn((cin, cout, 3, 3), generator=g) * std
                b = torch.zeros(cout)
            self.p[name + ".w"] = w.to(device=device, dtype=self.dtype).contiguous()
            self.p[name + ".b"] = b.to(device=device, dtype=self.dtype)

        def double(name, cin, cout):
            conv(name + ".c1", cin, cout, 3)
            bn(name + ".b1", cout)
            conv(name + ".c2", cout, cout, 3)
            bn(name + ".b2", cout)

        c = in_ch
        for i, f in enumerate(self.filters):
            double("d%d" % i, c, f)
            c = f
        double("mid", c, self.bottleneck)
        c = self.bottleneck
        for i, f in enumerate(reversed(self.filters)):
            convT("u%d.t" % i, c, f)
            double("u%d" % i, 2 * f, f)
            c = f
        conv("head", c, 2, 1)

    def _epilogue(self, x, bias, scale, shift):
        """Conv2D's bias -> ReLU -> BatchNormalization (inference: per-channel scale and shift).  float32 on the GPU: one
        in-place pass in libtissue_hip.so on torch's current stream (same float32 operations in the same order as the
        four torch passes it replaces); otherwise the torch expressions."""
        torch = self.torch
        if (x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and x.shape[0] == 1 and
                x.is_contiguous(memory_format=torch.channels_last) and os.environ.get("TISSUE_HIP_UNET_TORCH_EPILOGUE") != "1"
                and shares_runtime_with_torch(x)):
            lib = _lib.lib()
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _lib.check(lib.tip_bias_relu_affine_f32_dev(_lib.dptr(x.data_ptr()), _lib.dptr(bias.data_ptr()),
                                                        _lib.dptr(scale.data_ptr()), _lib.dptr(shift.data_ptr()),
                                                        ctypes.c_long(x.numel()), int(x.shape[1]), ctypes.c_void_p(stream)))
            return x
        x = torch.nn.functional.relu(x + bias.view(1, -1, 1, 1))
        return x * scale + shift

    def _double(self, x, name):
        F = self.torch.nn.functional
        p = self.p
        for k in ("1", "2"):
            x = F.conv2d(x, p[name + ".c" + k + ".w"], None, padding=1)
            x = self._epilogue(x, p[name + ".c" + k + ".b"], p[name + ".b" + k + ".s"], p[name + ".b" + k + ".t"])
        return x

    def calibrate_head(self, x, fraction):
        """Synthetic-weights helper (bench / tests; no trained weights ship with the reference): shifts the bias of the
        1x1 head so that class 0 exceeds the tail's 0.1 threshold (pl.py:168) on `fraction` of the pixels of input `x`.
        Random-init features follow the input's structure, so the thresholded map then has cell-sized blobs instead of
        being all-or-nothing, and the post-network tail sees a realistic boundary image."""
        torch = self.torch
        z = self.forward(x, logits=True)
        d = (z[0, 0] - z[0, 1]).reshape(-1).float()
        k = min(max(int(round((1.0 - fraction) * d.numel())), 1), d.numel())
        cut = float(torch.kthvalue(d, k).values)
        shift = float(np.log(0.1 / 0.9)) - cut          # p0 > 0.1  <=>  z0 - z1 > ln(1/9)
        self.p["head.b"][0] += shift
        return shift

    def randomize_statistics(self, seed=0):
        """Synthetic-weights helper (tests / bench; no trained weights ship with the reference): gives every convolution a
        non-zero bias and every BatchNormalization non-identity gamma / beta / moving mean / moving variance, drawn from a seeded
        generator on the host (two instances with the same seed get the same values), so that the activations have the dynamic
        range of a trained network instead of the half-zero, unit-scale maps of the he_normal / identity-BatchNorm initialisation."""
        torch = self.torch
        g = torch.Generator().manual_seed(1000 + seed)
        for k in sorted(self.p):
            v = self.p[k]
            if k.endswith(".b") and k != "head.b":
                new = torch.randn(v.numel(), generator=g) * 0.1
            elif k.endswith(".s"):                 # gamma / sqrt(var + eps): gamma in [0.5, 1.5], var in [0.5, 1.5]
                c = v.numel()
                gamma, var = torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) + 0.5
                beta, mean = torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.2
                scale = gamma.double() / torch.sqrt(var.double() + _BN_EPS)
                self.p[k] = scale.to(device=v.device, dtype=v.dtype).view(v.shape)
                t = k[:-2] + ".t"
                self.p[t] = (beta.double() - mean.double() * scale).to(device=v.device, dtype=v.dtype).view(self.p[t].shape)
                continue
            else:
                continue
            self.p[k] = new.to(device=v.device, dtype=v.dtype).view(v.shape)
        for m in list(_MODES):
            if hasattr(self, "_hipw_" + m):
                delattr(self, "_hipw_" + m)

    # -- hand-written convolution path (csrc/tip_unet_conv.h) ---------------------------------------------------------------
    def _split_pack(self, taps, planes, fmt=0):
        """taps: (T, Cin, Cout) float32 on the device -> packed split weights [T][Cin/16][Cout/128][plane][128][16] in bf16 (fmt 0) or
        fp16 (fmt 1: the caller has scaled the taps into fp16's range) pieces.

        Row order inside every group of 32 output channels: row 8 g + 4 h + j (g < 4, h < 2, j < 4) holds channel 16 h + 4 g + j --
        the matrix core's output register i = 4 g + j of half-wave h is then channel 16 h + i, i.e. a lane of the kernel ends up with
        sixteen ADJACENT channels of its pixel (csrc/tip_unet_conv.h, epilogue)."""
        torch = self.torch
        T, cin, cout = taps.shape
        pieces, rest = [], taps.float()
        for _ in range(planes):
            h = rest.to(torch.float16 if fmt else torch.bfloat16)
            pieces.append(h)
            rest = rest - h.float()
        row = torch.arange(32, device=taps.device)
        chan = 16 * ((row >> 2) & 1) + 4 * (row >> 3) + (row & 3)          # channel (within its group of 32) stored in each row
        pk = torch.stack(pieces, 0).view(planes, T, cin // 16, 16, cout // 32, 32)[..., chan]
        pk = pk.reshape(planes, T, cin // 16, 16, cout // 128, 128)
        return pk.permute(1, 2, 4, 0, 5, 3).contiguous()

    def _hip_weights(self, mode):
        """Packed weights and per-channel constants of one arithmetic mode.  fp16 pieces (mode f16x3) carry power-of-two scales:
        activations are stored times A = 2^4, a layer's weights times W = the power of two that puts its largest weight in
        [2^14, 2^15); the kernel multiplies the accumulator by 1 / (A W) before the bias (entry 3 of a layer's tuple), the
        BatchNorm scale / shift (and a bias-only layer's bias and accumulator factor) are multiplied by A, the head's weights by
        1 / A -- exact, so the stored values are A times what the unscaled network computes, bit for bit."""
        key = "_hipw_" + mode
        if getattr(self, key, None) is not None:
            return getattr(self, key)
        torch = self.torch
        p = self.p
        planes, fmt = _MODES[mode][:2]
        act = _F16_ACT_SCALE if fmt else 1.0
        hw = {}

        def pack(taps, bias_only):
            if not fmt:
                return self._split_pack(taps, planes), 1.0
            big = float(taps.abs().max())
            wscale = 2.0 ** (14 - int(np.floor(np.log2(big)))) if big > 0 and np.isfinite(big) else 1.0
            inv = 1.0 / (act * wscale)
            return self._split_pack(taps * wscale, planes, fmt), (inv * act if bias_only else inv)

        def conv3(name):
            w = p[name + ".w"].float()                                  # (cout, cin, 3, 3): cross-correlation, tap (ky, kx) reads (y + ky - 1, x + kx - 1)
            taps = torch.stack([w[:, :, ky, kx].t() for ky in range(3) for kx in range(3)], 0)
            wp, inv = pack(taps, False)
            hw[name] = (wp, [ky - 1 for ky in range(3) for kx in range(3)], [kx - 1 for ky in range(3) for kx in range(3)], inv)

        def conv_t(name):
            # conv_transpose2d(stride 2): out[2 i + k] += in[i] w[k], cropped to the first 2N rows / columns.  Even outputs take
            # k = 0 from i = o / 2 and k = 2 from i = o / 2 - 1, odd outputs k = 1 from i = (o - 1) / 2: four parity classes
            w = p[name + ".w"].float()                                  # (cin, cout, 3, 3)
            per_axis = {0: [(0, 0), (2, -1)], 1: [(1, 0)]}             # parity -> [(k, input offset)]
            for py in (0, 1):
                for px in (0, 1):
                    tl = [(ky, dy, kx, dx) for ky, dy in per_axis[py] for kx, dx in per_axis[px]]
                    taps = torch.stack([w[:, :, ky, kx] for ky, _, kx, _ in tl], 0)
                    wp, inv = pack(taps, True)
                    hw["%s.%d%d" % (name, py, px)] = (wp, [t[1] for t in tl], [t[3] for t in tl], inv)

        def conv_t_fused(name):
            # the same transposed convolution as ONE launch (csrc/tip_unet_conv.h, template TF): the taps are the four input offsets
            # (dy, dx) = (0, 0), (0, -1), (-1, 0), (-1, -1); output channels are VIRTUAL, ordered [group of 32][class py * 2 + px][32]:
            # class (py, px) takes kernel element (ky, kx) = (py ? 1 : (dy ? 2 : 0), px ? 1 : (dx ? 2 : 0)) where its parity has one
            # at that offset (odd parities only at offset 0), zeros elsewhere -- nmask says which classes a tap feeds
            w = p[name + ".w"].float()                                  # (cin, cout, 3, 3)
            cin, cout = int(w.shape[0]), int(w.shape[1])
            offs = [(0, 0), (0, -1), (-1, 0), (-1, -1)]
            taps = torch.zeros((4, cin, cout // 32, 4, 32), dtype=torch.float32, device=w.device)
            masks = []
            for t, (dy, dx) in enumerate(offs):
                m = 0
                for py in (0, 1):
                    for px in (0, 1):
                        if (py == 1 and dy != 0) or (px == 1 and dx != 0):
                            continue
                        ky = 1 if py else (2 if dy else 0)
                        kx = 1 if px else (2 if dx else 0)
                        taps[t, :, :, py * 2 + px, :] = w[:, :, ky, kx].reshape(cin, cout // 32, 32)
                        m |= 1 << (py * 2 + px)
                masks.append(m)
            wp, inv = pack(taps.reshape(4, cin, 4 * cout), True)
            bias = p[name + ".b"].float().reshape(cout // 32, 1, 32).expand(cout // 32, 4, 32).reshape(-1)
            hw[name + ".fused"] = (wp, [o[0] for o in offs], [o[1] for o in offs], inv, masks, (bias * act).contiguous())

        for blk in ("d0", "d1", "d2", "mid", "u0", "u1", "u2"):
            if blk != "d0":
                conv3(blk + ".c1")
            conv3(blk + ".c2")
        for i in range(3):
            conv_t("u%d.t" % i)
            if planes == 2:
                conv_t_fused("u%d.t" % i)
        w0 = p["d0.c1.w"].float()                                       # (128, 2, 3, 3) -> [tap][ci][cout]
        hw["first"] = w0.permute(2, 3, 1, 0).reshape(18, 128).contiguous()
        hw["head"] = (p["head.w"].float().reshape(2, 128) / act).contiguous()
        for k in list(p):
            if k.endswith((".b", ".s", ".t")) and not k.endswith(".t.w"):
                v = p[k].float().reshape(-1)
                # times A: BatchNorm scale (".s") / shift (".t"), and the bias of a bias-only (transposed convolution) layer (".t.b")
                scaled = k.endswith((".s", ".t", ".t.b"))
                hw["f:" + k] = (v * act if scaled else v).contiguous()
        setattr(self, key, hw)
        return hw

    def hip_path_ok(self, x):
        """The hand-written kernels tile every level's grid in 8 x 32 pixels: extents that are multiples of 64 x 256 (what
        prepare_image's padding to powers of two yields for every frame of at least 33 x 129 pixels).  No size limit: a tile
        addresses its halo window, not the tensor."""
        torch = self.torch
        return (_unet_mode() != "miopen" and x.is_cuda and self.dtype == torch.float32 and x.shape[0] == 1 and x.shape[1] == 2
                and self.filters == _FILTERS and self.bottleneck == 1024        # (the reference's widths, pl.py:60-69)
                and x.shape[2] % 64 == 0 and x.shape[3] % 256 == 0 and shares_runtime_with_torch(x))

    def _forward_gated(self, x, logits):
        """One network at a time on a device.  Frames in flight (worker threads, each with its own stream: movie.py, bench.py) would
        otherwise run their forward passes CONCURRENTLY -- the queues share the chip kernel by kernel, every pass takes N times as
        long, all of them end together and the frames' tails (small kernels, host stages) then run together with nothing to
        hide behind: the kernel trace shows the matrix cores idle for 8.5 % of the time (profiles/r04n_*).  Here a pass waits ON THE
        DEVICE (stream.wait_event, no host stall) for the pass queued before it, so that the passes run back to back in ticket order
        and the other frames' tails and projections fill in beside them.  TISSUE_HIP_UNET_SERIAL=0 restores the free-for-all."""
        torch = self.torch
        if os.environ.get("TISSUE_HIP_UNET_SERIAL", "1") == "0":
            return self._forward_hip(x, logits)
        gate = _forward_gate(x.device.index)
        with gate["lock"]:
            s = torch.cuda.current_stream(x.device)
            if gate["event"] is not None and gate["stream"] != s.cuda_stream:
                s.wait_event(gate["event"])
            out = self._forward_hip(x, logits)
            ev = torch.cuda.Event()
            ev.record(s)
            gate["event"], gate["stream"] = ev, s.cuda_stream
        return out

    def _forward_hip(self, x, logits):
        torch = self.torch
        mode = _unet_mode()
        planes, fmt = _MODES[mode][:2]
        hw = self._hip_weights(mode)
        self.last_mode = mode                 # (bench.py / tests: which arithmetic the last forward pass really used)
        lib = _lib.lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        H, W = int(x.shape[2]), int(x.shape[3])
        x = x.to(torch.float32).contiguous()
        D = lambda t: ctypes.c_void_p(t.data_ptr())

        trace = getattr(self, "trace", None)      # tools/unet_layers.py: [(layer, flop, event, event)] per launch
        tconv_mode = os.environ.get("TISSUE_HIP_UNET_TCONV", "split")
        side = _side_streams(x.device) if tconv_mode == "parallel" else None

        def timed(name, flop, fn):
            if trace is None:
                return fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn()
            e1.record()
            trace.append((name, flop, e0, e1))
            return r

        def buf(h, w, c):
            return torch.empty((planes, h, w, c), dtype=torch.float16 if fmt else torch.bfloat16, device=x.device)

        def conv(name, src, skip, h, w, bn, out=None, oh=None, ow=None, sy=1, sx=1, oy=0, ox=0, bias=None, pooled=None, head=None, fused_t=False,
                 on=None):
            wp, dy, dx, inv = hw[name][:4]
            cout = wp.shape[2] * 128
            d = _ConvDesc()
            d.format, d.acc_scale = fmt, inv
            if fused_t:
                d.tf = 1
                for i, m in enumerate(hw[name][4]):
                    d.nmask[i] = m
            d.in0, d.c0 = src.data_ptr(), src.shape[3]
            d.in1, d.c1 = (skip.data_ptr(), skip.shape[3]) if skip is not None else (None, 0)
            d.h, d.w, d.planes = h, w, planes
            d.weights, d.ntaps = wp.data_ptr(), len(dy)
            for i in range(len(dy)):
                d.dy[i], d.dx[i] = dy[i], dx[i]
            d.cout = cout
            if bn is not None:
                d.bias, d.scale, d.shift = hw["f:" + name + ".b"].data_ptr(), hw["f:" + bn + ".s"].data_ptr(), hw["f:" + bn + ".t"].data_ptr()
            elif fused_t:
                d.bias, d.scale, d.shift = hw[name][5].data_ptr(), None, None
            else:
                d.bias, d.scale, d.shift = hw["f:" + bias + ".b"].data_ptr(), None, None
            if head is not None:                  # the network's head in this layer's epilogue: the layer's own output is not stored
                d.head_w, d.head_b, d.head_out = hw["head"].data_ptr(), hw["f:head.b"].data_ptr(), head.data_ptr()
                out, oh, ow = None, h, w
            elif out is None:
                out, oh, ow = buf(h, w, cout), h, w
            d.out, d.out_h, d.out_w, d.sy, d.sx, d.oy, d.ox = (out.data_ptr() if out is not None else None), oh, ow, sy, sx, oy, ox
            d.pool_out = pooled.data_ptr() if pooled is not None else None
            timed("%s %dx%d %d+%d->%d x%d taps" % (name, h, w, d.c0, d.c1, cout // 4 if fused_t else cout, 9 if fused_t else len(dy)),
                  2.0 * h * w * (9 * (cout // 4) if fused_t else len(dy) * cout) * (d.c0 + d.c1),
                  lambda: _lib.check(lib.tip_unet_conv_dev(ctypes.byref(d), stream if on is None else ctypes.c_void_p(on.cuda_stream))))
            return out

        def double(blk, src, skip, h, w, first=False, pooled=None, head=None):
            if first:
                a = buf(h, w, 128)
                timed("first %dx%d 2->128" % (h, w), 2.0 * h * w * 18 * 128,
                      lambda: _lib.check(lib.tip_unet_conv_first_dev(D(x), h, w, D(hw["first"]), D(hw["f:d0.c1.b"]), D(hw["f:d0.b1.s"]),
                                                                     D(hw["f:d0.b1.t"]), D(a), planes, fmt, stream)))
            else:
                a = conv(blk + ".c1", src, skip, h, w, blk + ".b1")
            return conv(blk + ".c2", a, None, h, w, blk + ".b2", pooled=pooled, head=head)

        def pool(t, h, w):
            o = buf(h // 2, w // 2, t.shape[3])
            timed("pool %dx%d x%d" % (h, w, t.shape[3]), 0.0,
                  lambda: _lib.check(lib.tip_unet_pool2_dev(D(t), h, w, int(t.shape[3]), planes, fmt, D(o), stream)))
            return o

        with torch.no_grad():
            skips = []
            h, w = H, W
            cur = None
            for i in range(3):
                nxt = buf(h // 2, w // 2, _FILTERS[i])           # MaxPool2D(2) comes out of the second convolution's epilogue
                f = double("d%d" % i, cur, None, h, w, first=(i == 0), pooled=nxt)
                skips.append(f)
                cur = nxt
                h, w = h // 2, w // 2
            cur = double("mid", cur, None, h, w)
            for i in range(3):
                name = "u%d.t" % i
                cout = hw[name + ".00"][0].shape[2] * 128
                up = buf(2 * h, 2 * w, cout)
                if planes == 2 and tconv_mode == "fused":
                    # ONE launch for the four output parity classes (nine products per staged tile; bit-identical).  Measured SLOWER
                    # than the four launches (6.44 against 6.02 ms over the three layers at 2048^2): a workgroup then covers 32 output
                    # channels and its four steps per chunk carry 24 / 12 / 12 / 6 MFMAs per wave behind a barrier each, where the
                    # four-tap class launch carries 24 behind each -- the per-step barrier cost outweighs the shared staging
                    conv(name + ".fused", cur, None, h, w, None, out=up, oh=2 * h, ow=2 * w, sy=2, sx=2, fused_t=True)
                elif tconv_mode == "parallel":
                    # the four parity classes side by side on four streams (they read one tensor and write disjoint pixels): the idea was
                    # that the one- and two-tap launches (bound by the L2 -> LDS copies of a tile they use once or twice) and the four-tap
                    # launch (bound by the matrix pipe) fill each other's gaps.  Measured SLOWER: forward pass 49.7 / 50.0 ms against
                    # 48.0 / 48.8 ms on one box -- a 512-thread workgroup has its CU to itself whichever launch it belongs to, so nothing
                    # overlaps inside a CU and the fork / join events add their latency.  An experiment switch, like "fused".
                    main = torch.cuda.current_stream(x.device)
                    fork = torch.cuda.Event()
                    fork.record(main)
                    for k, (py, px) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
                        on = main if k == 0 else side[k - 1]
                        if k:
                            on.wait_event(fork)
                        conv("%s.%d%d" % (name, py, px), cur, None, h, w, None, out=up, oh=2 * h, ow=2 * w, sy=2, sx=2, oy=py, ox=px, bias=name, on=on)
                        if k:
                            join = torch.cuda.Event()
                            join.record(on)
                            main.wait_event(join)
                else:
                    for py in (0, 1):
                        for px in (0, 1):
                            conv("%s.%d%d" % (name, py, px), cur, None, h, w, None, out=up, oh=2 * h, ow=2 * w, sy=2, sx=2, oy=py, ox=px, bias=name)
                h, w = 2 * h, 2 * w
                fuse_head = i == 2 and not logits and not os.environ.get("TISSUE_HIP_UNET_SEPARATE_HEAD")
                if fuse_head:                          # softmax probabilities straight out of the last convolution's epilogue
                    out = torch.empty((1, 2, H, W), dtype=torch.float32, device=x.device)
                    double("u%d" % i, up, skips[2 - i], h, w, head=out)
                    return out
                cur = double("u%d" % i, up, skips[2 - i], h, w)
            out = torch.empty((1, 2, H, W), dtype=torch.float32, device=x.device)
            timed("head %dx%d" % (H, W), 2.0 * H * W * 256,
                  lambda: _lib.check(lib.tip_unet_head_dev(D(cur), ctypes.c_long(H * W), D(hw["head"]), D(hw["f:head.b"]), D(out), planes, fmt,
                                                           1 if logits else 0, stream)))
        return out

    def forward(self, x, logits=False):
        """x: (1, C, H, W) tensor on the device -> class probabilities (1, 2, H, W) float32."""
        torch = self.torch
        F = torch.nn.functional
        if self.hip_path_ok(x):
            return self._forward_gated(x, logits)
        self.last_mode = "miopen"
        with torch.no_grad():
            x = x.to(self.dtype).contiguous(memory_format=torch.channels_last)
            skips = []
            for i in range(3):
                f = self._double(x, "d%d" % i)
                skips.append(f)
                x = F.max_pool2d(f, 2)
            x = self._double(x, "mid")
            for i in range(3):
                n_h, n_w = x.shape[2] * 2, x.shape[3] * 2
                x = F.conv_transpose2d(x, self.p["u%d.t.w" % i], self.p["u%d.t.b" % i], stride=2)[:, :, :n_h, :n_w]
                x = torch.cat([x, skips[2 - i]], dim=1)
                x = self._double(x, "u%d" % i)
            x = F.conv2d(x, self.p["head.w"], self.p["head.b"])
            if logits:
                return x.float()
            return torch.softmax(x.float(), dim=1)

    def flops(self, h, w):
        """Dense multiply-add count x2 of one forward pass (for the MFMA roofline)."""
        total = 0
        c, hh, ww = 2, h, w
        for f in self.filters:
            total += 2 * hh * ww * 9 * (c * f + f * f)
            c, hh, ww = f, hh // 2, ww // 2
        total += 2 * hh * ww * 9 * (c * self.bottleneck + self.bottleneck * self.bottleneck)
        c = self.bottleneck
        for f in reversed(self.filters):
            hh, ww = hh * 2, ww * 2
            total += 2 * (hh // 2) * (ww // 2) * 9 * c * f            # transpose conv
            total += 2 * hh * ww * 9 * (2 * f * f + f * f)
            c = f
        total += 2 * hh * ww * c * 2
        return total


def load_keras_weight_list(path):
    """The checkpoint's arrays in model.get_weights() order (layer order of build_unet_model, pl.py:31-72; per layer Conv2D:
    kernel, bias; BatchNormalization: gamma, beta, moving_mean, moving_variance; Conv2DTranspose: kernel, bias).

    `.h5` / `.hdf5` / `.keras`-named HDF5 files -- what model.save_weights() / model.save() write and what pl.py:86
    (`model.load_weights(self.weights_path)`) reads -- go through the self-contained reader of _hdf5.py: like Keras' loading
    by topology, the file's layers that carry weights are taken in file order; the layer kinds are checked against the
    network's (a checkpoint of another architecture raises ValueError, as Keras does).  `.npz`: np.savez(path, *model.get_weights())."""
    if path is None:
        return None
    if not os.path.exists(path):
        raise OSError("Unable to open file (unable to open file: name = '%s')" % path)  # h5py/Keras wording
    if str(path).endswith(".npz"):
        z = np.load(path)
        return [z["arr_%d" % i] for i in range(len(z.files))]
    from . import _hdf5
    layers = [(name, ws) for name, ws in _hdf5.load_keras_weights_h5(path) if ws]
    # build_unet_model's weight-carrying layers: 7 double blocks of (Conv2D, BN, Conv2D, BN), a Conv2DTranspose in front of each
    # of the last three, the 1x1 Conv2D head
    kinds = ["conv", "bn", "conv", "bn"] * 4
    for _ in range(3):
        kinds += ["convT", "conv", "bn", "conv", "bn"]
    kinds += ["conv"]
    if len(layers) != len(kinds):
        raise ValueError("You are trying to load a weight file containing %d layers into a model with %d layers." % (len(layers), len(kinds)))
    out = []
    for (name, ws), kind in zip(layers, kinds):
        arrs = [a for _, a in ws]
        ok = {"conv": len(arrs) == 2 and arrs[0].ndim == 4 and arrs[1].ndim == 1,
              "convT": len(arrs) == 2 and arrs[0].ndim == 4 and arrs[1].ndim == 1 and arrs[0].shape[2] == arrs[1].shape[0],
              "bn": len(arrs) == 4 and all(a.ndim == 1 for a in arrs)}[kind]
        if kind == "conv" and ok:
            ok = arrs[0].shape[3] == arrs[1].shape[0]
        if not ok:
            raise ValueError("checkpoint layer '%s' (%s) does not fit the %s layer of the U-Net of pl.py:31-72 at that position"
                             % (name, ", ".join(str(a.shape) for a in arrs), {"conv": "Conv2D", "convT": "Conv2DTranspose", "bn": "BatchNormalization"}[kind]))
        if kind == "bn":                          # by name where Keras names them (gamma / beta may be absent from odd configurations)
            order = {"gamma": 0, "beta": 1, "moving_mean": 2, "moving_variance": 3}
            keyed = sorted(ws, key=lambda t: order.get(t[0].split("/")[-1].split(":")[0], 99))
            if all(t[0].split("/")[-1].split(":")[0] in order for t in ws):
                arrs = [a for _, a in keyed]
        elif all(t[0].split("/")[-1].split(":")[0] in ("kernel", "bias") for t in ws):
            arrs = [a for _, a in sorted(ws, key=lambda t: 0 if t[0].split("/")[-1].startswith("kernel") else 1)]
        out.extend(arrs)
    return out


class SegmentationPredictor:
    """pl.py:74-198.  `predict(image)` with image (C=2, Y, X) returns (labels int32 (X, Y), HC float64 (X, Y))."""

    def __init__(self, model_weights_path, image_shape, device=None):
        import torch
        self.torch = torch
        self.weights_path = model_weights_path
        if device is None:
            device = int(os.environ.get("TISSUE_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self.device_index = int(device)
        _lib.init(self.device_index)
        if not torch.cuda.is_available():
            raise _lib.TissueHipError("SegmentationPredictor needs an MI355X (no CPU fallback)")
        self.device = torch.device("cuda", self.device_index)
        first_axis_shape, second_axis_shape = find_desired_shape(image_shape[-2], image_shape[-1])
        self.model_shape = (first_axis_shape, second_axis_shape, 2)
        self.model = self.initialize_model()
        self.forward_ms = None   # bench.py sets this to a list: per-call duration of the network's forward pass

    def initialize_model(self):
        weights = load_keras_weight_list(self.weights_path)
        return _UNet(2, self.device, weights=weights)

    # -- U1 -----------------------------------------------------------------------------------------------
    def prepare_image(self, image):
        """pl.py:90-122: per-channel 1/99-percentile normalisation, (C,Y,X)->(1,X,Y,C), front-pad to powers of two.
        Returns a torch tensor laid out (1, C, X', Y') (the NCHW view of the reference's NHWC array) and npad."""
        torch = self.torch
        if isinstance(image, torch.Tensor):
            src_dtype = image.dtype
            t = image.to(device=self.device)
        else:
            arr = np.ascontiguousarray(image)
            if arr.dtype == np.uint16:          # torch has no arithmetic on uint16: the values fit int32
                arr = arr.astype(np.int32)
            elif arr.dtype in (np.uint32, np.uint64):
                arr = arr.astype(np.int64)
            t = torch.as_tensor(arr, device=self.device)
            src_dtype = t.dtype
        if t.dim() != 3:
            raise ValueError("image should be in axes order (C, Y, X)")
        integer_in = not src_dtype.is_floating_point
        single_in = src_dtype == torch.float32
        t = t.to(torch.float64)                 # exact for every integer / float32 input value
        C, Y, X = t.shape
        shape1, shape2 = X, Y
        first_axis_pixels, second_axis_pixels = find_desired_shape(shape1, shape2)
        npad = ((0, 0), (first_axis_pixels - shape1, 0), (second_axis_pixels - shape2, 0), (0, 0))
        # One library submission on torch's stream (tip_unet_prepare_f64_dev): the four order statistics of every channel by
        # radix select, numpy's lerp, clip / scale / transpose / pad in one pass -- nothing comes back to the host.  (The torch
        # expressions below sort every channel and make a dozen elementwise passes: 2.2 ms of a 2048^2 frame against 0.4.)
        dense = t.is_cuda and C <= 8 and ((t.stride(1) == 1 and t.stride(2) == Y) or (t.stride(2) == 1 and t.stride(1) == X))
        if dense and (C == 1 or t.stride(0) >= X * Y) and shares_runtime_with_torch(t) and not os.environ.get("TISSUE_HIP_PREPARE_TORCH"):
            if self.model_shape != (first_axis_pixels, second_axis_pixels, 2):
                self.model_shape = (first_axis_pixels, second_axis_pixels, 2)
            padded = torch.empty((1, C, first_axis_pixels, second_axis_pixels), dtype=torch.float32, device=t.device)
            kind = 2 if integer_in else (1 if single_in else 0)
            _lib.check(_lib.lib().tip_unet_prepare_f64_dev(
                _lib.dptr(t.data_ptr()), C, Y, X, ctypes.c_long(t.stride(0)), ctypes.c_long(t.stride(1)), ctypes.c_long(t.stride(2)), kind,
                _lib.dptr(padded.data_ptr()), second_axis_pixels, first_axis_pixels,
                ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)))
            return padded, npad
        chans = []
        for c in range(C):
            ch = t[c]
            srt = torch.sort(ch.reshape(-1)).values
            per99 = _percentile_linear_t(srt, 99)
            per1 = _percentile_linear_t(srt, 1)
            # normalize_channel (pl.py:21-29) stores the percentiles into a copy of the INPUT: the clip values take its dtype
            hi, lo = per99, per1
            if integer_in:
                hi, lo = float(np.trunc(per99)), float(np.trunc(per1))
            elif single_in:
                hi, lo = float(np.float32(per99)), float(np.float32(per1))
            clipped = torch.where(ch > per99, torch.full_like(ch, hi), ch)
            clipped = torch.where(ch < per1, torch.full_like(ch, lo), clipped)
            if single_in:                        # numpy 1.x keeps float32 array (op) float64 scalar in float32
                n32 = (clipped.to(torch.float32) - float(np.float32(per1))) / float(np.float32(per99 - per1))
                chans.append(n32.to(torch.float64))
            else:
                chans.append((clipped - per1) / (per99 - per1))
        norm = torch.stack(chans)                      # (C, Y, X) float64
        xy = norm.permute(0, 2, 1)                     # np.transpose(normalized) -> (X, Y, C); NCHW view: (C, X, Y)
        if self.model_shape != (first_axis_pixels, second_axis_pixels, 2):
            self.model_shape = (first_axis_pixels, second_axis_pixels, 2)
        padded = torch.zeros((1, C, first_axis_pixels, second_axis_pixels), dtype=torch.float32, device=self.device)
        padded[0, :, npad[1][0]:, npad[2][0]:] = xy.to(torch.float32)
        return padded, npad

    # -- U2-U5 --------------------------------------------------------------------------------------------
    def predict(self, image, debug=False, return_device=False):
        torch = self.torch
        padded, npad = self.prepare_image(image)
        if self.forward_ms is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        prob = self.model.forward(padded)                              # (1, 2, X', Y') float32
        if self.forward_ms is not None:
            ev1.record()
            ev1.synchronize()
            self.forward_ms.append(ev0.elapsed_time(ev1))
        unp = prob[:, :, npad[1][0]:, npad[2][0]:]
        p0 = unp[0, 0]
        labels, hc = self.segment_probability(p0, return_device=return_device)
        return labels, hc

    def segment_probability(self, p0, thr=0.1, return_device=False):
        """pl.py:167-194 on a device-resident probability map (torch tensor (X, Y)): threshold -> 5x5 closing ->
        7x7 erosion -> boundary -> watershed, one submission on the library's stream (tip_unet_tail_dev).  Returns
        (labels int32, HC float64) as numpy arrays, or as torch tensors with return_device=True.

        Stream ordering: the two output tensors come from torch's caching allocator, whose blocks may still be in use by
        kernels queued on torch's current stream (another worker thread's freed temporaries when the threads share the
        default stream), and p0 is produced on that stream -- so the library's stream waits for the torch stream AFTER the
        allocations (tip_wait_stream), and the host waits for the library before torch sees the results."""
        torch = self.torch
        lib = _lib.lib()
        if p0.dim() != 2:
            raise ValueError("segment_probability takes a 2-D probability map")
        if p0.dtype not in (torch.float32, torch.float64):
            p0 = p0.to(torch.float32)
        if p0.stride(1) != 1:
            p0 = p0.contiguous()
        if not shares_runtime_with_torch(p0):
            raise _lib.TissueHipError("libtissue_hip.so and torch use different HIP runtimes in this process: import torch before "
                                      "the first tissue_image_processing_amd call (see INTEGRATION.md)")
        Xn, Yn = int(p0.shape[0]), int(p0.shape[1])
        lab = torch.empty((Xn, Yn), dtype=torch.int32, device=p0.device)
        hc = torch.empty((Xn, Yn), dtype=torch.float64, device=p0.device)
        stream = torch.cuda.current_stream(p0.device).cuda_stream
        _lib.check(lib.tip_wait_stream(ctypes.c_void_p(stream)))
        flags = ctypes.c_int32(0)
        rc = lib.tip_unet_tail_dev(_lib.dptr(p0.data_ptr()), 0 if p0.dtype == torch.float32 else 1, ctypes.c_long(int(p0.stride(0))),
                                   Xn, Yn, ctypes.c_double(thr), _lib.dptr(lab.data_ptr()), _lib.dptr(hc.data_ptr()),
                                   ctypes.byref(flags))
        self.last_flags = flags.value
        self.last_markers = int(lib.tip_last_watershed_labels())
        _lib.check(rc)
        _lib.check(lib.tip_sync())
        if return_device:
            return lab, hc
        return lab.cpu().numpy(), hc.cpu().numpy()
