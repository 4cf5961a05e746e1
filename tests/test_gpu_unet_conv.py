"""GPU: the hand-written U-Net layers (csrc/tip_unet_conv.h: implicit-GEMM convolutions on the bf16 matrix cores with split
float32 operands) against float64 references of the same layers.  The network's trained-weight parity is unpinned (no
TensorFlow, no weights ship with the reference: pl.py:31-72 is restated in prediction_local._UNet); what is pinned here is
that the hand-written kernels compute the SAME network as the torch expressions, to the error bound the split arithmetic
states: per term <= 1.6e-5 relative (bf16x3: two pieces, three products) resp. 9e-8 (bf16x6)."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def arith(monkeypatch):
    def set_mode(m):
        monkeypatch.setenv("TISSUE_HIP_UNET_ARITH", m)
    return set_mode


def _split(t, planes):
    import torch
    pieces, rest = [], t.float()
    for _ in range(planes):
        h = rest.to(torch.bfloat16)
        pieces.append(h)
        rest = rest - h.float()
    return torch.stack(pieces, 0).contiguous()


def _join(planes_t):
    return planes_t.float().sum(0)


@pytest.mark.parametrize("planes", [2, 3])
def test_single_layers_against_float64(planes, arith):
    """One 3x3 convolution with two concatenated inputs, one transposed convolution, the pooling and the head, each against
    torch float64 on the host, with asymmetric random data (a transposed or mirrored tap / channel order cannot pass)."""
    import torch
    from tissue_image_processing_amd import prediction_local as pl, _lib
    arith("bf16x3" if planes == 2 else "bf16x6")
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    net = pl._UNet(2, dev, dtype=torch.float32, seed=5)
    lib = _lib.lib()
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    H, W, C0, C1, CO = 16, 64, 32, 16, 128
    a0 = torch.randn((H, W, C0), generator=g)
    a1 = torch.randn((H, W, C1), generator=g)
    wt = torch.randn((CO, C0 + C1, 3, 3), generator=g) * 0.1
    bias, scale, shift = torch.randn(CO, generator=g), torch.rand(CO, generator=g) + 0.5, torch.randn(CO, generator=g)
    taps = torch.stack([wt[:, :, ky, kx].t() for ky in range(3) for kx in range(3)], 0).to(dev)
    wp = net._split_pack(taps, planes)
    p0, p1 = _split(a0, planes).to(dev), _split(a1, planes).to(dev)
    out = torch.empty((planes, H, W, CO), dtype=torch.bfloat16, device=dev)
    d = pl._ConvDesc()
    d.in0, d.c0, d.in1, d.c1, d.h, d.w, d.planes = p0.data_ptr(), C0, p1.data_ptr(), C1, H, W, planes
    d.weights, d.ntaps, d.cout = wp.data_ptr(), 9, CO
    for i in range(9):
        d.dy[i], d.dx[i] = i // 3 - 1, i % 3 - 1
    fb, fs, ft = bias.to(dev), scale.to(dev), shift.to(dev)
    d.bias, d.scale, d.shift = fb.data_ptr(), fs.data_ptr(), ft.data_ptr()
    d.out, d.out_h, d.out_w, d.sy, d.sx, d.oy, d.ox = out.data_ptr(), H, W, 1, 1, 0, 0
    fused_pool = torch.zeros((planes, H // 2, W // 2, CO), dtype=torch.bfloat16, device=dev)
    d.pool_out = fused_pool.data_ptr()          # MaxPool2D(2) out of the same epilogue
    _lib.check(lib.tip_unet_conv_dev(ctypes.byref(d), stream))
    torch.cuda.synchronize()
    got = _join(out.cpu()).double()
    assert torch.equal(_join(fused_pool.cpu()), torch.nn.functional.max_pool2d(_join(out.cpu()).permute(2, 0, 1)[None], 2)[0].permute(1, 2, 0))
    # reference on the values the kernel was given (the split inputs / weights), in float64
    x64 = torch.cat([_join(p0.cpu()), _join(p1.cpu())], 2).double().permute(2, 0, 1)[None]
    w64 = _join(_split(wt, planes)).double()
    ref = torch.nn.functional.conv2d(x64, w64, None, padding=1)[0].permute(1, 2, 0)
    ref = torch.relu(ref + bias.double()) * scale.double() + shift.double()
    err = float((got - ref).abs().max() / ref.abs().max())
    print("conv3x3 (%d pieces): max error / max |value| = %.2e" % (planes, err))
    assert err < (3e-5 if planes == 2 else 2e-6)

    # transposed convolution 3x3 stride 2 'same' (= conv_transpose2d cropped to 2N), bias only
    CI, CO2 = 32, 128
    a = torch.randn((H, W, CI), generator=g)
    wtt = torch.randn((CI, CO2, 3, 3), generator=g) * 0.1
    bt = torch.randn(CO2, generator=g)
    pa = _split(a, planes).to(dev)
    up = torch.zeros((planes, 2 * H, 2 * W, CO2), dtype=torch.bfloat16, device=dev)
    fbt = bt.to(dev)
    per_axis = {0: [(0, 0), (2, -1)], 1: [(1, 0)]}
    keep = []
    for py in (0, 1):
        for px in (0, 1):
            tl = [(ky, dy, kx, dx) for ky, dy in per_axis[py] for kx, dx in per_axis[px]]
            wpk = net._split_pack(torch.stack([wtt[:, :, ky, kx] for ky, _, kx, _ in tl], 0).to(dev), planes)
            keep.append(wpk)
            d = pl._ConvDesc()
            d.in0, d.c0, d.in1, d.c1, d.h, d.w, d.planes = pa.data_ptr(), CI, None, 0, H, W, planes
            d.weights, d.ntaps, d.cout = wpk.data_ptr(), len(tl), CO2
            for i, t in enumerate(tl):
                d.dy[i], d.dx[i] = t[1], t[3]
            d.bias, d.scale, d.shift = fbt.data_ptr(), None, None
            d.out, d.out_h, d.out_w, d.sy, d.sx, d.oy, d.ox = up.data_ptr(), 2 * H, 2 * W, 2, 2, py, px
            _lib.check(lib.tip_unet_conv_dev(ctypes.byref(d), stream))
    torch.cuda.synchronize()
    got = _join(up.cpu()).double()
    x64 = _join(pa.cpu()).double().permute(2, 0, 1)[None]
    ref = torch.nn.functional.conv_transpose2d(x64, _join(_split(wtt, planes)).double(), bt.double(), stride=2)[0, :, :2 * H, :2 * W].permute(1, 2, 0)
    err = float((got - ref).abs().max() / ref.abs().max())
    print("conv-transpose (%d pieces): max error / max |value| = %.2e" % (planes, err))
    assert err < (3e-5 if planes == 2 else 2e-6)

    # MaxPool2D(2): exact on the split values
    pooled = torch.empty((planes, H // 2, W // 2, CO), dtype=torch.bfloat16, device=dev)
    _lib.check(lib.tip_unet_pool2_dev(ctypes.c_void_p(out.data_ptr()), H, W, CO, planes, ctypes.c_void_p(pooled.data_ptr()), stream))
    torch.cuda.synchronize()
    want = torch.nn.functional.max_pool2d(_join(out.cpu()).permute(2, 0, 1)[None], 2)[0].permute(1, 2, 0)
    assert torch.equal(_join(pooled.cpu()), want)


@pytest.mark.parametrize("mode,tol", [("bf16x3", 2e-4), ("bf16x6", 2e-5)])
def test_network_hip_path_vs_float64(mode, tol, arith):
    """The whole network through the hand-written kernels (extents that are multiples of 64 x 256 take that path) against the
    float64 torch network on the host: class probabilities to `tol` absolute; the MIOpen float32 path of the same network
    is held to 1e-4 by test_gpu_unet.py."""
    import torch
    from tissue_image_processing_amd import prediction_local as pl
    arith(mode)
    gpu = pl._UNet(2, torch.device("cuda", 0), dtype=torch.float32, seed=3)
    ref = pl._UNet(2, "cpu", dtype=torch.float64, seed=3)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.random((1, 2, 64, 256)))
    xg = x.to("cuda").float()
    assert gpu.hip_path_ok(xg)
    out = gpu.forward(xg).cpu().double()
    exp = ref.forward(x)
    err = float((out - exp).abs().max())
    z = gpu.forward(xg, logits=True).cpu().double()
    ze = ref.forward(x, logits=True)
    zerr = float((z - ze).abs().max() / ze.abs().max())
    print("%s network 64x256: max |dp| = %.2e, max logit error / max |logit| = %.2e" % (mode, err, zerr))
    assert err < tol
    arith("miopen")
    assert not gpu.hip_path_ok(xg)


def test_hip_and_miopen_paths_segment_alike(arith):
    """512 x 512 frame through predict() with both convolution paths: the class maps differ in a handful of pixels that sit
    within rounding of the 0.1 threshold (any two float32 convolution orders do), the segmentations agree (IoU)."""
    import torch
    from tissue_image_processing_amd import prediction_local as pl, synthetic
    N = 512
    sites = synthetic.make_sites(N, N, seed=8)[0]
    d1, d2, i1 = synthetic._two_nearest(sites, N, N)
    rng = np.random.default_rng(8)
    zo = 3000 * np.exp(-(d2 - d1) ** 2 / 4) + rng.poisson(100, (N, N))
    atoh = 1500 * (i1 % 3 == 0) + rng.poisson(100, (N, N))
    img = np.stack([atoh, zo]).astype(np.float64)
    pred = pl.SegmentationPredictor(None, img.shape)
    padded, _ = pred.prepare_image(img)
    arith("miopen")
    pred.model.calibrate_head(padded, 0.5)
    p_m = pred.model.forward(padded)[0, 0]
    arith("bf16x3")
    p_h = pred.model.forward(padded)[0, 0]
    dmax = float((p_m - p_h).abs().max())
    flips = int(((p_m > 0.1) != (p_h > 0.1)).sum())
    print("512^2: max |dp0| between the MIOpen and the bf16x3 paths %.2e, thresholded pixels that differ: %d of %d" % (dmax, flips, N * N))
    assert dmax < 5e-4 and flips < N * N * 1e-3


@pytest.mark.parametrize("shape", [(128, 512), (192, 256), (64, 768)])
def test_network_hip_path_other_extents(shape, arith, monkeypatch):
    """Extents that mix the kernel's tile flavours over the levels: 16-row tiles with the four-step weight schedule, 16-row
    tiles with the two-chunk activation schedule (one- and two-tap classes), 8-row tiles where a level's grid is not a multiple
    of 16 rows (192 -> 24 rows at the bottleneck), non-square frames; and the 8-row flavour forced everywhere."""
    import torch
    from tissue_image_processing_amd import prediction_local as pl, _lib
    arith("bf16x3")
    gpu = pl._UNet(2, torch.device("cuda", 0), dtype=torch.float32, seed=7)
    ref = pl._UNet(2, "cpu", dtype=torch.float64, seed=7)
    rng = np.random.default_rng(shape[0])
    x = torch.from_numpy(rng.random((1, 2) + shape))
    xg = x.to("cuda").float()
    assert gpu.hip_path_ok(xg)
    exp = ref.forward(x)
    err = float((gpu.forward(xg).cpu().double() - exp).abs().max())
    with _lib.tuning(TIP_UNET_TILE8="1"):
        err8 = float((gpu.forward(xg).cpu().double() - exp).abs().max())
    with _lib.tuning(TIP_UNET_TILE8="0"):
        err16 = float((gpu.forward(xg).cpu().double() - exp).abs().max())
    # the head as its own kernel on the stored split planes (the default fuses it into the last convolution's epilogue)
    fused = gpu.forward(xg)
    monkeypatch.setenv("TISSUE_HIP_UNET_SEPARATE_HEAD", "1")
    sep = gpu.forward(xg)
    errsep = float((sep.cpu().double() - exp).abs().max())
    dhead = float((sep - fused).abs().max())
    monkeypatch.delenv("TISSUE_HIP_UNET_SEPARATE_HEAD")
    # steps per barrier of the 3x3 16-row kernel (default three) and the workgroup order change the schedule, not the arithmetic
    for knob, val in (("TIP_UNET_SPB", "1"), ("TIP_UNET_SPB", "2"), ("TIP_UNET_XCD_MAP", "0")):
        with _lib.tuning(**{knob: val}):
            assert torch.equal(gpu.forward(xg), fused), (knob, val)
    print("%dx%d: max |dp| %.2e (8-row tiles everywhere: %.2e, 16-row wherever possible: %.2e, separate head: %.2e, fused vs separate head %.2e)"
          % (shape[0], shape[1], err, err8, err16, errsep, dhead))
    assert err < 2e-4 and err8 < 2e-4 and err16 < 2e-4 and errsep < 2e-4 and dhead < 5e-5
