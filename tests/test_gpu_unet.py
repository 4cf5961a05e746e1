"""GPU: SegmentationPredictor (pl.py:74-198 mirror): tail parity vs golden, network vs float64 CPU torch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tail_vs_golden(golden):
    import torch
    from tissue_image_processing_amd import prediction_local as pl
    g = golden("unet_tail")
    pred = pl.SegmentationPredictor(None, g["p0"].shape)
    p0 = torch.as_tensor(g["p0"], device=pred.device)
    labels, hc = pred.segment_probability(p0, thr=0.55)
    np.testing.assert_array_equal(hc, g["hc"])            # rank-filter chain is bit exact
    assert labels.dtype == np.int32
    np.testing.assert_array_equal(labels, g["labels"])   # the watershed on the binary boundary image, bit exact


def test_network_gpu_vs_cpu_float64():
    import torch
    from tissue_image_processing_amd import prediction_local as pl
    gpu = pl._UNet(2, torch.device("cuda", 0), dtype=torch.float32, seed=3)
    ref = pl._UNet(2, "cpu", dtype=torch.float64, seed=3)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.random((1, 2, 64, 96)))
    out = gpu.forward(x.to("cuda").float()).cpu().double()
    exp = ref.forward(x)
    # fp32 network, tolerance 1e-4 absolute on softmax probabilities
    assert float((out - exp).abs().max()) < 1e-4


def test_predictor_loads_a_keras_h5_checkpoint():
    """SegmentationPredictor(UNET_WEIGHTS_PATH, shape) as gui.py:2062 calls it, with a Keras-layout `.h5` (fixture written by
    the real HDF5 library, reduced widths): the weights arrive in build_unet_model's order -- the device network equals the
    float64 host network built from the same file -- and predict() runs end to end."""
    import os
    import torch
    from tissue_image_processing_amd import prediction_local as pl
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keras_tiny_unet_weights.h5")
    rng = np.random.default_rng(2)
    img = rng.random((2, 100, 70)) * 1000
    pred = pl.SegmentationPredictor(path, img.shape)
    assert pred.model.filters == (4, 8, 16) and pred.model.bottleneck == 32
    ref = pl._UNet(2, "cpu", dtype=torch.float64, weights=pl.load_keras_weight_list(path))
    padded, _ = pred.prepare_image(img)
    out = pred.model.forward(padded).cpu().double()
    exp = ref.forward(padded.cpu().double())
    assert float((out - exp).abs().max()) < 1e-4
    labels, hc = pred.predict(img)
    assert labels.shape == (70, 100) and labels.dtype == np.int32 and hc.shape == (70, 100)


def test_predict_shapes_and_padding():
    from tissue_image_processing_amd import prediction_local as pl, synthetic
    rng = np.random.default_rng(1)
    img = rng.random((2, 100, 70)) * 1000        # (C, Y, X) -> network runs on (X', Y') = (128, 128)
    pred = pl.SegmentationPredictor(None, img.shape)
    assert pred.model_shape == (128, 128, 2)
    labels, hc = pred.predict(img)
    assert labels.shape == (70, 100) and hc.shape == (70, 100)   # (X, Y) like the reference
    assert labels.dtype == np.int32 and hc.dtype == np.float64
    padded, npad = pred.prepare_image(img)
    assert tuple(padded.shape) == (1, 2, 128, 128) and npad[1][0] == 58 and npad[2][0] == 28
    ref = np.stack([pl.normalize_channel(img[c]) for c in range(2)])
    got = padded[0, :, 58:, 28:].cpu().numpy()
    np.testing.assert_allclose(got, np.transpose(ref, (0, 2, 1)).astype(np.float32), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("case", ["f64", "f64_transposed_view", "f32", "u16", "ties", "one_pixel_rows"])
def test_prepare_image_fused_pass_equals_torch_expressions(case, monkeypatch):
    """U1 as one library submission (tip_unet_prepare_f64_dev: radix-select order statistics, numpy's lerp, clip / scale /
    transpose / pad) against the torch expressions it replaces (sort + where + divide), bit for bit, for every input dtype rule
    of normalize_channel (pl.py:21-29), both plane orientations, ragged extents and heavy value ties."""
    import torch
    from tissue_image_processing_amd import prediction_local as pl
    rng = np.random.default_rng(11)
    if case == "f64":
        img = rng.random((2, 301, 423)) * 4000.0 - 100.0
    elif case == "f64_transposed_view":
        base = torch.as_tensor(rng.random((2, 260, 517)) * 900.0, device="cuda")
        img = base.transpose(1, 2)                           # (C, 517, 260) view, first plane index with unit stride
    elif case == "f32":
        img = (rng.random((2, 129, 65)) * 70000.0).astype(np.float32)
    elif case == "u16":
        img = rng.integers(0, 4000, (2, 200, 333)).astype(np.uint16)
    elif case == "ties":
        img = rng.integers(0, 5, (3, 97, 131)).astype(np.float64)
    else:
        img = rng.random((1, 1, 700))
    pred = pl.SegmentationPredictor(None, tuple(img.shape))
    fused, npad = pred.prepare_image(img)
    monkeypatch.setenv("TISSUE_HIP_PREPARE_TORCH", "1")
    ref, npad_ref = pred.prepare_image(img)
    assert npad == npad_ref and fused.shape == ref.shape and fused.dtype == ref.dtype
    torch.cuda.synchronize()
    if case == "f32":
        # float32 arithmetic: numpy divides (pl.py:29), the fused pass divides, torch's tensor / scalar multiplies by the
        # reciprocal -- one ulp apart on some pixels; the clip decisions and the pad are identical
        np.testing.assert_allclose(fused.cpu().numpy(), ref.cpu().numpy(), rtol=2.5e-7, atol=1e-7)
        host = np.stack([pl.normalize_channel(img[c]) for c in range(img.shape[0])])       # numpy's own division
        got = fused[0, :, npad[1][0]:, npad[2][0]:].cpu().numpy()
        np.testing.assert_allclose(got, np.transpose(host, (0, 2, 1)), rtol=2.5e-7, atol=1e-7)
    else:
        assert torch.equal(fused, ref)


@pytest.mark.parametrize("shape", [(300, 421), (64, 64), (65, 129), (7, 500), (1, 40), (3, 3), (1024, 1024)])
def test_tail_morphology_in_one_kernel_equals_the_rank_filter_chain(shape):
    """pl.py:168-193 (threshold, 5x5 closing, 7x7 erosion, boundary dilation) as one byte-image kernel against the same chain as
    separate float64 rank-filter launches (themselves pinned by the golden `unet_tail`): HC maps and labels bit for bit, on extents
    that are smaller than the 9-pixel halo, not multiples of the 64-pixel tile, and on blobs that touch the borders."""
    import torch
    from tissue_image_processing_amd import prediction_local as pl, _lib
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    p0 = 0.5 + 0.5 * np.sin(yy / 3.7) * np.cos(xx / 5.3) + 0.15 * rng.standard_normal(shape)
    pred = pl.SegmentationPredictor(None, (2,) + shape)
    t = torch.as_tensor(p0.astype(np.float32), device=pred.device)
    lab, hc = pred.segment_probability(t, thr=0.5)
    with _lib.tuning(TIP_UNET_TAIL_UNFUSED="1"):
        lab_u, hc_u = pred.segment_probability(t, thr=0.5)
    np.testing.assert_array_equal(hc, hc_u)
    np.testing.assert_array_equal(lab, lab_u)
    lab64, hc64 = pred.segment_probability(t.double(), thr=0.5)       # float64 probability maps take the same kernel
    np.testing.assert_array_equal(hc64, hc)
    np.testing.assert_array_equal(lab64, lab)


class _FakeNet(object):
    """Stands in for the trained network (no weights ship with the reference): returns a fixed class-probability map."""

    def __init__(self, prob_nhwc, device):
        import torch
        self.prob = torch.as_tensor(np.ascontiguousarray(np.transpose(prob_nhwc, (0, 3, 1, 2))), device=device)

    def forward(self, x):
        assert tuple(x.shape) == tuple(self.prob.shape)
        return self.prob


def test_prepare_image_matches_reference_golden(golden):
    """U1 pinned by the reference's own prepare_image / normalize_channel / find_desired_shape (tools/make_goldens.py
    gold_unet_predict): the reference hands float64 NHWC to Keras, which casts to float32; ours is that float32 NCHW."""
    from tissue_image_processing_amd import prediction_local as pl
    g = golden("unet_predict")
    for (a, b), want in zip(g["fds_in"], g["fds_out"]):
        assert tuple(pl.find_desired_shape(int(a), int(b))) == tuple(want)
    for key, img in (("norm_c0", g["image"][0]), ("norm_c1", g["image"][1]), ("norm_u16_c0", g["image_u16"][0])):
        np.testing.assert_array_equal(pl.normalize_channel(img), g[key])
    pred = pl.SegmentationPredictor(None, g["image"].shape)
    padded, npad = pred.prepare_image(g["image"])
    np.testing.assert_array_equal(np.array(npad), g["npad"])
    np.testing.assert_array_equal(padded.cpu().numpy(), np.transpose(g["padded"], (0, 3, 1, 2)).astype(np.float32))
    pred2 = pl.SegmentationPredictor(None, (2, 64, 64))
    padded2, npad2 = pred2.prepare_image(g["image2"])
    np.testing.assert_array_equal(np.array(npad2), g["npad2"])
    assert tuple(pred2.model_shape) == tuple(g["model_shape2"])
    np.testing.assert_array_equal(padded2.cpu().numpy(), np.transpose(g["padded2"], (0, 3, 1, 2)).astype(np.float32))
    # uint16 planes (what gui.py:2059-2061 passes): the clip values are truncated to the input dtype (pl.py:25-26)
    pu, npu = pred.prepare_image(g["image_u16"])
    got = pu[0, 0, npu[1][0]:, npu[2][0]:].cpu().numpy()
    np.testing.assert_array_equal(got, g["norm_u16_c0"].T.astype(np.float32))


def test_predict_matches_reference_golden_with_fixed_network_output(golden):
    """The reference's predict() (pl.py:124-199) run with a stand-in network that returns a fixed probability map:
    int32 labels and the HC map, bit for bit."""
    from tissue_image_processing_amd import prediction_local as pl
    g = golden("unet_predict")
    pred = pl.SegmentationPredictor(None, g["image"].shape)
    pred.model = _FakeNet(g["prob"], pred.device)
    labels, hc = pred.predict(g["image"])
    assert labels.dtype == np.int32 and hc.dtype == np.float64
    np.testing.assert_array_equal(hc, g["hc"])
    np.testing.assert_array_equal(labels, g["labels"])


def test_unet_leg_at_headline_size():
    """BASELINE config 3 at size: 2048 x 2048 planes through prepare_image -> the 3-level U-Net (random-init, head bias
    calibrated so that the class map has structure) -> threshold / closing / erosion / boundary / watershed.  The tail
    (rank filters and the two-valued watershed with ~3 M equal-keyed markers) is compared bit for bit with the oracle
    on the network's own output."""
    import torch
    from oracle import oracle as orc
    from tissue_image_processing_amd import prediction_local as pl, synthetic
    N = 2048
    sites = synthetic.make_sites(N, N, seed=6)[0]
    d1, d2, i1 = synthetic._two_nearest(sites, N, N)
    rng = np.random.default_rng(6)
    zo = 3000 * np.exp(-(d2 - d1) ** 2 / 4) + rng.poisson(100, (N, N))
    atoh = 1500 * (i1 % 3 == 0) + rng.poisson(100, (N, N))
    img = np.stack([atoh, zo]).astype(np.float64)
    pred = pl.SegmentationPredictor(None, img.shape)
    assert pred.model_shape == (N, N, 2)
    padded, npad = pred.prepare_image(img)
    assert tuple(padded.shape) == (1, 2, N, N) and npad[1][0] == 0 and npad[2][0] == 0
    pred.model.calibrate_head(padded, 0.5)
    prob = pred.model.forward(padded)
    assert tuple(prob.shape) == (1, 2, N, N) and prob.dtype == torch.float32
    p0 = prob[0, 0]
    frac = float((p0 > 0.1).float().mean())
    assert 0.3 < frac < 0.7
    labels, hc = pred.segment_probability(p0)
    assert labels.shape == (N, N) and labels.dtype == np.int32 and hc.dtype == np.float64
    p0h = p0.cpu().numpy()
    closed = orc.erosion(orc.dilation(255.0 * (p0h > 0.1), 5), 5)
    hc_ref = orc.erosion(closed, 7)
    np.testing.assert_array_equal(hc, hc_ref)
    boundary = orc.dilation(closed - hc_ref, 5)
    ref = orc.watershed(boundary)
    mism = int((labels != ref).sum())
    print("unet leg 2048^2: %d labels, %.0f%% foreground, %.0f%% markers, mismatches %d" % (
        ref.max(), 100 * frac, 100 * float((boundary == 0).mean()), mism))
    assert mism == 0


def test_fused_conv_epilogue_equals_torch_ops(monkeypatch):
    """bias -> ReLU -> BN scale/shift in one in-place HIP pass (tip_bias_relu_affine_f32_dev, torch's current stream)
    is bit-identical to the torch expressions it replaces; the whole network agrees with either epilogue to float32
    rounding (MIOpen's convolution kernels are not bit-reproducible from call to call, so no exact equality there)."""
    import torch
    from tissue_image_processing_amd import prediction_local as pl
    net = pl._UNet(2, torch.device("cuda", 0), dtype=torch.float32, seed=5)
    g = torch.Generator().manual_seed(1)
    for name in list(net.p):                      # non-trivial BN statistics and biases
        if name.endswith(".s") or name.endswith(".t") or name.endswith(".b"):
            net.p[name] = (net.p[name] + 0.3 * torch.randn(net.p[name].shape, generator=g).to(net.p[name])).contiguous()
    x = torch.randn((1, 128, 37, 52), generator=g).to("cuda").contiguous(memory_format=torch.channels_last)
    b, s, t = net.p["d0.c2.b"], net.p["d0.b2.s"], net.p["d0.b2.t"]
    want = torch.nn.functional.relu(x + b.view(1, -1, 1, 1)) * s + t
    got = net._epilogue(x.clone(memory_format=torch.channels_last), b, s, t)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    inp = torch.rand((1, 2, 64, 96), generator=g).to("cuda")
    fused = net.forward(inp)
    monkeypatch.setenv("TISSUE_HIP_UNET_TORCH_EPILOGUE", "1")
    plain = net.forward(inp)
    assert float((fused - plain).abs().max()) < 1e-5


def test_unet_leg_three_frames_in_flight_every_step_exact():
    """Three worker threads, each with its own library stream and its own torch stream, run the U-Net leg on 2048^2 frames
    at the same time (bench.py's arrangement).  EVERY step's watershed must be the two-valued mode and its labels and HC map
    must equal the oracle's on that step's own class map: the outputs of the tail are torch tensors written by the library's
    stream, so a missing ordering edge between the two stream domains shows up here as a corrupted boundary image (round 2
    recorded such a run: profiles/r02e_unet_kernel_stats.csv)."""
    import threading
    import torch
    from oracle import oracle as orc
    from tissue_image_processing_amd import prediction_local as pl, synthetic, _lib
    N, STEPS, THREADS = 2048, 2, 3
    results, errors, base = {}, [], {}

    def make_image(seed):
        sites = synthetic.make_sites(N, N, seed=seed)[0]
        d1, d2, i1 = synthetic._two_nearest(sites, N, N)
        rng = np.random.default_rng(seed)
        zo = 3000 * np.exp(-(d2 - d1) ** 2 / 4) + rng.poisson(100, (N, N))
        atoh = 1500 * (i1 % 3 == 0) + rng.poisson(100, (N, N))
        return np.stack([atoh, zo]).astype(np.float64)

    images = [make_image(20 + k) for k in range(THREADS)]
    start = threading.Barrier(THREADS)

    def work(k):
        try:
            _lib.init(0)
            with torch.cuda.stream(torch.cuda.Stream(device=0)):
                pred = pl.SegmentationPredictor(None, images[k].shape)
                padded, _ = pred.prepare_image(images[k])
                pred.model.calibrate_head(padded, 0.5)
                base[k] = pred.model.forward(padded)[0, 0].cpu().numpy()     # (before the threads are released together)
                start.wait()
                for step in range(STEPS):
                    padded, npad = pred.prepare_image(images[k])
                    p0 = pred.model.forward(padded)[0, 0]
                    junk = [torch.empty((N, N), dtype=torch.float64, device=p0.device).normal_() for _ in range(3)]   # churn the allocator
                    del junk
                    lab, hc = pred.segment_probability(p0, return_device=True)
                    results[(k, step)] = (p0.cpu().numpy(), lab.cpu().numpy(), hc.cpu().numpy(), pred.last_flags)
        except BaseException as e:
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(THREADS)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == THREADS * STEPS
    for (k, step), (p0, lab, hc, flags) in sorted(results.items()):
        assert flags & _lib.WS_FLAG_TWO_VALUED and not (flags & (_lib.WS_FLAG_SERIAL_EXACT | _lib.WS_FLAG_SERIAL_FINISH)), (k, step, flags)
        # the network passes of the three threads are ordered on the device (prediction_local._forward_gated): same input, same bits,
        # whatever ran beside the pass
        np.testing.assert_array_equal(p0, base[k])
        closed = orc.erosion(orc.dilation(255.0 * (p0 > np.float32(0.1)), 5), 5)
        hc_ref = orc.erosion(closed, 7)
        np.testing.assert_array_equal(hc, hc_ref)
        ref = orc.watershed(orc.dilation(closed - hc_ref, 5))
        mism = int((lab != ref).sum())
        print("thread %d step %d: %d labels, flags %#x, mismatches %d" % (k, step, ref.max(), flags, mism))
        assert mism == 0
