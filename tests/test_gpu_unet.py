"""GPU: SegmentationPredictor (pl.py:74-198 mirror): tail parity vs golden, network vs float64 CPU torch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def label_iou(test, ref):
    ious = []
    for l in np.unique(ref):
        if l == 0:
            continue
        m = ref == l
        cand = np.bincount(test[m])
        cand[0] = 0
        if cand.sum() == 0:
            ious.append(0.0)
            continue
        k = cand.argmax()
        ious.append((m & (test == k)).sum() / float((m | (test == k)).sum()))
    return float(np.mean(ious))


def test_tail_vs_golden(golden):
    import torch
    from tissue_image_processing_amd import prediction_local as pl
    g = golden("unet_tail")
    pred = pl.SegmentationPredictor(None, g["p0"].shape)
    p0 = torch.as_tensor(g["p0"], device=pred.device)
    labels, hc = pred.segment_probability(p0, thr=0.55)
    np.testing.assert_array_equal(hc, g["hc"])            # rank-filter chain is bit exact
    assert labels.dtype == np.int32 and labels.max() == g["labels"].max()
    iou = label_iou(labels, g["labels"])
    print("unet tail: label IoU vs reference %.4f, mismatching pixels %.2f%%" % (iou, 100 * float((labels != g["labels"]).mean())))
    assert iou > 0.9
    zeros = g["boundary"] == 0
    np.testing.assert_array_equal(labels[zeros], g["labels"][zeros])   # marker components identical


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
