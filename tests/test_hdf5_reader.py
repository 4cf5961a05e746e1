"""CPU: Keras `.h5` checkpoints (pl.py:76-88: model.load_weights(path); gui.py:38-39 names a `.h5`) through the self-contained
HDF5 reader.  The fixtures were written by the real HDF5 library (h5py of the golden interpreter, tools/make_h5_fixtures.py) in
Keras' documented weight-file layout for the network of pl.py:31-72 at reduced width: one as model.save_weights() writes it
(h5py defaults: superblock 0, symbol-table groups, contiguous datasets), one as a whole-model file with the newest format
features (superblock 3, dense link storage, compressed chunked datasets, weights under `model_weights`)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from tissue_image_processing_amd import _hdf5, prediction_local as pl
from test_unet_host import _np_conv_same, _np_convT_same_s2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = ["keras_tiny_unet_weights.h5", "keras_tiny_unet_model.h5"]


@pytest.fixture(scope="module")
def expected():
    z = np.load(os.path.join(GOLD, "keras_tiny_unet_expected.npz"))
    return z


@pytest.mark.parametrize("fname", FILES)
def test_reader_returns_the_arrays_h5py_wrote(fname, expected):
    layers = _hdf5.load_keras_weights_h5(os.path.join(GOLD, fname))
    assert [n for n, _ in layers] == [str(n) for n in expected["layer_names"]]
    seen = 0
    for lname, ws in layers:
        for wn, arr in ws:
            assert wn.startswith(lname + "/")
            assert arr.dtype == np.float32
            np.testing.assert_array_equal(arr, expected[wn])
            seen += 1
    assert seen == 92                                  # 29 weight-carrying layers: 15 Conv2D + 3 Conv2DTranspose (2 each), 14 BN (4 each)
    f = _hdf5.Hdf5File(os.path.join(GOLD, fname))
    g = f["model_weights"] if "model_weights" in f else f
    assert g.attrs["backend"] == "tensorflow" and g.attrs["keras_version"] == "2.4.0"        # variable-length strings (global heap)
    assert g["max_pooling2d"].attrs["weight_names"].shape == (0,)


@pytest.mark.parametrize("fname", FILES)
def test_checkpoint_order_matches_build_unet_model(fname, expected):
    """load_keras_weight_list -> _UNet consumes the checkpoint in build_unet_model's layer order: the float64 network built from
    the `.h5` equals a numpy restatement of pl.py:31-72 that picks every array BY ITS KERAS NAME (conv2d_5/kernel:0, ...)."""
    ws = pl.load_keras_weight_list(os.path.join(GOLD, fname))
    assert len(ws) == 92
    net = pl._UNet(2, "cpu", dtype=torch.float64, weights=ws)
    assert net.filters == (4, 8, 16) and net.bottleneck == 32
    rng = np.random.default_rng(3)
    x = rng.random((16, 24, 2))
    out = net.forward(torch.from_numpy(x).permute(2, 0, 1)[None]).numpy()[0].transpose(1, 2, 0)

    E = lambda n: expected[n].astype(np.float64)
    conv_i, bn_i, ct_i = [0], [0], [0]

    def nm(kind, counter):
        k = counter[0]
        counter[0] += 1
        return kind if k == 0 else "%s_%d" % (kind, k)

    def double(t):
        for _ in range(2):
            c = nm("conv2d", conv_i)
            t = np.maximum(_np_conv_same(t, E(c + "/kernel:0"), E(c + "/bias:0")), 0)
            b = nm("batch_normalization", bn_i)
            t = (t - E(b + "/moving_mean:0")) / np.sqrt(E(b + "/moving_variance:0") + 1e-3) * E(b + "/gamma:0") + E(b + "/beta:0")
        return t

    def pool(t):
        H, W, C = t.shape
        return t.reshape(H // 2, 2, W // 2, 2, C).max(axis=(1, 3))

    t, skips = x, []
    for _ in range(3):
        f = double(t)
        skips.append(f)
        t = pool(f)
    t = double(t)
    for i in range(3):
        c = nm("conv2d_transpose", ct_i)
        t = _np_convT_same_s2(t, E(c + "/kernel:0"), E(c + "/bias:0"))
        t = np.concatenate([t, skips[2 - i]], axis=-1)
        t = double(t)
    c = nm("conv2d", conv_i)
    logits = t @ E(c + "/kernel:0")[0, 0] + E(c + "/bias:0")
    e = np.exp(logits - logits.max(axis=-1, keepdims=True))
    ref = e / e.sum(axis=-1, keepdims=True)
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-9)


def test_wrong_files_fail_like_keras(tmp_path):
    p = tmp_path / "not_hdf5.h5"
    p.write_bytes(b"this is not an HDF5 file" * 50)
    with pytest.raises(OSError, match="signature"):
        pl.load_keras_weight_list(str(p))
    with pytest.raises(OSError):
        pl.load_keras_weight_list(str(tmp_path / "absent.h5"))
    # a truncated copy of a good file: an error, not garbage weights
    good = open(os.path.join(GOLD, FILES[0]), "rb").read()
    t = tmp_path / "truncated.h5"
    t.write_bytes(good[:len(good) // 3])
    with pytest.raises((OSError, ValueError)):
        pl.load_keras_weight_list(str(t))
    # a checkpoint of another architecture
    ws = pl.load_keras_weight_list(os.path.join(GOLD, FILES[0]))
    bad = list(ws)
    bad[12] = bad[12][..., :3]                         # second block's first kernel with a filter missing
    with pytest.raises(ValueError, match="does not fit"):
        pl._UNet(2, "cpu", dtype=torch.float64, weights=bad)
    with pytest.raises(ValueError, match="92 weight arrays"):
        pl._UNet(2, "cpu", dtype=torch.float64, weights=ws[:-2])
