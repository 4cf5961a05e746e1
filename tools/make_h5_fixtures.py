#!/opt/conda/bin/python3.9
"""Keras-layout `.h5` weight files written by the REAL HDF5 library (h5py of the container's Anaconda tree; the default
interpreter has none), as fixtures for tissue_image_processing_amd/_hdf5.py, plus a self-check of that reader against h5py on
randomly structured files (both `libver` settings: old-style symbol-table groups and new-style compact / dense groups).

    /opt/conda/bin/python3.9 tools/make_h5_fixtures.py

TensorFlow / Keras are absent from this container, so the files are laid out as keras/saving/hdf5_format.py documents
(`save_weights_to_hdf5_group`: root attributes `layer_names`, `backend`, `keras_version`; one group per layer with attribute
`weight_names` and one dataset per weight at `<layer>/<layer>/<kernel:0 | bias:0 | gamma:0 | ...>`), for the network of
pl.py:31-72 with Keras' automatic layer names -- at reduced width (4 / 8 / 16 / 32 filters) so that the fixtures stay small.
Only data is written."""
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


def unet_layers(filters=(4, 8, 16), bottleneck=32, in_ch=2, seed=0):
    """[(keras layer name, [(weight name, array)])] in model.layers order (pl.py:60-72), weights seeded"""
    rng = np.random.default_rng(seed)
    counters = {}

    def name(kind):
        k = counters.get(kind, 0)
        counters[kind] = k + 1
        return kind if k == 0 else "%s_%d" % (kind, k)

    layers = [("input_1", [])]

    def conv(cin, cout, k=3):
        n = name("conv2d")
        layers.append((n, [("%s/kernel:0" % n, rng.normal(0, (2.0 / (cin * k * k)) ** 0.5, (k, k, cin, cout)).astype(np.float32)),
                           ("%s/bias:0" % n, rng.normal(0, 0.1, cout).astype(np.float32))]))

    def bn(c):
        n = name("batch_normalization")
        layers.append((n, [("%s/gamma:0" % n, (rng.random(c) + 0.5).astype(np.float32)), ("%s/beta:0" % n, rng.normal(0, 0.3, c).astype(np.float32)),
                           ("%s/moving_mean:0" % n, rng.normal(0, 0.2, c).astype(np.float32)),
                           ("%s/moving_variance:0" % n, (rng.random(c) + 0.5).astype(np.float32))]))

    def double(cin, cout):
        conv(cin, cout)
        bn(cout)
        conv(cout, cout)
        bn(cout)

    c = in_ch
    for f in filters:
        double(c, f)
        layers.append((name("max_pooling2d"), []))
        layers.append((name("dropout"), []))
        c = f
    double(c, bottleneck)
    c = bottleneck
    for f in reversed(filters):
        n = name("conv2d_transpose")
        layers.append((n, [("%s/kernel:0" % n, rng.normal(0, (1.0 / (c * 9)) ** 0.5, (3, 3, f, c)).astype(np.float32)),     # Keras: (kh, kw, out, in)
                           ("%s/bias:0" % n, rng.normal(0, 0.1, f).astype(np.float32))]))
        layers.append((name("concatenate"), []))
        layers.append((name("dropout"), []))
        double(2 * f, f)
        c = f
    conv(c, 2, 1)
    return layers


def write_keras_weights(group, layers, **dset_kw):
    group.attrs["layer_names"] = np.asarray([n.encode("utf8") for n, _ in layers])
    group.attrs["backend"] = "tensorflow"
    group.attrs["keras_version"] = "2.4.0"
    for lname, weights in layers:
        g = group.create_group(lname)
        g.attrs["weight_names"] = np.asarray([wn.encode("utf8") for wn, _ in weights]) if weights else np.zeros((0,), dtype="S1")
        for wn, val in weights:
            d = g.create_dataset(wn, val.shape, dtype=val.dtype, **(dset_kw if val.ndim else {}))
            if val.shape:
                d[:] = val
            else:
                d[()] = val


def self_check():
    """the reader against h5py on randomly structured files"""
    import tempfile
    from tissue_image_processing_amd import _hdf5
    rng = np.random.default_rng(7)
    tmp = tempfile.mkdtemp()
    checked = 0
    for trial in range(24):
        libver = "latest" if trial % 2 else "earliest"
        path = os.path.join(tmp, "t%d.h5" % trial)
        expect = {}
        with h5py.File(path, "w", libver=libver, userblock_size=(512 if trial % 5 == 4 else 0)) as f:
            def fill(g, prefix, depth):
                nsub = int(rng.integers(0, 4 if depth else (40 if trial % 3 == 0 else 6)))
                for i in range(int(rng.integers(1, 30 if trial % 4 == 1 else 5))):
                    val = {0: np.int64(rng.integers(-5, 5)), 1: rng.random(3).astype(np.float32), 2: "text %d" % i,
                           3: np.asarray([b"abc", b"defgh", b"i" * int(rng.integers(1, 40))])}[int(rng.integers(0, 4))]
                    g.attrs["a%d" % i] = val
                    expect[(prefix, "attr", "a%d" % i)] = val
                for i in range(int(rng.integers(0, 4))):
                    shape = tuple(int(x) for x in rng.integers(1, 9, int(rng.integers(0, 4))))
                    dt = [np.float32, np.float64, np.int32, np.uint8, ">f4", "<i2"][int(rng.integers(0, 6))]
                    val = (rng.random(shape) * 100).astype(dt)
                    kw = {}
                    mode = int(rng.integers(0, 4)) if shape else 0
                    if mode == 1:
                        kw = dict(chunks=tuple(max(1, s // 2) for s in shape))
                    elif mode == 2:
                        kw = dict(chunks=tuple(max(1, (s + 1) // 2) for s in shape), compression="gzip", shuffle=bool(i % 2), fletcher32=bool(trial % 2))
                    elif mode == 3:
                        kw = dict(chunks=shape)
                    g.create_dataset("d%d" % i, data=val, **kw)
                    expect[(prefix, "data", "d%d" % i)] = val
                if depth < 2:
                    for i in range(nsub):
                        fill(g.create_group("grp_%d_with_a_longer_name" % i if i % 2 else "g%d" % i), prefix + ("grp_%d_with_a_longer_name" % i if i % 2 else "g%d" % i,), depth + 1)
            fill(f, (), 0)
        mine = _hdf5.Hdf5File(path)
        for (prefix, kind, key), val in expect.items():
            obj = mine["/".join(prefix)] if prefix else mine
            if kind == "attr":
                got = obj.attrs[key]
                if isinstance(val, str):
                    assert got == val, (path, prefix, key, got, val)
                else:
                    np.testing.assert_array_equal(np.asarray(got), np.asarray(val), err_msg=str((path, prefix, key)))
            else:
                d = obj[key]
                got = d.read()
                assert got.shape == val.shape and got.dtype == np.dtype(val.dtype).newbyteorder("="), (path, prefix, key, got.dtype, val.dtype)
                np.testing.assert_array_equal(got, val.astype(got.dtype), err_msg=str((path, prefix, key)))
            checked += 1
        # the group listings agree too
        with h5py.File(path, "r") as f:
            def walk(g, mg):
                assert sorted(g.keys()) == sorted(mg.keys()), (path, g.name, sorted(g.keys()), sorted(mg.keys()))
                for k in g.keys():
                    if isinstance(g[k], h5py.Group):
                        walk(g[k], mg[k])
            walk(f, mine)
    # a group wide enough for a two-level B-tree v2 name index, and attributes beyond one fractal-heap block
    path = os.path.join(tmp, "wide.h5")
    with h5py.File(path, "w", libver="latest") as f:
        for i in range(700):
            f.create_group("layer_with_a_long_name_%04d" % i).attrs["weight_names"] = np.asarray([b"w%d" % i])
        for i in range(300):
            f.attrs["attribute_%03d" % i] = np.arange(i % 7 + 1, dtype=np.int32) + i
    mine = _hdf5.Hdf5File(path)
    assert sorted(mine.keys()) == ["layer_with_a_long_name_%04d" % i for i in range(700)]
    assert all(mine["layer_with_a_long_name_%04d" % i].attrs["weight_names"][0] == b"w%d" % i for i in range(0, 700, 37))
    assert all(np.array_equal(mine.attrs["attribute_%03d" % i], np.arange(i % 7 + 1, dtype=np.int32) + i) for i in range(300))
    checked += 1000
    print("self-check: %d attributes / datasets in 24 random files (earliest and latest format) read identically to h5py" % checked)


def main():
    layers = unet_layers()
    p1 = os.path.join(OUT, "keras_tiny_unet_weights.h5")
    with h5py.File(p1, "w") as f:                                    # model.save_weights(path): h5py's defaults, as Keras
        write_keras_weights(f, layers)
    p2 = os.path.join(OUT, "keras_tiny_unet_model.h5")
    with h5py.File(p2, "w", libver="latest") as f:                   # model.save(path): weights under `model_weights`; newest format,
        f.attrs["model_config"] = "{}"                               # compressed chunks: the other code paths of the reader
        write_keras_weights(f.create_group("model_weights"), layers, chunks=True, compression="gzip", shuffle=True)
    flat = {}
    for lname, weights in layers:
        for wn, val in weights:
            flat[wn] = val
    np.savez_compressed(os.path.join(OUT, "keras_tiny_unet_expected.npz"), layer_names=np.asarray([n for n, _ in layers]), **flat)
    for p in (p1, p2):
        print("wrote", p, os.path.getsize(p), "bytes")
    self_check()
    from tissue_image_processing_amd import _hdf5
    for p in (p1, p2):
        got = _hdf5.load_keras_weights_h5(p)
        assert [n for n, _ in got] == [n for n, _ in layers]
        for (_, gw), (_, ew) in zip(got, layers):
            assert [a for a, _ in gw] == [a for a, _ in ew]
            for (_, ga), (_, ea) in zip(gw, ew):
                np.testing.assert_array_equal(ga, ea)
    print("fixtures read back identically")


if __name__ == "__main__":
    main()
