"""GPU: drift estimation (hand-written FFT + upsampled DFT) vs goldens from the reference and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_drift_goldens(golden):
    from tissue_image_processing_amd import tissue_info as ti, basic_image_manipulations as bim
    g = golden("drift")
    for tag in "abc":
        imgs = g[tag + "_images"]
        t = ti.Tissue(2)
        sy, sx = t.update_drift(2, 1, images=imgs, image_in_memory=True)
        np.testing.assert_array_equal(np.array([sy, sx]), g[tag + "_drift"])
        np.testing.assert_array_equal(t.drifts[1], g[tag + "_drifts_row"])
        np.testing.assert_array_equal(bim.calculate_drift(imgs[0], imgs[1]), g[tag + "_calc"])
        np.testing.assert_array_equal(bim.calculate_drift(imgs[0], imgs[1], sub_pixel_precision=False), g[tag + "_calc_whole"])
        np.testing.assert_array_equal(bim.calculate_drift(g[tag + "_prev_f64"], g[tag + "_cur_f64"]), g[tag + "_calc_f64"])


@pytest.mark.parametrize("shape,shift", [((512, 1024), (7.31, -12.77)), ((2048, 2048), (-0.43, 0.61)), ((4, 8), (1.0, 0.0)),
                                         # extents that are not powers of two (what calculate_refine_drift's overlap crop
                                         # produces, ti.py:1952-1975): Bluestein rows, primes included
                                         ((500, 731), (3.27, -8.4)), ((2041, 2037), (-5.38, 9.12)), ((97, 101), (2.5, 1.25)),
                                         ((3, 5), (1.0, 0.0)), ((4096, 100), (0.3, -0.7))])
def test_drift_vs_oracle(shape, shift):
    from oracle import oracle as orc
    from tissue_image_processing_amd._registration import phase_cross_correlation
    rng = np.random.default_rng(shape[0] + shape[1])
    base = orc.blur_image(rng.random((shape[0], shape[1])), 2.0) if min(shape) > 16 else rng.random(shape)
    # circular sub-pixel shift in Fourier space
    fy = np.fft.fftfreq(shape[0])[:, None]
    fx = np.fft.fftfreq(shape[1])[None, :]
    moved = np.real(np.fft.ifft2(np.fft.fft2(base) * np.exp(-2j * np.pi * (fy * shift[0] + fx * shift[1]))))
    a = np.round(base * 30000).astype(np.uint16)
    b = np.round(np.clip(moved, 0, None) * 30000).astype(np.uint16)
    got, _, _ = phase_cross_correlation(a, b, upsample_factor=100)
    ref = orc.phase_cross_correlation(a, b, upsample_factor=100)
    np.testing.assert_array_equal(got, ref)
    if min(shape) > 16:
        np.testing.assert_allclose(got, [-shift[0], -shift[1]], atol=0.02)


def test_drift_errors():
    from tissue_image_processing_amd._registration import phase_cross_correlation
    with pytest.raises(NotImplementedError):
        phase_cross_correlation(np.zeros((4100, 16)), np.zeros((4100, 16)))
    with pytest.raises(ValueError):
        phase_cross_correlation(np.zeros((64, 128)), np.zeros((128, 64)))


def test_refine_drift_crops_to_odd_extents():
    """ti.py:1941-1980 with a non-zero stage shift: the overlap crop has arbitrary extents."""
    from oracle import oracle as orc
    from tissue_image_processing_amd.tissue_info import TissueHipMixin
    rng = np.random.default_rng(11)
    base = orc.blur_image(rng.random((600, 640)), 2.0)
    prev = np.round(base[20:532, 30:542] * 30000).astype(np.uint16)
    cur = np.round(base[13:525, 41:553] * 30000).astype(np.uint16)      # content moved by (+7, -11) (row, col)
    got = TissueHipMixin.calculate_refine_drift(prev, cur, -10.6, 6.9)  # coarse (x, y) as the stage table would give
    rx, ry = int(np.floor(-10.6)), int(np.floor(6.9))
    p = prev[:rx, ry:]
    c = cur[-rx:, :-ry]
    assert p.shape == c.shape and (p.shape[0] & (p.shape[0] - 1)) and (p.shape[1] & (p.shape[1] - 1))
    ref = orc.phase_cross_correlation(p, c, upsample_factor=100)
    assert got == (rx + ref[-2], ry + ref[-1])
