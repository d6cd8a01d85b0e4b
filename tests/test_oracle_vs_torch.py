"""Independent sanity pins for two of the oracle's OpenCV restatements, against PyTorch (the only other image-processing
code in this image; CPU only).  They are NOT bit-exact references -- torch works in floating point, OpenCV in the fixed
point the oracle restates -- but they do not share a line of code or a formula with the oracle, so they pin the parts a
restatement gets wrong first: the sampling geometry of cv::resize(INTER_LINEAR) (half-pixel centres, edge clamping) and
the kernel, anchor and BORDER_REFLECT_101 of cv::GaussianBlur(7x7, sigma 2).  Fixed point against float can differ by
one grey level, never by more."""
import numpy as np
import pytest

import oracle_lib as O

torch = pytest.importorskip("torch")
F = torch.nn.functional


def synth(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 128 + 60 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + 40 * ((xx // 16 + yy // 16) % 2) + rng.integers(-20, 21, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("sw,sh,dw,dh", [(1280, 720, 1067, 600), (640, 480, 533, 400), (357, 201, 298, 168), (200, 150, 100, 75),
                                         (333, 217, 257, 181)])
def test_resize_linear_is_torch_bilinear_within_one_level(sw, sh, dw, dh):
    img = synth(sh, sw, 3)
    got = O.resize_linear(img, dw, dh).astype(np.int32)
    ref = F.interpolate(torch.from_numpy(img.astype(np.float64))[None, None], size=(dh, dw), mode="bilinear", align_corners=False)[0, 0].numpy()
    d = np.abs(got - ref)
    assert d.max() < 1.0 + 1e-9, d.max()                    # truncation / rounding of the fixed-point path only
    assert np.mean(np.abs(got - np.rint(ref)) == 0) > 0.80   # and mostly the very grey level of the rounded float result (87 % measured)


@pytest.mark.parametrize("h,w", [(64, 80), (201, 357), (37, 53)])
def test_gaussian_blur_is_float_convolution_within_one_level(h, w):
    img = synth(h, w, 11)
    got = O.gaussian_blur(img).astype(np.int32)
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-x * x / (2 * 2.0 * 2.0))
    k /= k.sum()                                             # cv::getGaussianKernel(7, 2): normalised exp(-x^2 / 2 sigma^2)
    t = torch.from_numpy(img.astype(np.float64))[None, None]
    t = F.pad(t, (3, 3, 3, 3), mode="reflect")               # torch's "reflect" = BORDER_REFLECT_101 (edge pixel not repeated)
    kk = torch.from_numpy(np.outer(k, k))[None, None]
    ref = F.conv2d(t, kk)[0, 0].numpy()
    d = np.abs(got - ref)
    # the 8-bit kernel {18, 34, 48, 56, ...} / 256 is the float kernel rounded (17.97, 33.56, 48.83, 55.28): up to ~0.5 % off per
    # tap, plus the final rounding
    assert d.max() < 1.5, d.max()
    assert np.mean(d < 0.75) > 0.95
