"""Independent sanity pins for two of the oracle's OpenCV restatements, against PyTorch (the only other image-processing
code in this image; CPU only).  They are NOT bit-exact references -- torch works in floating point, OpenCV in the fixed
point the oracle restates -- but they do not share a line of code or a formula with the oracle, so they pin the parts a
restatement gets wrong first: the sampling geometry of cv::resize(INTER_LINEAR) (half-pixel centres, edge clamping) and
the kernel, anchor and BORDER_REFLECT_101 of cv::GaussianBlur(7x7, sigma 2).  Fixed point against float can differ by
one grey level, never by more."""
import numpy as np
import pytest

import oracle_lib as O



def _torch():
    # imported inside the tests, not at collection: a `-m gpu` run collects this file too, and torch brings its own copy of
    # the HIP runtime into the process, which the GPU tests that talk to libamdhip64 directly
    # must not find loaded first
    torch = pytest.importorskip("torch")
    return torch, torch.nn.functional


def synth(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 128 + 60 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + 40 * ((xx // 16 + yy // 16) % 2) + rng.integers(-20, 21, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("sw,sh,dw,dh", [(1280, 720, 1067, 600), (640, 480, 533, 400), (357, 201, 298, 168), (200, 150, 100, 75),
                                         (333, 217, 257, 181)])
def test_resize_linear_is_torch_bilinear_within_one_level(sw, sh, dw, dh):
    torch, F = _torch()
    img = synth(sh, sw, 3)
    got = O.resize_linear(img, dw, dh).astype(np.int32)
    ref = F.interpolate(torch.from_numpy(img.astype(np.float64))[None, None], size=(dh, dw), mode="bilinear", align_corners=False)[0, 0].numpy()
    d = np.abs(got - ref)
    assert d.max() < 1.0 + 1e-9, d.max()                    # truncation / rounding of the fixed-point path only
    assert np.mean(np.abs(got - np.rint(ref)) == 0) > 0.80   # and mostly the very grey level of the rounded float result (87 % measured)


@pytest.mark.parametrize("h,w", [(64, 80), (201, 357), (37, 53)])
def test_gaussian_blur_is_float_convolution_within_one_level(h, w):
    torch, F = _torch()
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


# cv::FAST (FAST-9/16 + cornerScore + 3x3 non-max suppression) against the textbook definition, written with whole-array
# numpy operations -- no code or formula shared with the oracle's pixel loops.
RING16 = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1),
          (-2, 2), (-1, 3)]


def fast_by_definition(img, T):
    img = img.astype(np.int32)
    H, W = img.shape
    v = img[3:H - 3, 3:W - 3]
    ring = np.stack([img[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx] for dx, dy in RING16])
    ext = np.concatenate([ring, ring[:8]])                                   # arcs wrap around
    bright = np.max(np.stack([np.min(ext[s:s + 9] - v, axis=0) for s in range(16)]), axis=0)   # best arc's weakest margin
    dark = np.max(np.stack([np.min(v - ext[s:s + 9], axis=0) for s in range(16)]), axis=0)
    A = np.maximum(bright, dark)
    corner = A > T                                                           # all nine strictly beyond v +- T
    score = np.where(corner, A - 1, 0)                                       # largest t for which it still is one
    full = np.zeros((H, W), np.int32)
    full[3:H - 3, 3:W - 3] = score
    return corner, score, full


@pytest.mark.parametrize("T", [7, 20])
def test_fast_is_the_segment_test_by_definition(T):
    img = synth(90, 120, 5)
    corner, score, full = fast_by_definition(img, T)
    ys, xs = np.nonzero(corner)
    want = sorted(zip((xs + 3).tolist(), (ys + 3).tolist(), score[ys, xs].tolist()), key=lambda t: (t[1], t[0]))
    gx, gy, gs = O.fast(img, T, nonmax=False)
    assert len(want) > 50
    assert list(zip(gx.tolist(), gy.tolist())) == [(x, y) for x, y, _ in want]     # raster order; cv::FAST leaves the response 0 here
    assert not gs.any()
    # 3x3 non-max suppression: strictly above all eight neighbours (non-corners count as 0)
    H, W = img.shape
    keep = []
    for x, y, s in want:
        nb = full[y - 1:y + 2, x - 1:x + 2].copy()
        nb[1, 1] = -1
        if s > nb.max():
            keep.append((x, y, s))
    gx, gy, gs = O.fast(img, T, nonmax=True)
    assert list(zip(gx.tolist(), gy.tolist(), gs.tolist())) == keep
