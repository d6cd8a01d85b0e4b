"""GPU parity over parameter / content variations the BASELINE configs do not cover:
odd image sizes, other scale factors and level counts, equal thresholds, feature budgets,
low-contrast / noise-only / saturated content (threshold-fallback and candidate-capacity paths),
and rigs with more than the reference's 5 cameras."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mc():
    import mcorb
    return mcorb


def same(ref, got, what):
    (m1, k1, d1), (m2, k2, d2) = ref, got
    assert m1 == m2 and len(k1) == len(k2), "%s: count %d/%d mono %d/%d" % (what, len(k1), len(k2), m1, m2)
    for f in k1.dtype.names:
        assert np.array_equal(k1[f], k2[f]), "%s: keypoint field %s" % (what, f)
    assert np.array_equal(d1, d2), "%s: descriptors" % what


@pytest.mark.parametrize("W,H", [(641, 479), (1000, 750), (1241, 376), (803, 601)])
def test_odd_sizes(mc, W, H):
    img = mc.synth_rig_frame(2, 1, 0, W, H)
    same(O.OracleExtractor(1200)(img), mc.ORBextractor(1200, 1.2, 8, 20, 7)(img), "%dx%d" % (W, H))


@pytest.mark.parametrize("nfeat,sf,nl,ini,mn", [(500, 1.2, 8, 20, 7), (3000, 1.2, 8, 20, 7), (1000, 1.5, 4, 20, 7),
                                                (1500, 1.1, 12, 20, 7), (1000, 1.2, 8, 10, 10), (800, 1.2, 1, 30, 5),
                                                (1000, 1.3, 6, 7, 20), (1000, 2.0, 3, 20, 7)])
def test_extractor_parameters(mc, nfeat, sf, nl, ini, mn):
    img = mc.synth_rig_frame(1, 1, 0, 960, 600)
    ora = O.OracleExtractor(nfeat, sf, nl, ini, mn)
    ext = mc.ORBextractor(nfeat, sf, nl, ini, mn)
    same(ora(img), ext(img), "params %s" % ((nfeat, sf, nl, ini, mn),))
    t = ora.tables()
    assert np.array_equal(ext.GetScaleFactors(), t["scale"])


def _content(kind, W=800, H=600):
    rng = np.random.default_rng(42)
    base = None
    if kind == "low_contrast":      # everything within +-12 grey levels: most cells fall back to minThFAST
        import mcorb
        base = (mcorb.synth_rig_frame(0, 1, 0, W, H).astype(np.int32) - 128) // 10 + 128
    elif kind == "noise":           # dense candidates everywhere: stresses cell slots and the candidate buffer
        base = rng.integers(0, 256, (H, W))
    elif kind == "saturated":       # large flat black / white blocks with hard edges
        base = np.kron(rng.integers(0, 2, (H // 50, W // 50)) * 255, np.ones((50, 50), np.int64))
    elif kind == "gradient":        # smooth ramp + a few dots: almost no corners (empty cells, tiny levels)
        base = np.add.outer(np.arange(H), np.arange(W)) * 255 // (H + W)
        base[100:104, 100:104] = 255
        base[300:303, 500:503] = 0
    elif kind == "clustered":       # every corner inside two small patches: the quad-tree goes far deeper than the GPU bucketing
        base = np.full((H, W), 128, np.int64)
        base[200:330, 420:560] = rng.integers(0, 256, (130, 140))
        base[60:110, 80:150] = rng.integers(0, 2, (50, 70)) * 255
    return np.clip(base, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("W,H,nfeat", [(1800, 600, 7000), (1280, 720, 6000)])
def test_large_feature_budget_on_a_wide_image(mc, W, H, nfeat):
    """Level 0 gets > 1 300 features: the quad-tree bucketing goes to depth 5 (3 x 1024 / 2 x 1024 buckets), where k_compact's four
    private counter copies no longer fit 64 KiB of LDS and it runs with one; noise content so that the budget is really used."""
    rng = np.random.default_rng(W + nfeat)
    img = np.clip(mc.synth_rig_frame(3, 1, 0, W, H).astype(np.int32) + rng.integers(-60, 61, (H, W)), 0, 255).astype(np.uint8)
    ora = O.OracleExtractor(nfeat)
    ext = mc.ORBextractor(nfeat, 1.2, 8, 20, 7)
    ref = ora(img, cap=nfeat + 4096)
    same(ref, ext(img), "%dx%d nfeatures %d" % (W, H, nfeat))
    assert len(ref[1]) > 0.8 * nfeat


@pytest.mark.parametrize("kind", ["low_contrast", "noise", "saturated", "gradient", "clustered"])
def test_image_content(mc, kind):
    img = _content(kind)
    ref = O.OracleExtractor(1000)(img)
    got = mc.ORBextractor(1000, 1.2, 8, 20, 7)(img)
    same(ref, got, kind)
    if kind == "noise":
        assert len(ref[1]) >= 1000


def _four_point_survivors_per_cell(img, T):
    """Survivors of k_fast_cells' pass-1 test (two neighbouring compass points beyond the threshold) per 35-px cell of
    level 0 -- numpy restatement, used to make sure an image really drives the kernel's chunked work-list path."""
    v = img.astype(np.int32)

    def sh(dx, dy):
        return np.roll(np.roll(v, -dy, 0), -dx, 1)
    U, D, L, R = sh(0, -3), sh(0, 3), sh(-3, 0), sh(3, 0)
    w = np.maximum(np.minimum(np.maximum(U, D), np.maximum(L, R)) - v, v - np.maximum(np.minimum(U, D), np.minimum(L, R)))
    H, W = img.shape
    x1, y1 = W - 16, H - 16
    nC, nR = (x1 - 16) // 35, (y1 - 16) // 35
    wC, hC = -(-(x1 - 16) // nC), -(-(y1 - 16) // nR)
    out = []
    for i in range(nR):
        for j in range(nC):
            ya, xa = 16 + i * hC + 3, 16 + j * wC + 3
            out.append(int((w[ya:min(ya + hC, y1 - 3), xa:min(xa + wC, x1 - 3)] > T).sum()))
    return np.array(out)


@pytest.mark.parametrize("ini,mn", [(20, 7), (5, 5), (60, 3)])
def test_fast_worklist_chunking(mc, ini, mn):
    """Cells with more pass-1 survivors than k_fast_cells' work list holds (kFastListCap = 768): the kernel scores them
    chunk by chunk and sweeps the cell a second time for the NMS.  Pure noise gives 500-1300 survivors per cell."""
    img = _content("noise", 640, 480)
    assert (_four_point_survivors_per_cell(img, ini) > 800).any() or (_four_point_survivors_per_cell(img, mn) > 800).any()
    same(O.OracleExtractor(1500, 1.2, 8, ini, mn)(img), mc.ORBextractor(1500, 1.2, 8, ini, mn)(img), "noise %d/%d" % (ini, mn))
    # half noise, half flat: chunked and ordinary cells side by side, empty cells retried at minTh
    img2 = img.copy()
    img2[:, 320:] = 90
    img2[200:260, 400:500] = _content("noise", 100, 60)
    same(O.OracleExtractor(1500, 1.2, 8, ini, mn)(img2), mc.ORBextractor(1500, 1.2, 8, ini, mn)(img2), "half noise %d/%d" % (ini, mn))


@pytest.mark.parametrize("C", [5, 8])
def test_rigs_wider_than_the_reference_track_type(mc, C):
    W, H, N = 480, 360, 400
    imgs = [mc.synth_rig_frame(3, C, c, W, H) for c in range(C)]
    rig = mc.Rig(C, W, H, 1, 1, nfeatures=N)
    rig.upload(imgs)
    rig.process_submit(1)
    rig.process_wait()
    ora = [O.OracleExtractor(N)(im) for im in imgs]
    for c in range(C):
        same(ora[c], rig.features(c), "cam %d" % c)
    for i in range(C - 1):
        for j in range(i + 1, C):
            g1, g2 = rig.pair_matches(0, i, j)
            o1, o2 = O.bruteforce_match(ora[i][2], ora[j][2])
            assert np.array_equal(g1, o1) and np.array_equal(g2, o2), (i, j)
    tr, mg = rig.tracks(0)
    otr, omg = O.intra_matches([o[2] for o in ora])
    assert tr.shape[1] == C and np.array_equal(tr, otr) and mg == omg
    rig.close()


def test_baseline_config2_eight_cameras_1080p(mc):
    """BASELINE configs[2] on one GPU: 8-camera rig, 1920x1080, 2000 keypoints per camera, all 28 camera pairs."""
    C, W, H, N = 8, 1920, 1080, 2000
    imgs = [mc.synth_rig_frame(11, C, c, W, H) for c in range(C)]
    rig = mc.Rig(C, W, H, 1, 1, nfeatures=N)
    rig.upload(imgs)
    rig.process(1)
    ora = [O.OracleExtractor(N)(im) for im in imgs]
    for c in range(C):
        same(ora[c], rig.features(c), "cam %d" % c)
    tr, mg = rig.tracks(0)
    otr, omg = O.intra_matches([o[2] for o in ora])
    assert np.array_equal(tr, otr) and mg == omg and len(tr) > 2000
    rig.close()


def test_matcher_thresholds(mc):
    W, H, C, N = 640, 480, 2, 800
    imgs = [mc.synth_rig_frame(9, C, c, W, H) for c in range(C)]
    rig = mc.Rig(C, W, H, 1, 1, nfeatures=N)
    rig.upload(imgs)
    rig.extract(C)
    d = [rig.features(c)[2] for c in range(C)]
    for thr, ratio in ((75.0, 0.85), (50.0, 0.7), (100.0, 1.0), (30.0, 0.5), (256.0, 2.0)):
        rig.match(1, dist_thresh=thr, ratio=ratio)
        g1, g2 = rig.pair_matches(0, 0, 1)
        o1, o2 = O.bruteforce_match(d[0], d[1], thr, ratio)
        assert np.array_equal(g1, o1) and np.array_equal(g2, o2), (thr, ratio)
    rig.close()
