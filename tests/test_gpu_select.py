"""DistributeOctTree's list discipline on the GPU (k_select / k_assemble, MCORB_SELECT_GPU): the device sort against libstdc++'s
std::sort, and the GPU-selected pipeline against the host-selected one and the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mc():
    import mcorb
    return mcorb


def _dev_sort(keys):
    from importlib import import_module
    L = import_module("mc-slam_amd")._lib.load()
    keys = np.ascontiguousarray(keys, np.uint32)
    pd, ps = np.zeros(len(keys), np.uint32), np.zeros(len(keys), np.uint32)
    assert L.mcorb_dev_sort_selftest(0, keys.ctypes.data, len(keys), pd.ctypes.data, ps.ctypes.data) == 0
    return pd, ps


def test_device_sort_leaves_std_sorts_permutation():
    """wave_std_sort (one GPU wave) == std::sort (libstdc++, run inside the same call) on multisets full of ties: every entry in the
    same place, for sizes around the 16-element threshold, the sizes DistributeOctTree sorts (32 .. 512) and beyond"""
    rng = np.random.default_rng(7)
    n_cases = 0
    for n in list(range(0, 40)) + [63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 300, 434, 512, 700, 1024, 1519, 3000]:
        for alpha in (1, 2, 3, 17, 200, 1 << 20, "octree"):
            if alpha == "octree":
                k = (rng.integers(2, 40, n).astype(np.uint32) << 12) | (rng.integers(0, 24, n) * 39).astype(np.uint32)
            else:
                k = rng.integers(0, alpha, n).astype(np.uint32)
            pd, ps = _dev_sort(k)
            assert np.array_equal(pd, ps), "n %d alphabet %s: first difference at %d" % (n, alpha, int(np.argmax(pd != ps)))
            n_cases += 1
    for n in (128, 512):   # structured
        for k in (np.arange(n), np.arange(n)[::-1], np.arange(n) % 7, np.arange(n) // 5, np.minimum(np.arange(n), n - np.arange(n))):
            pd, ps = _dev_sort(k.astype(np.uint32))
            assert np.array_equal(pd, ps)
    assert n_cases > 300


def test_device_sort_heap_branch_on_median_of_three_killers():
    """sequences built by McIlroy's adversary against std::sort itself (tests/cpp/test_sortmodel.cpp --dump-killer) use up the
    introsort depth budget: the device takes its single-lane heap-sort branch and still lands on std::sort's permutation"""
    exe = os.path.join(ROOT, "tests", "cpp", "test_sortmodel_plain")
    src = os.path.join(ROOT, "tests", "cpp", "test_sortmodel.cpp")
    if not os.path.exists(exe) or os.path.getmtime(src) > os.path.getmtime(exe):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "mc-slam_amd", "csrc"), src, "-o", exe])
    for n in (64, 128, 434, 1024):
        out = subprocess.run([exe, "--dump-killer", str(n)], capture_output=True, text=True, check=True).stdout.split()
        k = np.array(out, np.uint32)
        assert len(k) == n
        for keys in (k, k // 3):
            pd, ps = _dev_sort(keys)
            assert np.array_equal(pd, ps), "killer n %d" % n


def _all_outputs(rig, C):
    feats = [rig.features(c) for c in range(C)]
    tr = rig.tracks(0) if C > 1 else None
    return feats, tr


@pytest.mark.parametrize("W,H,C,N", [(640, 480, 2, 1000), (1280, 720, 4, 2000)])
def test_gpu_selected_job_equals_host_selected_job_and_oracle(mc, W, H, C, N):
    rigs = {m: mc.Rig(C, W, H, 1, 1, nfeatures=N, selection=sel) for m, sel in (("host", 1), ("gpu", 2))}
    assert rigs["gpu"].select_mode() == "gpu" and rigs["host"].select_mode() == "host"
    ex = O.OracleExtractor(N)
    for f in (0, 3):
        imgs = [mc.synth_rig_frame(f, C, c, W, H) for c in range(C)]
        outs = {}
        for m, rig in rigs.items():
            rig.upload(imgs)
            rig.process(1)
            outs[m] = _all_outputs(rig, C)
        refs = [ex(im) for im in imgs]
        for c in range(C):
            for m in ("host", "gpu"):
                (m1, k1, d1), (m2, k2, d2) = refs[c], outs[m][0][c]
                assert m1 == m2 and len(k1) == len(k2), "%s cam %d" % (m, c)
                for fld in k1.dtype.names:
                    assert np.array_equal(k1[fld], k2[fld]), "%s cam %d field %s" % (m, c, fld)
                assert np.array_equal(d1, d2), "%s cam %d descriptors" % (m, c)
        otr, omg = O.intra_matches([r[2] for r in refs])
        for m in ("host", "gpu"):
            tr, mg = outs[m][1]
            assert np.array_equal(tr, otr) and mg == omg, m
    assert rigs["gpu"].select_fallbacks() == 0, "the bench frames must not need the host stage"
    for r in rigs.values():
        r.close()


def _clustered_image(W, H, seed=42, patches=40):
    rng = np.random.default_rng(seed)
    base = np.full((H, W), 128, np.int64)
    for _ in range(patches):
        x, y = int(rng.integers(30, W - 60)), int(rng.integers(30, H - 60))
        base[y:y + 24, x:x + 24] = rng.integers(0, 256, (24, 24))
    return base.astype(np.uint8)


@pytest.mark.parametrize("batch", [1, 9])   # up to 8 images the results travel through host-mapped memory (signalled per image), above by copies
@pytest.mark.parametrize("deep_cap,fallbacks", [(None, 0), ("8", 1)])
def test_clustered_corners_below_the_bucketing_depth(mc, monkeypatch, deep_cap, fallbacks, batch):
    """corners only inside forty small patches: fewer non-empty buckets than the level's quota, so the tree divides nodes BELOW the
    bucketing depth.  k_select does that itself (the bucket's candidates filtered by the split lines of the node's path); with the
    scan cap turned down to 8 candidates (test knob) it raises the flag instead and the batch is redone through the host stage.
    Either way the result is the oracle's."""
    if deep_cap:
        monkeypatch.setenv("MCORB_SELECT_DEEP_CAP", deep_cap)
    W, H = 800, 600
    img = _clustered_image(W, H)
    rig = mc.Rig(1, W, H, batch, 1, nfeatures=1000, selection=2)
    plain = mc.synth_rig_frame(3, 1, 0, W, H)
    imgs = [img if m == batch - 1 else plain for m in range(batch)]   # the clustered one last: the others must come out right beside it
    rig.upload(imgs)
    rig.extract(batch)
    for m in (0, batch - 1):
        ref = O.OracleExtractor(1000)(imgs[m])
        m2, k2, d2 = rig.features(m)
        assert ref[0] == m2 and len(ref[1]) == len(k2) and len(k2) > 300
        for fld in ref[1].dtype.names:
            assert np.array_equal(ref[1][fld], k2[fld]), fld
        assert np.array_equal(ref[2], d2)
    assert rig.select_fallbacks() == fallbacks
    rig.close()


@pytest.mark.parametrize("seed", range(6))
def test_deep_trees_on_the_gpu_random_clusters(mc, seed):
    """more clustered layouts (few / many patches, different sizes and budgets): no level may need the host stage, and the
    keypoints are the oracle's"""
    rng = np.random.default_rng(100 + seed)
    W, H = int(rng.integers(500, 1100)), int(rng.integers(400, 700))
    img = _clustered_image(W, H, seed=seed, patches=int(rng.integers(3, 60)))
    nf = int(rng.integers(300, 2000))
    rig = mc.Rig(1, W, H, 1, 1, nfeatures=nf, selection=2)
    rig.upload([img])
    rig.extract(1)
    ref = O.OracleExtractor(nf)(img)
    m2, k2, d2 = rig.features(0)
    assert ref[0] == m2 and len(ref[1]) == len(k2)
    for fld in ref[1].dtype.names:
        assert np.array_equal(ref[1][fld], k2[fld]), fld
    assert np.array_equal(ref[2], d2)
    assert rig.select_fallbacks() == 0
    rig.close()


@pytest.mark.parametrize("batch", [1, 9])
def test_lapping_partition_on_the_gpu(mc, batch):
    """operator()'s stereo / mono partition (k_assemble) with a lapping area, against the oracle"""
    W, H = 640, 480
    img = mc.synth_rig_frame(1, 1, 0, W, H)
    rig = mc.Rig(1, W, H, batch, 1, nfeatures=800, selection=2)
    rig.upload([mc.synth_rig_frame(2 + m, 1, 0, W, H) for m in range(batch - 1)] + [img])
    rig.extract(batch, lap=(200, 420))
    ref = O.OracleExtractor(800)(img, lap=(200, 420))
    m2, k2, d2 = rig.features(batch - 1)
    assert ref[0] == m2 and 0 < m2 < len(k2)
    for fld in ref[1].dtype.names:
        assert np.array_equal(ref[1][fld], k2[fld]), fld
    assert np.array_equal(ref[2], d2)
    rig.close()


@pytest.mark.parametrize("frames", [1, 5])   # 2 images: host-mapped results; 10 images: copied results
def test_job_redone_by_the_host_stage_still_matches(mc, monkeypatch, frames):
    """extract + match in one job when a level's tree makes k_select raise its flag (scan cap turned down): the host stage redoes the
    selection, and descriptors, k-NN tables and tracks of every frame are the oracle's"""
    monkeypatch.setenv("MCORB_SELECT_DEEP_CAP", "8")
    W, H, C, N = 800, 600, 2, 1000
    imgs = []
    for f in range(frames):
        imgs += [_clustered_image(W, H, seed=7 + f), _clustered_image(W, H, seed=7 + f, patches=44)]
    rig = mc.Rig(C, W, H, frames, 1, nfeatures=N, selection=2)
    rig.upload(imgs)
    rig.process(frames)
    assert rig.select_fallbacks() == 1
    for f in (0, frames - 1):
        descs = []
        for c in range(C):
            ref = O.OracleExtractor(N)(imgs[f * C + c])
            m2, k2, d2 = rig.features(f * C + c)
            assert ref[0] == m2 and len(k2) == len(ref[1]) > 100
            assert np.array_equal(ref[1]["x"], k2["x"]) and np.array_equal(ref[1]["y"], k2["y"]) and np.array_equal(ref[2], d2)
            descs.append(d2)
        otr, omg = O.intra_matches(descs)
        tr, mg = rig.tracks(f)
        assert np.array_equal(tr, otr) and mg == omg
        gi, gd = rig.pair_knn2(f, 0, 1)
        oi, od = O.knn2(descs[0], descs[1])
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    rig.close()
