"""Randomised differential test: random image sizes / contents / extractor parameters, GPU path against the CPU oracle,
every keypoint field and descriptor byte.  usage: python3 scripts/fuzz_parity.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import mcorb  # noqa: E402
import oracle_lib as O  # noqa: E402


def content(rng, W, H):
    kind = rng.integers(0, 6)
    base = mcorb.synth_rig_frame(int(rng.integers(0, 1000)), 1, 0, W, H).astype(np.int32)
    if kind == 1:
        base = (base - 128) // int(rng.integers(2, 12)) + 128                      # low contrast
    elif kind == 2:
        base = base + rng.integers(-40, 41, (H, W))                                # noisy
    elif kind == 3:
        base = np.full((H, W), 128) ; y0, x0 = rng.integers(20, H // 2), rng.integers(20, W // 2)
        base[y0:y0 + H // 3, x0:x0 + W // 3] = rng.integers(0, 256, (H // 3, W // 3))   # clustered corners
    elif kind == 4:
        base = np.kron(rng.integers(0, 2, (H // 40 + 1, W // 40 + 1)) * 255, np.ones((40, 40), np.int64))[:H, :W]
    elif kind == 5:
        base = rng.integers(0, 256, (H, W))                                        # pure noise
    return np.clip(base, 0, 255).astype(np.uint8), int(kind)


def rig_cases(ncases, rng):
    """Random rigs: camera count, size, feature budget, match threshold / ratio; pair lists and tracks against the oracle."""
    bad = 0
    for case in range(ncases):
        C = int(rng.integers(2, 7))
        W, H = int(rng.integers(320, 900)), int(rng.integers(240, 640))
        nf = int(rng.integers(200, 1500))
        thr, ratio = float(rng.integers(30, 110)), float(rng.choice([0.6, 0.7, 0.85, 0.95, 1.0]))
        frame = int(rng.integers(0, 500))
        tag = "rig case %d: %d cams %dx%d nf %d thr %.0f ratio %.2f" % (case, C, W, H, nf, thr, ratio)
        imgs = [mcorb.synth_rig_frame(frame, C, c, W, H) for c in range(C)]
        if rng.random() < 0.3:   # one camera sees noise: few matches, different counts per camera
            imgs[int(rng.integers(0, C))] = rng.integers(0, 256, (H, W)).astype(np.uint8)
        try:
            rig = mcorb.Rig(C, W, H, 1, 1, nfeatures=nf)
        except Exception as e:   # sizes the reference's cell / root-node arithmetic cannot handle (MCORB_E_SIZE); the oracle says -2
            assert "error -2" in str(e) and O.OracleExtractor(nf)(imgs[0])[0] == -2, (tag, e)
            print(tag, "both refuse:", str(e)[:50])
            continue
        rig.upload(imgs)
        rig.process(1, dist_thresh=thr, ratio=ratio)
        ora = [O.OracleExtractor(nf)(im, cap=nf + 4096) for im in imgs]
        ok = True
        for c in range(C):
            m, k, d = rig.features(c)
            ok &= m == ora[c][0] and np.array_equal(d, ora[c][2]) and all(np.array_equal(k[f], ora[c][1][f]) for f in k.dtype.names)
        for i in range(C - 1):
            for j in range(i + 1, C):
                g1, g2 = rig.pair_matches(0, i, j)
                o1, o2 = O.bruteforce_match(ora[i][2], ora[j][2], thr, ratio)
                ok &= np.array_equal(g1, o1) and np.array_equal(g2, o2)
        tr, mg = rig.tracks(0)
        otr, omg = O.intra_matches([o[2] for o in ora], thr, ratio)
        ok &= np.array_equal(tr, otr) and mg == omg
        rig.close()
        if not ok:
            bad += 1
            print(tag, "MISMATCH")
        elif case % 5 == 0:
            print(tag, "ok,", len(tr), "tracks", flush=True)
    return bad


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    if len(sys.argv) > 3 and sys.argv[3] == "rig":
        bad = rig_cases(ncases, rng)
        print("fuzz: %d cases, %d bad" % (ncases, bad))
        return 1 if bad else 0
    bad = 0
    for case in range(ncases):
        W, H = int(rng.integers(160, 1700)), int(rng.integers(120, 1200))
        if W > 2.4 * H:
            W = int(2.4 * H)
        if H > 1.4 * W:
            H = int(1.4 * W)
        nf = int(rng.integers(50, 3500))
        sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5, 2.0]))
        nl = int(rng.integers(1, 10))
        ini, mn = int(rng.integers(5, 40)), int(rng.integers(3, 25))
        orient = int(os.environ.get("FUZZ_ORIENT", "0")) and int(rng.integers(0, 2))   # IC-angle mode (not the reference's default)
        img, kind = content(rng, W, H)
        tag = "case %d: %dx%d kind %d nf %d sf %.1f nl %d th %d/%d orient %d" % (case, W, H, kind, nf, sf, nl, ini, mn, orient)
        if os.environ.get("FUZZ_ONLY") and case != int(os.environ["FUZZ_ONLY"]):   # same random stream, one case
            continue
        try:
            ora = O.OracleExtractor(nf, sf, nl, ini, mn, orient)
            ref = ora(img, cap=nf + 64 * nl + 4096)
        except Exception as e:
            print(tag, "oracle refused:", e)
            continue
        try:
            ext = mcorb.ORBextractor(nf, sf, nl, ini, mn, orient)
            got = ext(img)
            ext.close()
        except Exception as e:
            if ref[0] == -2 or "too small" in str(e) or "too tall" in str(e) or "too wide" in str(e):
                print(tag, "both refuse / size limit:", str(e)[:60])
                continue
            print(tag, "GPU ERROR", e); bad += 1
            continue
        (m1, k1, d1), (m2, k2, d2) = ref, got
        if m1 < 0:
            print(tag, "oracle status", m1, "but GPU returned", m2)
            bad += m1 != m2
            continue
        ok = m1 == m2 and len(k1) == len(k2) and all(np.array_equal(k1[f], k2[f]) for f in k1.dtype.names)
        if ok and orient:   # rotated taps go through cos/sin of two different libms (tests/test_gpu_parity.py's orientation test)
            same = np.all(d1 == d2, axis=1)
            ok = len(same) == 0 or same.mean() >= 0.995
            if ok and not same.all():
                print(tag, "orientation mode: %d of %d descriptors differ (libm last-ulp)" % ((~same).sum(), len(same)))
        elif ok:
            ok = np.array_equal(d1, d2)
        if not ok:
            bad += 1
            print(tag, "MISMATCH", m1, m2, len(k1), len(k2))
            if len(k1) == len(k2):
                for f in k1.dtype.names:
                    w = np.flatnonzero(k1[f] != k2[f])
                    if len(w):
                        print("   field %s: %d differ, first at %d: oracle %r gpu %r" % (f, len(w), w[0], k1[f][w[0]], k2[f][w[0]]))
                w = np.flatnonzero((d1 != d2).any(1))
                if len(w):
                    print("   descriptors: %d rows differ, first at %d; keypoint there: oracle %r gpu %r" % (len(w), w[0], k1[w[0]], k2[w[0]]))
        elif case % 10 == 0:
            print(tag, "ok,", len(k1), "keypoints", flush=True)
    print("fuzz: %d cases, %d bad" % (ncases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
