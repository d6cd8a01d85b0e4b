"""GPU-only consistency stress: random sizes / parameters, a fresh extractor per case (the fuzz's way), two different images
alternating A B A B: call 3 must equal call 1 and call 4 call 2, in every keypoint field and descriptor byte.  A result read by the host
before it had landed shows as an empty first result (the buffers of a fresh extractor hold zeros) or as the OTHER image's keypoints.
usage: python3 scripts/stress_consistency.py <cases> <seed>"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")]
import mcorb  # noqa: E402
from fuzz_parity import content  # noqa: E402

ncases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = done = 0
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
    img, kind = content(rng, W, H)
    try:
        ext = mcorb.ORBextractor(nf, sf, nl, ini, mn, 0)
        img2 = np.ascontiguousarray(img[::-1, ::-1])
        res = [ext(img), ext(img2), ext(img), ext(img2)]
        ext.close()
    except Exception as e:
        if "error -2" in str(e) or "too" in str(e):
            continue
        print("case", case, "ERROR", e, flush=True)
        bad += 1
        continue
    done += 1
    def same(a, b):
        return a[0] == b[0] and len(a[1]) == len(b[1]) and np.array_equal(a[2], b[2]) and all(np.array_equal(a[1][f], b[1][f]) for f in a[1].dtype.names)
    ok = same(res[0], res[2]) and same(res[1], res[3])
    if not ok:
        bad += 1
        print("case %d: %dx%d kind %d nf %d sf %.1f nl %d th %d/%d INCONSISTENT: counts %s" % (case, W, H, kind, nf, sf, nl, ini, mn, [len(r[1]) for r in res]), flush=True)
    if case % 1000 == 0:
        print("case", case, "done", done, "bad", bad, flush=True)
print("stress: %d cases run, %d bad" % (done, bad))
