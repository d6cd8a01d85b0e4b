"""Re-run one case of scripts/fuzz_parity.py (same random stream) and print where the GPU and the oracle part ways.
usage: python3 scripts/repro_case.py <seed> <case>"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")]
import mcorb  # noqa: E402
import oracle_lib as O  # noqa: E402
from fuzz_parity import content  # noqa: E402

seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
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
print("case", want, W, H, "kind", kind, "nf", nf, "sf", sf, "nl", nl, "th", ini, mn)
ref = O.OracleExtractor(nf, sf, nl, ini, mn, 0)(img, cap=nf + 64 * nl + 4096)
print("oracle: mono", ref[0], "keypoints", len(ref[1]), "per level", np.bincount(ref[1]["octave"], minlength=nl).tolist())
rig = mcorb.Rig(1, W, H, 1, 1, nfeatures=nf, scale_factor=sf, nlevels=nl, ini_th_fast=ini, min_th_fast=mn)
print("select mode", rig.select_mode())
for rep in range(3):
    rig.upload([img])
    try:
        rig.extract(1)
        m, k, d = rig.features(0)
        print("rep", rep, "gpu: mono", m, "keypoints", len(k), "fallbacks", rig.select_fallbacks(),
              "equal" if len(k) == len(ref[1]) and np.array_equal(d, ref[2]) else "DIFFERENT")
    except Exception as e:
        print("rep", rep, "gpu error:", e)
rig.close()
