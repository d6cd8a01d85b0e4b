"""Hunt for a rare empty result: the same image through fresh ORBextractor objects (the fuzz's way) and through one reused object.
usage: python3 scripts/repro_loop.py <seed> <case> <reps>"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")]
import mcorb  # noqa: E402
from fuzz_parity import content  # noqa: E402

seed, want, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
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
print("case", want, W, H, "kind", kind, "nf", nf, "sf", sf, "nl", nl, "th", ini, mn, flush=True)
counts = {}
for r in range(reps):
    ext = mcorb.ORBextractor(nf, sf, nl, ini, mn, 0)
    m, k, d = ext(img)
    ext.close()
    counts[len(k)] = counts.get(len(k), 0) + 1
    if r % 500 == 0:
        print("fresh", r, counts, flush=True)
print("fresh extractor per call:", counts, flush=True)
counts = {}
ext = mcorb.ORBextractor(nf, sf, nl, ini, mn, 0)
for r in range(reps * 4):
    m, k, d = ext(img)
    counts[len(k)] = counts.get(len(k), 0) + 1
ext.close()
print("one extractor reused:", counts, flush=True)
