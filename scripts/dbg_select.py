import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import numpy as np
import mcorb
W, H = 800, 600
rng = np.random.default_rng(42)
base = np.full((H, W), 128, np.int64)
base[200:330, 420:560] = rng.integers(0, 256, (130, 140))
base[60:110, 80:150] = rng.integers(0, 2, (50, 70)) * 255
img = base.astype(np.uint8)
for sel in (2, 1):
    rig = mcorb.Rig(1, W, H, 1, 1, nfeatures=1000, selection=sel)
    rig.upload([img]); rig.extract(1)
    m, k, d = rig.features(0)
    print("mode", rig.select_mode(), "fallbacks", rig.select_fallbacks(), "n", len(k), "per level", np.bincount(k["octave"], minlength=8),
          "cands", [len(rig.candidates(0, l)[0]) for l in range(8)], rig.timing())
    rig.close()
