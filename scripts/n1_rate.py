"""inter-frame knnMatch between the LF descriptor sets of consecutive keyframes (3000 x 3000, ratio 0.7, threshold 50,
FrontEnd.cpp:3344-3500).  Device-resident path (mcorb_descblock + mcorb_rig_match_sets: one set uploaded per keyframe, the previous
one stays in HBM) against mcorb_match_ratio (both sets from the host every call).    python scripts/n1_rate.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mcorb  # noqa: E402

N, REPS = 3000, 60
rng = np.random.default_rng(3)
base = rng.integers(0, 256, (N, 32), dtype=np.uint8)


def keyframe(k):
    d = base.copy()
    flip = rng.random((N, 32)) < 0.1
    d[flip] ^= rng.integers(1, 256, (N, 32), dtype=np.uint8)[flip]
    return rng.permutation(d)


kfs = [keyframe(k) for k in range(8)]
rig = mcorb.Rig(2, 640, 480, 1, 1, nfeatures=N)
blk = mcorb.DescriptorBlock(2, rig.kcap)
blk.upload(0, kfs[0])
t_res, nm = [], 0
for k in range(1, REPS + 5):
    cur = kfs[k % len(kfs)]
    t0 = time.perf_counter()
    blk.upload(k % 2, cur)
    rig.match_sets(blk, [[(k - 1) % 2, k % 2]], dist_thresh=50.0, ratio=0.7)
    i1, i2 = rig.pairlist(0)
    if k >= 5:
        t_res.append(time.perf_counter() - t0)
    nm = len(i1)
ex = mcorb.ORBextractor(N, 1.2, 8, 20, 7)
t_host = []
for k in range(1, 25):
    a, b = kfs[(k - 1) % len(kfs)], kfs[k % len(kfs)]
    t0 = time.perf_counter()
    j1, j2 = ex.matchRatio(a, b, 50.0, 0.7)
    if k >= 5:
        t_host.append(time.perf_counter() - t0)
rig.close()
print(json.dumps({"n1_match_sets_ms": round(float(np.median(t_res)) * 1e3, 4), "n1_match_ratio_host_arrays_ms": round(float(np.median(t_host)) * 1e3, 4),
                  "descriptors": N, "accepted": int(nm), "dist_thresh": 50, "ratio": 0.7,
                  "note": "per keyframe: upload of the new set (96 KB) + knnMatch(k=2) + filter + read-back of the accepted pairs, through the "
                          "ctypes binding; host-array path: mcorb_match_ratio uploads both sets each call"}))
