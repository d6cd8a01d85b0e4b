"""FrontEnd::obtainLfFeatures (N3) per rig frame on the slot's keypoints and tracks: host code (mask filtering, N-view DLT,
mono fill).  4-cam 1280x720 @2000, 32 frames in the slot.    python scripts/lf_rate.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mcorb  # noqa: E402

C, W, H, N, F = 4, 1280, 720, 2000, 32
rig = mcorb.Rig(C, W, H, max_frames=F, nslots=1, nfeatures=N)
rig.upload([mcorb.synth_rig_frame(f, C, c, W, H) for f in range(F) for c in range(C)])
rig.process(F)
K = np.array([[700.0, 0, W / 2], [0, 700.0, H / 2], [0, 0, 1]])
Ks = [K] * C
Rs = [np.eye(3)] * C
ts = [np.array([0.12 * c, 0.0, 0.0]) for c in range(C)]
seg = [np.zeros((H, W), np.float32) for _ in range(C)]
times, nf, nt = [], 0, 0
for rep in range(3):
    for f in range(F):
        tr, _ = rig.tracks(f)
        t0 = time.perf_counter()
        feats, n_intra, n_mono, _ = rig.obtain_lf_features(f, tr, Ks, Rs, ts, seg_masks=seg)
        times.append(time.perf_counter() - t0)
        nf, nt = len(feats), len(tr)
# all frames of the slot in one call (mcorb_rig_obtain_lf_features_frames)
trs = [rig.tracks(f)[0] for f in range(F)]
segs = seg * F
tb = []
for rep in range(5):
    t0 = time.perf_counter()
    res = rig.obtain_lf_features_frames(0, trs, Ks, Rs, ts, seg_masks=segs)
    tb.append((time.perf_counter() - t0) / F)
rig.close()
print(json.dumps({"obtain_lf_features_ms_per_rig_frame": round(float(np.median(times)) * 1e3, 3),
                  "obtain_lf_features_frames_ms_per_rig_frame": round(float(np.median(tb)) * 1e3, 4), "frames_per_call": F,
                  "tracks": int(nt), "features": int(nf),
                  "note": "single: median of 96 calls through the ctypes binding; frames: %d frames per call, median of 5 calls / %d "
                          "(both include marshalling the camera matrices, masks and the numpy copies of the results)" % (F, F)}))
