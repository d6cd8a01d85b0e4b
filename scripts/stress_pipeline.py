"""The batch pipeline under load, checked: S slots of F rig frames each in rolling submission for many rounds; after every job the
slot's results (keypoint counts, descriptor bytes, tracks of every frame) must equal what the same slot produced in its first round,
and frame 0 of slot 0 must equal the oracle's.  usage: python3 scripts/stress_pipeline.py [rounds] [slots] [frames per job] [gpu_jobs]"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import mcorb  # noqa: E402
import oracle_lib as O  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
F = int(sys.argv[3]) if len(sys.argv) > 3 else 16
GJ = int(sys.argv[4]) if len(sys.argv) > 4 else 0
C, W, H, N = 4, 640, 480, 1000
rig = mcorb.Rig(C, W, H, max_frames=F, nslots=S, nfeatures=N, gpu_jobs=GJ)
imgs = {s: [mcorb.synth_rig_frame(100 * s + f, C, c, W, H) for f in range(F) for c in range(C)] for s in range(S)}
for s in range(S):
    rig.upload(imgs[s], slot=s)


def signature(s):
    sig = 0
    for f in range(F):
        tr, mg = rig.tracks(f, slot=s)
        sig = zlib.crc32(np.ascontiguousarray(tr).tobytes(), sig) + mg
        for c in range(C):
            m, k, d = rig.features(f * C + c, slot=s)
            sig = zlib.crc32(np.ascontiguousarray(d).tobytes(), zlib.crc32(np.ascontiguousarray(k["x"]).tobytes(), sig)) + m
    return sig


ref = {}
jobs = rounds * S
for s in range(S):
    rig.process_submit(F, slot=s)
bad = 0
for j in range(jobs):
    s = j % S
    rig.process_wait(slot=s)
    g = signature(s)
    if s not in ref:
        ref[s] = g
    elif g != ref[s]:
        bad += 1
        print("round %d slot %d: results differ from the slot's first round" % (j // S, s), flush=True)
    if j + S < jobs:
        rig.process_submit(F, slot=s)
    if j % (50 * S) == 0:
        print("round", j // S, "bad", bad, flush=True)
# frame 0 of slot 0 against the oracle
descs = []
for c in range(C):
    o = O.OracleExtractor(N)(imgs[0][c])
    m, k, d = rig.features(c, slot=0)
    assert o[0] == m and np.array_equal(o[2], d) and np.array_equal(o[1]["x"], k["x"]), "slot 0 frame 0 cam %d differs from the oracle" % c
    descs.append(d)
otr, omg = O.intra_matches(descs)
tr, mg = rig.tracks(0, slot=0)
assert np.array_equal(tr, otr) and mg == omg
print("pipeline stress: %d jobs of %d images on %d slots (gpu_jobs %d), %d bad; slot 0 frame 0 equals the oracle" % (jobs, F * C, S, GJ, bad))
rig.close()
sys.exit(1 if bad else 0)
