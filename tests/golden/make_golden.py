#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/mcorb_oracle.cpp).

The reference has no fixtures for this path and cannot be built or imported here
(C++ on OpenCV/DBoW2/glog, none installed), so these vectors are produced by the
restatement, not by the reference: they pin the oracle against regressions and give
the GPU tests a second, committed target.  Inputs are the deterministic synthetic
rig frames (mc-slam_amd/synth.py); only a checksum of each input is stored.
Run from the repository root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys
from importlib import import_module

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402

synth = import_module("mc-slam_amd.synth")

CASES = {
    # name: (ncams, w, h, nfeatures, nlevels, frame)
    "rig2_160x120_n300_l4": (2, 160, 120, 300, 4, 0),
    "cam1_640x480_n1000_l8": (1, 640, 480, 1000, 8, 2),
}


def build(name):
    ncams, w, h, nfeat, nlev, frame = CASES[name]
    out = {"meta": np.array([ncams, w, h, nfeat, nlev, frame], np.int32)}
    descs = []
    for c in range(ncams):
        img = synth.synth_rig_frame_numpy(frame, ncams, c, w, h)
        out["img_sha1_%d" % c] = np.frombuffer(hashlib.sha1(img.tobytes()).digest(), np.uint8)
        ex = O.OracleExtractor(nfeat, 1.2, nlev, 20, 7)
        mono, k, d = ex(img)
        assert mono >= 0
        out["mono_%d" % c] = np.array([mono], np.int32)
        out["kps_%d" % c] = k
        out["desc_%d" % c] = d
        out["ncand_%d" % c] = np.array([len(ex.candidates(l)[0]) for l in range(nlev)], np.int32)
        out["level_sha1_%d" % c] = np.stack([np.frombuffer(hashlib.sha1(ex.level(l).tobytes()).digest(), np.uint8)
                                             for l in range(nlev)])
        descs.append(d)
    if ncams > 1:
        idx, dist = O.knn2(descs[0], descs[1])
        out["knn_idx_01"], out["knn_dist_01"] = idx, dist
        i1, i2 = O.bruteforce_match(descs[0], descs[1])
        out["match_01"] = np.stack([i1, i2]).astype(np.int32)
        tr, mg = O.intra_matches(descs)
        out["tracks"], out["mergeable"] = tr, np.array([mg], np.int32)
    return out


if __name__ == "__main__":
    for name in CASES:
        data = build(name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print(name, {k: v.shape for k, v in data.items() if k.startswith(("kps", "tracks"))}, os.path.getsize(path), "bytes")
