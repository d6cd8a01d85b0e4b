"""BASELINE configs[3]: 4-cam front-end + DBoW2 BoW-vector build, at the real vocabulary's size (k=10, L=6: 1.1 M nodes;
the vocabulary file itself is not part of the reference, so the tree is synthetic) -- timings of
transform() per camera and of the BoW-guided computeIntraMatches(matches, words_) per rig frame."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def full_vocabulary(k=10, L=6, seed=1):
    rng = np.random.default_rng(seed)
    n = sum(k ** d for d in range(1, L + 1))
    parent = np.zeros(n, np.int32)
    leaf = np.zeros(n, np.uint8)
    # breadth-first blocks of k children, node ids 1..n: children of node p (level d) are consecutive
    first = 1
    ids_prev = np.array([0])
    pos = 0
    for d in range(1, L + 1):
        cnt = k ** d
        parent[pos:pos + cnt] = np.repeat(ids_prev, k)
        if d == L:
            leaf[pos:pos + cnt] = 1
        ids_prev = np.arange(first, first + cnt)
        first += cnt
        pos += cnt
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    weight = np.where(leaf == 1, rng.uniform(0.1, 9.0, n), 0.0)
    return dict(k=k, L=L, scoring=0, weighting=0, parent=parent, is_leaf=leaf, desc=desc, weight=weight)


def main():
    import mcorb
    C, W, H, N, F = 4, 1280, 720, 2000, 32
    t0 = time.perf_counter()
    v = full_vocabulary()
    voc = mcorb.ORBVocabulary().create(**v)
    t_build = time.perf_counter() - t0
    rig = mcorb.Rig(C, W, H, F, 1, nfeatures=N)
    rig.upload([mcorb.synth_rig_frame(f, C, c, W, H) for f in range(F) for c in range(C)])
    rig.extract(F * C)
    L = voc.L_
    tt, tm, bt, bm = [], [], [], []
    for it in range(20):
        # one frame at a time (how a per-frame caller drives it)
        t0 = time.perf_counter()
        for c in range(C):
            voc.transform_rig_image(rig, c, levelsup=4)
        t1 = time.perf_counter()
        tr, nr, words = voc.match_rig_frame(rig, 0, levelsup=4)
        t2 = time.perf_counter()
        tt.append(t1 - t0); tm.append(t2 - t1)
        # all 32 frames of the slot per call (the C entry points only: reading the results back through ctypes is the
        # caller's cost and is not part of the figure)
        t0 = time.perf_counter()
        rc = L.mcorb_rig_transform_images(rig.h_rig, 0, 0, F * C, voc.h, 4)
        t1 = time.perf_counter()
        rc |= L.mcorb_rig_match_bow_frames(rig.h_rig, 0, 0, F, voc.h, 4, 0.85, None)
        t2 = time.perf_counter()
        assert rc == 0
        bt.append((t1 - t0) / F); bm.append((t2 - t1) / F)
    out = {"vocabulary": "synthetic k=10 L=6, %d nodes" % len(v["parent"]), "vocab_upload_s": round(t_build, 2),
           "transform_ms_per_rig_frame": round(float(np.median(tt[5:])) * 1e3, 3),
           "bow_guided_match_ms_per_rig_frame": round(float(np.median(tm[5:])) * 1e3, 3),
           "batched_32_frames": {"transform_ms_per_rig_frame": round(float(np.median(bt[5:])) * 1e3, 4),
                                 "bow_guided_match_ms_per_rig_frame": round(float(np.median(bm[5:])) * 1e3, 4)},
           "tracks": int(len(tr)), "keypoints_per_camera": int(len(rig.features(0)[1]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
