"""Stage-by-stage GPU-vs-oracle comparison (development aid; the real tests live in tests/)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import mcorb
import oracle_lib as O

W, H, C, N = (int(a) for a in (sys.argv[1:5] + ["1280", "720", "4", "2000"][len(sys.argv) - 1:]))
imgs = [mcorb.synth_rig_frame(0, C, c, W, H) for c in range(C)]
rig = mcorb.Rig(C, W, H, 1, 1, nfeatures=N)
rig.upload(imgs)
t = time.time(); rig.extract(C); print("extract ms", (time.time() - t) * 1e3, rig.timing())
t = time.time(); rig.extract(C); print("extract ms (2nd)", (time.time() - t) * 1e3, rig.timing())
ex = O.OracleExtractor(N)
ok = True
descs = []
for c in range(C):
    mono, k, d = ex(imgs[c])
    for l in range(8):
        a = rig.level(c, l); b = ex.level(l)
        if not np.array_equal(a, b):
            ok = False; print("cam", c, "level", l, "pyramid mismatch", (a != b).sum())
        a = rig.level(c, l, blurred=True); b = ex.blurred(l)
        if b is not None and not np.array_equal(a, b):
            ok = False; print("cam", c, "level", l, "blur mismatch", (a != b).sum())
        gx, gy, gr = rig.candidates(c, l); ox, oy, orr = ex.candidates(l)
        if not (np.array_equal(gx, ox.astype(np.int32)) and np.array_equal(gy, oy.astype(np.int32)) and np.array_equal(gr, orr.astype(np.int32))):
            ok = False; print("cam", c, "level", l, "candidates mismatch", len(gx), len(ox))
    m2, k2, d2 = rig.features(c)
    descs.append(d)
    same_k = len(k) == len(k2) and all(np.array_equal(k[f], k2[f]) for f in k.dtype.names)
    same_d = d.shape == d2.shape and np.array_equal(d, d2)
    print("cam", c, "n", len(k), len(k2), "mono", mono, m2, "kps", same_k, "desc", same_d)
    ok &= same_k and same_d and mono == m2
t = time.time(); rig.match(1); print("match ms", (time.time() - t) * 1e3, rig.timing())
for i in range(C - 1):
    for j in range(i + 1, C):
        gi, gd = rig.pair_knn2(0, i, j); oi, od = O.knn2(descs[i], descs[j])
        s = np.array_equal(gi, oi) and np.array_equal(gd, od)
        g1, g2 = rig.pair_matches(0, i, j); o1, o2 = O.bruteforce_match(descs[i], descs[j])
        s2 = np.array_equal(g1, o1) and np.array_equal(g2, o2)
        print("pair", i, j, "knn2", s, "matches", s2, len(g1))
        ok &= s and s2
tr, mg = rig.tracks(0); otr, omg = O.intra_matches(descs)
print("tracks", tr.shape, otr.shape, np.array_equal(tr, otr), mg, omg)
ok &= np.array_equal(tr, otr) and mg == omg
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
