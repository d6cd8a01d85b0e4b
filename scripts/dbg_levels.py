import sys, os, numpy as np
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import mcorb, oracle_lib as O
W, H = 640, 480
img = mcorb.synth_rig_frame(0, 2, 0, W, H)
rig = mcorb.Rig(1, W, H, 1, 1, nfeatures=1000)
rig.upload([img]); rig.process_submit(1); rig.process_wait()
ex = O.OracleExtractor(1000); ex(img)
for l in range(8):
    a, b = rig.level(0, l), ex.level(l)
    d = np.argwhere(a != b)
    print("level", l, a.shape, "diff px", len(d), (d[:5].tolist(), d[-3:].tolist()) if len(d) else "")
    if len(d):
        ys, xs = d[:, 0], d[:, 1]
        print("   rows", np.unique(ys)[:20], "... cols min/max", xs.min(), xs.max(), "col%16 hist", np.bincount(xs % 256 // 16, minlength=16))
        break
