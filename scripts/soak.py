"""Stability check: many create/destroy cycles (leaks, teardown races) and a long run of mixed synchronous / asynchronous
calls on several slots from two threads; results must stay identical throughout."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mcorb  # noqa: E402

C, W, H = 4, 640, 480
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2   # rig frames per job: 2 = 8 images (results through host-mapped memory), 3 = 12 (copied)
imgs = [mcorb.synth_rig_frame(1, C, c, W, H) for c in range(C)]


def signature(rig, slot, frame=0):
    tr, mg = rig.tracks(frame, slot=slot)
    return int(tr.sum()) * 31 + mg + sum(int(rig.features(frame * C + c, slot=slot)[2].sum()) for c in range(C))


t0 = time.time()
ref = None
for k in range(40):   # create / use / destroy
    rig = mcorb.Rig(C, W, H, max_frames=F, nslots=3, nfeatures=700)
    rig.upload(imgs * F, slot=k % 3)
    rig.process(F, slot=k % 3)
    s = signature(rig, k % 3), signature(rig, k % 3, F - 1)
    assert s[0] == s[1]
    ref = ref or s
    assert s == ref, (k, s, ref)
    rig.close()
print("create/destroy x40 ok, %.1f s" % (time.time() - t0))

rig = mcorb.Rig(C, W, H, max_frames=F, nslots=4, nfeatures=700)
for s in range(4):
    rig.upload(imgs * F, slot=s)
errors = []


def worker(slots, n):
    try:
        for k in range(n):
            for s in slots:
                if k % 2:
                    rig.process_submit(F, slot=s)
                else:
                    rig.process(F, slot=s)
            for s in slots:
                if k % 2:
                    rig.process_wait(slot=s)
                assert signature(rig, s) == ref[0], (k, s)
    except Exception as e:   # noqa: BLE001
        errors.append(e)


t0 = time.time()
th = [threading.Thread(target=worker, args=(sl, 150)) for sl in ((0, 1), (2, 3))]
for t in th:
    t.start()
for t in th:
    t.join()
assert not errors, errors
print("2 threads x 150 rounds x 2 slots (sync + async mixed) ok, %.1f s" % (time.time() - t0))
rig.close()

# two rigs of different sizes, each driven by its own thread at the same time (separate worker pools, separate graphs)
def rig_worker(w, h, n, out):
    try:
        r = mcorb.Rig(C, w, h, max_frames=1, nslots=1, nfeatures=600)
        im = [mcorb.synth_rig_frame(3, C, c, w, h) for c in range(C)]
        first = None
        for k in range(n):
            r.upload(im)
            r.process(1)
            tr, mg = r.tracks(0)
            s_ = int(tr.sum()) * 31 + mg + sum(int(r.features(c)[2].sum()) for c in range(C))
            first = first if first is not None else s_
            assert s_ == first, (w, h, k)
        r.close()
    except Exception as e:   # noqa: BLE001
        out.append(e)


t0 = time.time()
errs = []
th = [threading.Thread(target=rig_worker, args=(w, h, 300, errs)) for w, h in ((640, 480), (752, 480), (512, 384))]
for t in th:
    t.start()
for t in th:
    t.join()
assert not errs, errs
print("3 rigs x 300 frames in parallel threads ok, %.1f s" % (time.time() - t0))
