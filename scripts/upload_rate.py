"""PCIe-inclusive rate: upload + process per batch (DESIGN.md / profiles/README.md)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import mcorb
W, H, C, F, S = 1280, 720, 4, 8, 6
rig = mcorb.Rig(C, W, H, max_frames=F, nslots=S, nfeatures=2000)
batches = [[mcorb.synth_rig_frame(s * F + f, C, c, W, H) for f in range(F) for c in range(C)] for s in range(S)]
for s in range(S):
    rig.upload(batches[s], slot=s); rig.process_submit(F, slot=s)
for s in range(S):
    rig.process_wait(slot=s)
steps = 20
t0 = time.perf_counter()
for s in range(S):
    rig.upload(batches[s], slot=s); rig.process_submit(F, slot=s)
for j in range(steps * S):
    s = j % S
    rig.process_wait(slot=s)
    if j + S < steps * S:
        rig.upload(batches[s], slot=s); rig.process_submit(F, slot=s)
dt = time.perf_counter() - t0
print("upload+process: %.1f frames/s (%.3f ms/frame), %d frames" % (steps * S * F / dt, dt / (steps * S * F) * 1e3, steps * S * F))

# zero-copy hand-off: the "reader" writes straight into the pinned staging planes (here: a numpy copy stands for the decode)
def put(s):
    for m, im in enumerate(batches[s]):
        rig.staging(m, slot=s)[:] = im
    rig.upload_staged(F * C, slot=s)
t0 = time.perf_counter()
for s in range(S):
    put(s); rig.process_submit(F, slot=s)
for j in range(steps * S):
    s = j % S
    rig.process_wait(slot=s)
    if j + S < steps * S:
        put(s); rig.process_submit(F, slot=s)
dt = time.perf_counter() - t0
print("decode-into-staging+process: %.1f frames/s (%.3f ms/frame)" % (steps * S * F / dt, dt / (steps * S * F) * 1e3))
t0 = time.perf_counter()
for s in range(S):
    rig.upload_staged(F * C, slot=s); rig.process_submit(F, slot=s)
for j in range(steps * S):
    s = j % S
    rig.process_wait(slot=s)
    if j + S < steps * S:
        rig.upload_staged(F * C, slot=s); rig.process_submit(F, slot=s)
dt = time.perf_counter() - t0
print("DMA-from-staging+process (PCIe only, no host copy): %.1f frames/s (%.3f ms/frame)" % (steps * S * F / dt, dt / (steps * S * F) * 1e3))
