"""Single rig frame latency, the way MC-SLAM calls the front-end (one frame at a time, nothing in flight):
host u8 images -> upload -> extract + intra-rig match -> keypoints/descriptors/tracks back on the host.
    python scripts/latency.py [--frames 200] [--cams 4] [--width 1280] [--height 720] [--nfeatures 2000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def measure(mcorb, C=4, W=1280, H=720, N=2000, frames=200, distinct=8):
    rig = mcorb.Rig(C, W, H, max_frames=1, nslots=1, nfeatures=N)
    sets = [[mcorb.synth_rig_frame(f, C, c, W, H) for c in range(C)] for f in range(distinct)]
    t_up, t_proc, t_get, t_all = [], [], [], []
    for k in range(frames + 10):
        imgs = sets[k % distinct]
        t0 = time.perf_counter()
        rig.upload(imgs)
        t1 = time.perf_counter()
        rig.process(1)
        t2 = time.perf_counter()
        feats = [rig.features(c) for c in range(C)]
        tr, _ = rig.tracks(0)
        t3 = time.perf_counter()
        if k >= 10:
            t_up.append(t1 - t0); t_proc.append(t2 - t1); t_get.append(t3 - t2); t_all.append(t3 - t0)
    tm = rig.timing()
    # the same with the phases separated: upload() only enqueues the DMA of the staged planes, so in the loop above the PCIe transfer
    # (3.7 MB for 4 x 720p) runs at the head of the extract + match interval.  Here the stream is drained behind the upload.
    s_up, s_proc = [], []
    for k in range(frames // 2 + 10):
        imgs = sets[k % distinct]
        t0 = time.perf_counter()
        rig.upload(imgs)
        rig.staging(0)            # (synchronises the slot's stream)
        t1 = time.perf_counter()
        rig.process(1)
        t2 = time.perf_counter()
        if k >= 10:
            s_up.append(t1 - t0); s_proc.append(t2 - t1)
    rejected = rig.early_reads_rejected()
    rig.close()
    ms = lambda a: round(float(np.median(a)) * 1e3, 4)
    return {"frames": frames, "upload_ms": ms(t_up), "extract_match_ms": ms(t_proc), "readback_ms": ms(t_get),
            "total_ms": ms(t_all), "total_p95_ms": round(float(np.percentile(t_all, 95)) * 1e3, 4),
            "separated": {"upload_incl_dma_ms": ms(s_up), "extract_match_ms": ms(s_proc),
                          "note": "stream drained behind the upload: the PCIe transfer is in the first figure, not the second"},
            "early_reads_rejected": rejected,
            "keypoints": [int(len(f[1])) for f in feats], "tracks": int(len(tr)),
            "last_timing_us": {k: round(float(v), 1) for k, v in tm.items()}}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--cams", type=int, default=4)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--nfeatures", type=int, default=2000)
    a = ap.parse_args()
    import mcorb
    print(json.dumps(measure(mcorb, a.cams, a.width, a.height, a.nfeatures, a.frames)))
