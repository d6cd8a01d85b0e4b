#!/usr/bin/env python3
"""bench.py -- multi-camera ORB front-end throughput on MI355X.

Workload (BASELINE.json configs[1]): 4-camera rig, 1280x720, 2000 keypoints/camera,
extract + all-pairs intra-rig match.  One *step* = one batch of F synthetic rig
frames per rank through the whole hot path (pyramid -> FAST -> selection -> blur ->
BRIEF -> k-NN(k=2) + ratio filter -> track merge), inputs resident in HBM before the
timed region.  `value` = rig frames / second over all ranks.

N = 1: cameras and pairs all on one GPU, S slots in flight (the host selection stage
of one sub-batch overlaps the GPU phases of the others).
N > 1 (weak scaling, F frames per rank per step, F*N frames per step in total):
camera c of frame f is extracted on rank (c + f) mod N; per-camera descriptors are
exchanged with ONE RCCL all-gather per step; frame f is matched on rank f mod N.

Extra legs on rank 0 at N = 1: `cpu_baseline` (the CPU oracle, one thread per camera
as the reference's extractFeaturesParallel does, on a bounded sample) and a bit-exact
GPU-vs-oracle check of the first frame.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

W, H, NCAMS, NFEAT = 1280, 720, 4, 2000
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves


def level_pixels(rig):
    return [rig.level_size(l) for l in range(rig.nlevels)]


def algorithmic_bytes(kernel, S, S0, s_last, K, Kc, npairs_per_frame=6):
    """Per camera image (or per frame for knn2), SURVEY.md 8(d) with this build's record sizes."""
    if kernel == "k_fast_cells":      # read every level once + packed 4-byte candidate records out
        return S + 4 * Kc
    if kernel == "k_resize":          # read levels 0..n-2, write levels 1..n-1
        return (S - s_last) + (S - S0)
    if kernel == "k_blur":
        return 2 * S
    if kernel == "k_describe":
        return K * (512 + 32)
    if kernel == "k_knn2":            # per pair: both descriptor sets once + 8-byte partial per query
        return 2 * K * 32 + K * 16
    raise KeyError(kernel)


def cpu_baseline(frames, ncams):
    """Oracle timed the way the reference runs: one thread per camera for extraction
    (MultiCameraFrame.cpp:212-227), matching + track merge on the calling thread."""
    import mcorb
    import oracle_lib as O
    exs = [O.OracleExtractor(NFEAT) for _ in range(ncams)]
    times, t_extract, results = [], [], []
    for f in frames:
        imgs = [mcorb.synth_rig_frame(f, ncams, c, W, H) for c in range(ncams)]
        res = [None] * ncams
        t0 = time.perf_counter()

        def work(c):
            res[c] = exs[c](imgs[c])
        ths = [threading.Thread(target=work, args=(c,)) for c in range(ncams)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        t1 = time.perf_counter()
        tracks, _ = O.intra_matches([r[2] for r in res])
        times.append(time.perf_counter() - t0)
        t_extract.append(t1 - t0)
        results.append((res, tracks))
    return times, t_extract, results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=None, help="rig frames per rank per step (default 192; 128 on the N>1 path)")
    ap.add_argument("--slots", type=int, default=None, help="buffer sets in flight per rank (default 6; 8 = 4 groups x 2 on the N>1 path)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-frames", type=int, default=48, help="rig frames timed on the CPU oracle (about 12 s of CPU work on 4 threads)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-frame latency leg (use when profiling kernels)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL); 'gloo' only to rehearse N>1 on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses device 0")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: take the N>1 code path (export -> RCCL all-gather -> external match) even at N=1")
    args = ap.parse_args()

    import torch
    import mcorb

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if world != N and not (N == 1 and world == 1):
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (N, world))
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    gloo = args.dist_backend == "gloo"
    DIST = N > 1 or args.force_dist     # use the sharded path (export -> all-gather -> external match)
    if DIST:
        import torch.distributed as dist
        if args.force_dist and "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29555")
        if gloo:
            dist.init_process_group("gloo")
        else:
            # RCCL prints a version banner on STDOUT when the communicator is created; stdout is reserved for the one JSON
            # line, so the banner is sent to stderr (fd-level redirect around init + the first collective)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
                warm = torch.zeros(1, device="cuda")
                dist.all_reduce(warm)
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)

    F = args.frames if args.frames else (128 if DIST else 192)   # N = 1: 6 slots x 32 rig frames = 128 images per launch
    S = args.slots if args.slots else (8 if DIST else 6)
    S = max(1, min(S, F))
    G = int(os.environ.get("MCORB_BENCH_GROUPS", "4")) if DIST else 1   # slot groups: with N > 1 G-1 groups extract steps k+1.. while one matches step k
    if DIST and S % G:
        S += G - S % G
    SG = S // G                        # slots per group
    fps = F // SG                      # rig frames per slot job
    if fps * SG != F:
        raise SystemExit("--frames must be a multiple of the slots per group (%d)" % SG)
    if world > 1:
        # several ranks share the node's CPUs: let the slot driver threads sleep on their HIP events instead of spinning
        # (2-4 % slower on a single GPU, but N x 8 spinning threads would eat the cores the selection workers need)
        os.environ.setdefault("MCORB_SYNC", "block")
    rig = mcorb.Rig(NCAMS, W, H, max_frames=fps, nslots=S, nfeatures=NFEAT, device_id=local)
    kcap = rig.kcap
    total_frames = F * N

    # ---- inputs: synthetic rig frames, staged into HBM before the timed region ----
    from importlib import import_module
    shard = import_module("mc-slam_amd.sharding")
    if not DIST:
        mine = [(f, c) for f in range(F) for c in range(NCAMS)]
    else:
        mine = shard.images_of_rank(rank, N, NCAMS, total_frames)
    assert len(mine) == F * NCAMS, (len(mine), F * NCAMS)
    per_slot = fps * NCAMS
    for s in range(S):
        i = s % SG
        imgs = [mcorb.synth_rig_frame(f, NCAMS, c, W, H) for (f, c) in mine[i * per_slot:(i + 1) * per_slot]]
        rig.upload(imgs, slot=s)

    if DIST:
        local_desc = [torch.zeros((F * NCAMS, kcap, 32), dtype=torch.uint8, device="cuda") for _ in range(G)]
        all_desc = [torch.zeros((N * F * NCAMS, kcap, 32), dtype=torch.uint8, device="cuda") for _ in range(G)]
        local_cnt = [torch.zeros(F * NCAMS, dtype=torch.int32, device="cuda") for _ in range(G)]
        all_cnt = [torch.zeros(N * F * NCAMS, dtype=torch.int32, device="cuda") for _ in range(G)]
        my_frames, sets = shard.match_sets(rank, N, NCAMS, total_frames)   # gathered set index of (f, c)
        assert len(my_frames) == F
        torch.cuda.synchronize()   # the zero fills above ran on torch's stream; the engine writes these tensors from its own

    def extract_submit(g):
        for i in range(SG):
            rig.extract_submit(per_slot, slot=g * SG + i)

    def exchange_and_match(g):
        """Waits for group g's extraction, exchanges the descriptors and queues the matching; match_finish(g) collects it."""
        cnt_host = np.zeros(F * NCAMS, np.int32)
        for i in range(SG):
            s = g * SG + i
            rig.extract_wait(slot=s)
            cnt_host[i * per_slot:(i + 1) * per_slot] = rig.export_descriptors(local_desc[g][i * per_slot].data_ptr(),
                                                                               per_slot, slot=s)
        local_cnt[g].copy_(torch.from_numpy(cnt_host))
        if gloo:                                                # rehearsal path only
            hd, hc = torch.zeros(all_desc[g].shape, dtype=torch.uint8), torch.zeros(all_cnt[g].shape, dtype=torch.int32)
            dist.all_gather_into_tensor(hd, local_desc[g].cpu())
            dist.all_gather_into_tensor(hc, local_cnt[g].cpu())
            all_desc[g].copy_(hd)
            all_cnt[g].copy_(hc)
        else:
            dist.all_gather_into_tensor(all_desc[g], local_desc[g])   # the exchange step (RCCL over xGMI)
            dist.all_gather_into_tensor(all_cnt[g], local_cnt[g])
        counts = all_cnt[g].cpu().numpy()                       # syncs the collective
        for i in range(SG):                                     # this rank's frames, split over the group's slots
            rig.match_external_submit(all_desc[g].data_ptr(), counts, sets[i * fps:(i + 1) * fps], slot=g * SG + i)

    def match_finish(g):
        for i in range(SG):
            rig.match_wait(slot=g * SG + i)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # per-kernel durations, accumulated from the HIP events the engine records on each slot's stream
    ksum = {"k_resize": 0.0, "k_fast_cells": 0.0, "k_blur": 0.0, "k_describe": 0.0, "k_knn2": 0.0,
            "side:k_compact": 0.0, "select_host": 0.0}

    def account(s):
        t = rig.timing(slot=s)
        ksum["k_resize"] += t["pyramid_us"]
        ksum["k_fast_cells"] += t["fast_us"]
        ksum["side:k_compact"] += t["compact_us"]
        ksum["k_blur"] += t["blur_us"]
        ksum["k_describe"] += t["describe_us"]
        ksum["k_knn2"] += t["knn2_us"]
        ksum["select_host"] += t["select_us"]

    def run_steps(nsteps, timed):
        """nsteps steps = nsteps*F frames per rank.
        N == 1: rolling submission, S jobs always in flight (a slot is resubmitted as soon as its previous job
        is collected), so step boundaries do not drain the pipeline.
        N > 1: G slot groups rotate; extraction of steps k+1 .. k+G-2 is in flight while step k's descriptors
        are all-gathered and matched."""
        if DIST:
            # step k runs on group k % G.  While step k's descriptors are exchanged, step k-1 is being matched and steps
            # k+1 .. k+G-2 are being extracted; a group is re-armed (next extraction) as soon as its matching is collected.
            def collect(g):
                match_finish(g)
                if timed:
                    for i in range(SG):
                        account(g * SG + i)
            if G == 1:
                for k in range(nsteps):
                    extract_submit(0)
                    exchange_and_match(0)
                    collect(0)
                return
            for k in range(min(G - 1, nsteps)):
                extract_submit(k % G)
            for k in range(nsteps):
                exchange_and_match(k % G)
                if k >= 1:
                    collect((k - 1) % G)
                if k + G - 1 < nsteps:
                    extract_submit((k + G - 1) % G)     # idle at k = 0, just collected afterwards
            if nsteps >= 1:
                collect((nsteps - 1) % G)
            return
        jobs = nsteps * S
        for s in range(min(S, jobs)):
            rig.process_submit(fps, slot=s)
        for j in range(jobs):
            s = j % S
            rig.process_wait(slot=s)
            if timed:
                account(s)
            if j + S < jobs:
                rig.process_submit(fps, slot=s)

    run_steps(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if gloo else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = total_frames * args.steps / dt
    launches = args.steps * SG
    # workload constants for the algorithmic-byte formulas
    lv = level_pixels(rig)
    Spx = sum(w * h for w, h in lv)
    S0, s_last = lv[0][0] * lv[0][1], lv[-1][0] * lv[-1][1]
    nimg_launch = per_slot
    Kc = np.mean([sum(len(rig.candidates(m, l, slot=0)[0]) for l in range(rig.nlevels)) for m in range(min(4, per_slot))])
    K = np.mean([rig.features(m, slot=0)[1].shape[0] for m in range(min(4, per_slot))])
    # Which kernel dominates is decided on ISOLATED durations (one job at a time, nothing else on the GPU): with six
    # jobs in flight every kernel's wall duration is stretched by whatever shares the GPU with it, and the ranking of
    # two close kernels flips from run to run.  The roofline figures themselves use the timed region, as specified.
    iso = None
    if not DIST:
        iso = {k: 0.0 for k in ksum}
        ISO_JOBS = 8
        for _ in range(ISO_JOBS):
            rig.process_submit(fps, slot=0)
            rig.process_wait(slot=0)
            t = rig.timing(slot=0)
            for k, f in (("k_resize", "pyramid_us"), ("k_fast_cells", "fast_us"), ("side:k_compact", "compact_us"), ("k_blur", "blur_us"),
                         ("k_describe", "describe_us"), ("k_knn2", "knn2_us"), ("select_host", "select_us")):
                iso[k] += t[f] / ISO_JOBS
    gpu_kernels = {k: v for k, v in ksum.items() if k.startswith("k_")}
    rank = {k: v for k, v in iso.items() if k.startswith("k_")} if iso else gpu_kernels
    dominant = max(rank, key=rank.get)
    avg_us = gpu_kernels[dominant] / launches
    if dominant == "k_knn2":
        unit_bytes, units = algorithmic_bytes(dominant, Spx, S0, s_last, K, Kc), 6 * fps
    else:
        unit_bytes, units = algorithmic_bytes(dominant, Spx, S0, s_last, K, Kc), nimg_launch
    achieved = unit_bytes * units / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch, if measured
    if os.path.exists(tpath):
        try:
            traffic = int(json.load(open(tpath)).get(dominant, {}).get("bytes_per_image") * nimg_launch)
        except Exception:
            traffic = None
    out = {
        "metric": "multi-cam frames/sec (4-cam 1280x720 @2000 kpts/cam, extract + intra-rig match)",
        "value": round(value, 2), "unit": "frames/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_frame": round(dt / args.steps / total_frames * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "4-cam rig 1280x720, 2000 kpts/cam, extract + all-pairs intra-rig match (configs[1])",
                   "frames_per_rank_per_step": F, "slots": S, "frames_per_launch": fps, "cameras": NCAMS, "nfeatures": NFEAT,
                   "sharding": "single GPU" if not DIST else "camera (c+f) mod N for extraction, RCCL all-gather of descriptors, frame f mod N for matching"},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "avg_launch_us": round(avg_us, 2), "algorithmic_bytes_per_launch": int(unit_bytes * units),
                     "images_per_launch": nimg_launch, "fast_candidates_per_image": int(Kc),
                     "note": "avg_launch_us is the HIP-event average over the timed region, where %d jobs share the GPU; "
                             "isolated_* is the same kernel with one job in flight" % S},
        "kernel_us_per_step": {k: round(v / args.steps, 2) for k, v in ksum.items()},
    }
    if iso:
        ia = unit_bytes * units / (iso[dominant] * 1e-6) / 1e9
        out["roofline"].update({"isolated_launch_us": round(iso[dominant], 2), "isolated_achieved": round(ia, 2),
                                "isolated_frac": round(ia / HBM_PEAK_GBS, 5)})
        out["kernel_us_per_launch_isolated"] = {k: round(v, 2) for k, v in iso.items()}

    if N == 1 and not DIST and not args.no_cpu:
        ncpu = os.cpu_count()
        times, t_ext, results = cpu_baseline(list(range(2 + args.cpu_frames)), NCAMS)
        times, t_ext = np.array(times[2:]), np.array(t_ext[2:])
        out["cpu_baseline"] = {"value": round(1.0 / float(np.median(times)), 3), "unit": "frames/s", "cores": NCAMS,
                               "kind": "port",
                               "sample": "%d rig frames (4-cam 1280x720 @2000) after 2 warm-ups, CPU oracle, one thread per "
                                         "camera for extraction + matching on the caller thread; median %.1f ms (extract %.1f "
                                         "+ match %.1f), p95 %.1f ms; host has %d logical cores"
                                         % (len(times), np.median(times) * 1e3, np.median(t_ext) * 1e3,
                                            np.median(times - t_ext) * 1e3, np.percentile(times, 95) * 1e3, ncpu)}
        # bit-exact check of every CPU-timed frame against what the GPU slots hold (slot s holds rig frames s*fps .. (s+1)*fps-1)
        ok, checked = True, 0
        for f, (res, tracks) in enumerate(results):
            slot, fi = f // fps, f % fps
            if slot >= S:
                break
            for c in range(NCAMS):
                mono, k, d = res[c]
                m2, k2, d2 = rig.features(fi * NCAMS + c, slot=slot)
                ok &= mono == m2 and len(k) == len(k2) and all(np.array_equal(k[n], k2[n]) for n in k.dtype.names) \
                    and np.array_equal(d, d2)
            tr, _ = rig.tracks(fi, slot=slot)
            ok &= np.array_equal(tr, tracks)
            checked += 1
        out["gpu_equals_oracle_frames_checked"] = checked
        out["gpu_equals_oracle_frame0"] = bool(ok)   # (kept under its round-1 name) true only if ALL checked frames are bit-identical
    if N == 1 and not DIST and not args.no_latency:
        # one rig frame at a time, nothing in flight (how MC-SLAM's tracking loop calls the front-end): host u8 images in,
        # keypoints / descriptors / tracks back on the host.  Not part of `value`.
        rig.close()
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from latency import measure
        lat = measure(mcorb, NCAMS, W, H, NFEAT, frames=100)
        out["single_frame_latency_ms"] = {k: lat[k] for k in ("upload_ms", "extract_match_ms", "readback_ms", "total_ms", "total_p95_ms")}
    print(json.dumps(out), flush=True)
    rig.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
