#!/usr/bin/env python3
"""bench.py -- multi-camera ORB front-end throughput on MI355X.

Workload (`--config`, default 720p4 = BASELINE.json configs[1]): 4-camera rig, 1280x720, 2000 keypoints/camera,
extract + all-pairs intra-rig match; 1080p8 = configs[2]: 8 cameras, 1920x1080 (one camera per GPU at --gpus 8).
One *step* = one batch of F synthetic rig frames per rank through the whole hot path (pyramid -> FAST -> selection ->
blur -> BRIEF -> k-NN(k=2) + ratio filter -> track merge), inputs resident in HBM before the timed region.
`value` = rig frames / second over all ranks: the median of `--repeats` timed regions of exactly `--steps` steps each
(`value_min` / `value_max` give the spread).

N = 1: cameras and pairs all on one GPU, S slots in flight (the host selection stage of one sub-batch overlaps the GPU
phases of the others).
N > 1 (weak scaling, F frames per rank per step, F*N frames per step in total): camera c of frame f is extracted on
rank (c + f) mod N; every descriptor set travels to the rank that matches its frame (f mod N) with ONE RCCL all-to-all
per step (uneven splits, mc-slam_amd/sharding.py) -- stream-ordered: export -> collective -> match without a host sync.

Extra legs on rank 0 at N = 1 (none of them part of `value`):
  value_with_staging / value_with_upload_u8 / pcie_gbs   the same steps with the PCIe hand-off in front of every batch
  cpu_baseline                                           the CPU oracle on a bounded sample, every frame checked bit-exact
  single_frame_latency_ms                                one rig frame at a time, nothing in flight
"""
import argparse
import hashlib
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

CONFIGS = {
    "720p4": dict(W=1280, H=720, NCAMS=4, NFEAT=2000,
                  label="4-cam rig 1280x720, 2000 kpts/cam, extract + all-pairs intra-rig match (configs[1])"),
    "1080p8": dict(W=1920, H=1080, NCAMS=8, NFEAT=2000,
                   label="8-cam rig 1920x1080, 2000 kpts/cam, extract + all-pairs intra-rig match (configs[2])"),
}
HOST_CORES_EXTRA_SLOTS = 2  # the `value_host_cores` leg runs with this many slots more than jobs admitted to the GPU at once
IMAGES_PER_LAUNCH = 512     # camera images per slot job (128 four-camera or 64 eight-camera rig frames): with the selection on the GPU the
                            # job has a latency-bound middle (k_compact, k_select, k_assemble: ~90 us whatever the batch) that larger
                            # batches amortise -- 128 images x 5 slots 37.5 k, 256 x 4 39.1 k, 512 x 4 40.0 k (profiles/r04_overlap.txt)
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves
GPU_KERNELS = ("k_resize", "k_fast_cells", "k_compact", "k_describe_fused", "k_knn2")   # k_blur: only with MCORB_BLUR_PLANES / orientation
TIMING_FIELD = {"k_resize": "pyramid_us", "k_fast_cells": "fast_us", "k_compact": "compact_us",
                "k_describe_fused": "describe_us", "k_knn2": "knn2_us", "select_host": "select_us"}


def algorithmic_bytes(kernel, S, S0, s_last, K, Kc, buckets):
    """Per camera image (per camera pair for k_knn2), SURVEY.md 8(d) with this build's record sizes."""
    if kernel == "k_fast_cells":      # read every level once + packed 4-byte candidate records out
        return S + 4 * Kc
    if kernel == "k_resize":          # read levels 0..n-2, write levels 1..n-1
        return (S - s_last) + (S - S0)
    if kernel == "k_blur":
        return 2 * S
    if kernel == "k_describe_fused":  # the 33 x 33 source window of each keypoint in, 32 bytes out
        return K * (33 * 33 + 32)
    if kernel == "k_knn2":            # per pair: both descriptor sets once + 8-byte partial per query
        return 2 * K * 32 + K * 16
    if kernel == "k_compact":         # candidates in, sorted candidates out, bucket starts + per-bucket winners out
        return 2 * 4 * Kc + 16 * buckets
    raise KeyError(kernel)


def cpu_baseline(cfg, frames, threads):
    """Oracle timed the way the reference runs: one thread per camera for extraction
    (MultiCameraFrame.cpp:212-227), matching + track merge on the calling thread; threads=1: everything on one thread."""
    import mcorb
    import oracle_lib as O
    W, H, C, NFEAT = cfg["W"], cfg["H"], cfg["NCAMS"], cfg["NFEAT"]
    exs = [O.OracleExtractor(NFEAT) for _ in range(C)]
    times, t_extract, results = [], [], []
    for f in frames:
        imgs = [mcorb.synth_rig_frame(f, C, c, W, H) for c in range(C)]
        res = [None] * C
        t0 = time.perf_counter()

        def work(c):
            res[c] = exs[c](imgs[c])
        if threads == 1:
            for c in range(C):
                work(c)
        else:
            ths = [threading.Thread(target=work, args=(c,)) for c in range(C)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        t1 = time.perf_counter()
        tracks, _ = O.intra_matches([r[2] for r in res])
        times.append(time.perf_counter() - t0)
        t_extract.append(t1 - t0)
        results.append((res, tracks))
    return np.array(times), np.array(t_extract), results


def free_port():
    """a TCP port nobody listens on right now (bind to 0, read it back): two sessions on one box do not collide on a fixed one"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` typed without a launcher: this process -- before it has touched torch or the GPU, and without
    exec -- starts the N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, same arguments), relays
    rank 0's JSON line and exits non-zero if any rank does."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()))
    reader.start()
    rcs = [None] * n
    failed_at = None
    while any(rc is None for rc in rcs):
        for r, pr in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = pr.poll()
        if failed_at is None and any(rc not in (None, 0) for rc in rcs):
            failed_at = time.time()                      # a rank died: the others are stuck in a collective
        if failed_at is not None and time.time() - failed_at > 20:
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    pr.kill()                            # (exactly the processes this function started)
        time.sleep(0.2)
    reader.join()
    for line in out0:
        sys.stdout.write(line)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed: %s\n" % bad)
        return 1
    return 0


def child_value(extra, timeout=600):
    """one more bench line from a fresh child process (own HIP / RCCL state); returns its parsed JSON or {"error": ...}"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu", "--no-latency", "--no-staging", "--host-cores", "0", "--no-extra-legs"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "timeout"}
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    if pr.returncode != 0 or not lines:
        return {"error": "rc %d: %s" % (pr.returncode, pr.stderr[-300:])}
    return json.loads(lines[-1])


def kernels_sha():
    return hashlib.sha256(open(os.path.join(ROOT, "mc-slam_amd", "csrc", "mcorb_kernels.hip"), "rb").read()).hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="720p4")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions; value = their median")
    ap.add_argument("--min-region-s", type=float, default=1.0,
                    help="a timed region runs at least this long: the steps per region are raised above --steps when needed "
                         "(20 steps last 0.11 s -- a 3 %% kernel change is invisible through that); `steps` in the output is the "
                         "number actually timed per region, `steps_requested` what was asked for.  0 = exactly --steps")
    ap.add_argument("--host-cores", type=int, default=4,
                    help="extra leg `value_host_cores`: the same steps with the process (engine workers, slot drivers, caller) "
                         "confined to this many cores; 0 = skip")
    ap.add_argument("--frames", type=int, default=None, help="rig frames per rank per step (default: slots x 128 images / cameras)")
    ap.add_argument("--slots", type=int, default=None, help="buffer sets in flight per rank (default 6; 12 = 6 groups x 2 on the N>1 path)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-frames", type=int, default=64, help="rig frames timed on the CPU oracle (one thread per camera)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-frame latency leg (use when profiling kernels)")
    ap.add_argument("--no-staging", action="store_true", help="skip the PCIe-inclusive legs")
    ap.add_argument("--iso-jobs", type=int, default=12, help="jobs run alone on the GPU for the isolated kernel durations")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL); 'gloo' only to rehearse N>1 on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses device 0")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: take the N>1 code path (export -> RCCL all-to-all -> external match) even at N=1")
    ap.add_argument("--exchange", choices=["a2a", "allgather"], default="a2a",
                    help="N > 1: a2a = one all-to-all per round, every set goes to the rank that matches its frame (throughput default); "
                         "allgather = every rank receives every set (what SURVEY 8e / north_star describe), used with --partition pairs")
    ap.add_argument("--partition", choices=["frames", "pairs"], default="frames",
                    help="N > 1: frames = frame f matched on rank f mod N; pairs = camera pair (i, j) of frame f matched on rank "
                         "(i + j + f) mod N, the accepted lists gathered on rank 0, which runs the serial track merge")
    ap.add_argument("--dump-tracks", default=None, help="(tests) write the tracks of this rank's first frames to an .npz")
    ap.add_argument("--graph", type=int, default=None,
                    help="mcorb_rig_set_graph: 0 = every job launch by launch, 1 = every job replayed from its HIP graph (no per-kernel "
                         "events: the roofline figures fall back to the isolated durations), K > 1 = all but every K-th job of a slot, "
                         "which is the timed sample (default: the engine's)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the two child-process legs of the N = 1 line: value_force_dist (the sharded path at world size 1, the "
                         "like-for-like denominator of value(N) / (N value(1))) and configs2_1080p8 (BASELINE configs[2] on one GPU)")
    ap.add_argument("--spawn-selftest", type=int, default=None, metavar="FAIL_RANK",
                    help="(tests, no GPU) the ranks only rendezvous over gloo, all-reduce their rank numbers and rank 0 prints "
                         "what it saw; rank FAIL_RANK (>= 0) exits with status 3 instead")
    args = ap.parse_args()
    if args.spawn_selftest is not None and "RANK" in os.environ:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        if rank == args.spawn_selftest:
            sys.exit(3)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1.0])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "sum": float(t.item()), "n_gpus": args.gpus}), flush=True)
        dist.destroy_process_group()
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        # typed the way the driver types `--gpus 1`: no launcher around it.  Start the ranks from here (no GPU call, no torch
        # import, no exec before this point) and relay rank 0's line.
        sys.exit(spawn_ranks(args.gpus))
    cfg = CONFIGS[args.config]
    W, H, NCAMS, NFEAT = cfg["W"], cfg["H"], cfg["NCAMS"], cfg["NFEAT"]

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
    DIST = N > 1 or args.force_dist     # use the sharded path (export -> all-to-all -> external match)
    if DIST:
        import torch.distributed as dist
        if args.force_dist and "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
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

    PAIRS = DIST and args.partition == "pairs"
    if (args.exchange == "allgather") != (args.partition == "pairs"):
        raise SystemExit("supported combinations: --exchange a2a --partition frames (default), --exchange allgather --partition pairs")
    if PAIRS:
        os.environ.setdefault("MCORB_BENCH_GROUPS", "1")   # the literal 8(e) path is run round by round, not pipelined
    # jobs in flight: with the selection on the GPU a job is one submission and four (of 512 images) are enough (profiles/r04_overlap.txt);
    # the host selection needs more to hide its round trip
    S = args.slots if args.slots else ((2 if PAIRS else 12) if DIST else (6 if os.environ.get("MCORB_SELECT") == "host" else 4))
    G = int(os.environ.get("MCORB_BENCH_GROUPS", "6")) if DIST else 1   # slot groups: with N > 1 G-1 groups extract ahead while one is matched
    IT = 2 if DIST else 1              # exchange rounds per step on the N > 1 path (one round = one group = SG slot jobs per rank)
    if DIST and S % G:
        S += G - S % G
    SG = S // G                        # slots per group
    # (the larger batch pays on the fused single-GPU path of the 720p rig only: the sharded pipeline and the 1080p rig measured
    # 5 - 8 % slower with it -- 35.9 vs 37.9 k, 8.1 vs 8.8 k -- and keep 128 images per job)
    ipl = IMAGES_PER_LAUNCH if (not DIST and args.config == "720p4") else 128
    F = args.frames if args.frames else SG * max(1, ipl // NCAMS)
    S = max(1, min(S, F)) if not DIST else S
    fps = F // SG                      # rig frames per slot job
    if fps * SG != F:
        raise SystemExit("--frames must be a multiple of the slots per group (%d)" % SG)
    # (several ranks share the node's CPUs: the engine divides its core budget by LOCAL_WORLD_SIZE and its slot drivers poll
    # their events with short sleeps instead of spinning, so nothing has to be set here)
    rig = mcorb.Rig(NCAMS, W, H, max_frames=fps, nslots=S, nfeatures=NFEAT, device_id=local)
    if args.graph is not None:
        rig.set_graph(args.graph)
    if rig.select_mode() == "gpu":   # the selection's time slot holds k_select + k_assemble (HIP events), not the host stage's wall time
        TIMING_FIELD.pop("select_host", None)
        TIMING_FIELD["k_select+k_assemble"] = "select_us"
    kcap = rig.kcap
    total_frames = F * N               # rig frames per exchange round over all ranks (N = 1: per step)

    # ---- inputs: synthetic rig frames, staged into HBM before the timed region ----
    from importlib import import_module
    shard = import_module("mc-slam_amd.sharding")
    if not DIST:
        mine = [(f, c) for f in range(F) for c in range(NCAMS)]
    elif PAIRS:
        mine = shard.images_of_rank(rank, N, NCAMS, total_frames)
    else:
        mine = shard.a2a_images_of_rank(rank, N, NCAMS, total_frames)
    assert len(mine) == F * NCAMS, (len(mine), F * NCAMS)
    per_slot = fps * NCAMS
    slot_imgs = []
    for s in range(S):
        i = s % SG
        imgs = [mcorb.synth_rig_frame(f, NCAMS, c, W, H) for (f, c) in mine[i * per_slot:(i + 1) * per_slot]]
        rig.upload(imgs, slot=s)
        slot_imgs.append(imgs)

    if PAIRS:
        # ---- SURVEY 8(e) as written: all-gather of {n, desc[K][32]} per camera, pair (i, j) of frame f on rank (i + j + f) mod N,
        #      the accepted (query, train) tables back to rank 0 for computeIntraMatches' serial merge ----
        nsets = F * NCAMS
        IT = 1
        local_desc = [torch.zeros((nsets, kcap, 32), dtype=torch.uint8, device="cuda")]
        local_cnt = [torch.zeros(nsets, dtype=torch.int32, device="cuda")]
        all_desc = torch.zeros((N * nsets, kcap, 32), dtype=torch.uint8, device="cuda")
        all_cnt = torch.zeros(N * nsets, dtype=torch.int32, device="cuda")
        batches = shard.pair_batches(rank, N, NCAMS, total_frames, fps)       # jobs of one slot's worth of frames
        my_pairs = shard.pairs_of_rank(rank, N, NCAMS, total_frames)
        maxp = max(len(shard.pairs_of_rank(r, N, NCAMS, total_frames)) for r in range(N))
        table_h = torch.zeros((maxp, 1 + 2 * kcap), dtype=torch.int32).pin_memory()
        table_d = torch.zeros((maxp, 1 + 2 * kcap), dtype=torch.int32, device="cuda")
        gathered = [torch.zeros_like(table_d) for _ in range(N)] if rank == 0 else None
        gidx = shard.gathered_set_index(N, NCAMS, total_frames)
        my_frames = list(range(total_frames))
        torch.cuda.synchronize()
        cstream = torch.cuda.Stream()
        tstream = cstream.cuda_stream
        assert tstream != 0
        exchange_bytes = N * nsets * (kcap * 32 + 4)
        pairs_result = {}

        def pairs_round(timed=False):
            for i in range(SG):
                rig.extract_submit(per_slot, slot=i)
            for i in range(SG):
                rig.extract_wait(slot=i)
                if timed:
                    account(i)      # (the extraction kernels; the k-NN jobs below overwrite the slot's timing record)
                rig.export_descriptors_dev(local_desc[0][i * per_slot].data_ptr(), local_cnt[0][i * per_slot:].data_ptr(), per_slot,
                                           slot=i, then_stream=None if gloo else tstream)
            if gloo:                                                # rehearsal on a 1-GPU box: the collective on host tensors
                torch.cuda.synchronize()
                hd = [torch.zeros(local_desc[0].shape, dtype=torch.uint8) for _ in range(N)]
                hc = [torch.zeros(local_cnt[0].shape, dtype=torch.int32) for _ in range(N)]
                dist.all_gather(hd, local_desc[0].cpu())
                dist.all_gather(hc, local_cnt[0].cpu())
                all_desc.copy_(torch.cat(hd))
                all_cnt.copy_(torch.cat(hc))
                torch.cuda.synchronize()
            else:
                with torch.cuda.stream(cstream):
                    dist.all_gather_into_tensor(all_desc, local_desc[0])    # the exchange step (RCCL over xGMI)
                    dist.all_gather_into_tensor(all_cnt, local_cnt[0])
            k, inflight = 0, {}
            tab = table_h.numpy()
            for b, (pairs, psets) in enumerate(batches):
                slot = b % SG
                if slot in inflight:
                    k = collect_pairs(slot, inflight.pop(slot), tab)
                rig.match_pairs_external_dev_submit(all_desc.data_ptr(), all_cnt.data_ptr(), N * nsets, psets, slot=slot,
                                                    after_stream=None if gloo else tstream)
                inflight[slot] = (b, len(pairs))
            for slot in sorted(inflight, key=lambda s_: inflight[s_][0]):
                collect_pairs(slot, inflight[slot], tab)
            if gloo:
                gathered_h = [torch.zeros_like(table_h) for _ in range(N)] if rank == 0 else None
                dist.gather(table_h, gathered_h, dst=0)
            else:
                table_d.copy_(table_h, non_blocking=True)
                dist.gather(table_d, gathered, dst=0)                       # the tables return to the rank that merges
            if rank == 0:
                counts = all_cnt.cpu().numpy()
                lists = {}
                for r_ in range(N):
                    g_ = (gathered_h[r_] if gloo else gathered[r_].cpu()).numpy()
                    for row, key in zip(g_, shard.pairs_of_rank(r_, N, NCAMS, total_frames)):
                        n_ = int(row[0])
                        lists[key] = (row[1:1 + n_].astype(np.uint32), row[1 + kcap:1 + kcap + n_].astype(np.uint32))
                for f in range(total_frames):
                    pl = [lists[(f, i, j)] for i in range(NCAMS - 1) for j in range(i + 1, NCAMS)]
                    pairs_result[f] = mcorb.merge_tracks(NCAMS, [counts[gidx[(f, c)]] for c in range(NCAMS)], pl)

        batch_base = np.cumsum([0] + [len(p) for p, _ in batches])

        def collect_pairs(slot, job, tab):
            b, npairs = job
            rig.match_wait(slot=slot)
            base = int(batch_base[b])
            for p_ in range(npairs):
                i1, i2 = rig.pairlist(p_, slot=slot)
                tab[base + p_, 0] = len(i1)
                tab[base + p_, 1:1 + len(i1)] = i1
                tab[base + p_, 1 + kcap:1 + kcap + len(i2)] = i2
            return base + npairs

    if DIST and not PAIRS:
        nsets = F * NCAMS                                       # sets a rank exports = sets it receives, per step
        send_splits = shard.a2a_send_splits(rank, N, NCAMS, total_frames)
        recv_splits = shard.a2a_recv_splits(rank, N, NCAMS, total_frames)
        assert sum(send_splits) == nsets and sum(recv_splits) == nsets
        local_desc = [torch.zeros((nsets, kcap, 32), dtype=torch.uint8, device="cuda") for _ in range(G)]
        recv_desc = [torch.zeros((nsets, kcap, 32), dtype=torch.uint8, device="cuda") for _ in range(G)]
        local_cnt = [torch.zeros(nsets, dtype=torch.int32, device="cuda") for _ in range(G)]
        recv_cnt = [torch.zeros(nsets, dtype=torch.int32, device="cuda") for _ in range(G)]
        my_frames, sets = shard.a2a_match_sets(rank, N, NCAMS, total_frames)   # set index of (f, c) inside the received block
        assert len(my_frames) == F
        torch.cuda.synchronize()   # the zero fills above ran on torch's stream; the engine writes these tensors from its own
        # The collectives are ordered on a stream of their own.  torch's default stream is the NULL stream, whose handle is
        # 0: passed to the engine that reads as "no stream" and the hand-offs fall back to synchronising the host (which
        # round 2's first measurements did without saying so: 0.55 ms of the main thread per export).
        cstream = torch.cuda.Stream()
        tstream = cstream.cuda_stream                               # raw HIP stream handle, never 0
        assert tstream != 0
        exchange_bytes = nsets * (kcap * 32 + 4)

    hostprof = {} if os.environ.get("MCORB_BENCH_PROF") else None   # main-thread seconds per phase of the sharded loop

    def hp(name, t0):
        if hostprof is not None:
            hostprof[name] = hostprof.get(name, 0.0) + time.perf_counter() - t0

    def extract_submit(g):
        t0 = time.perf_counter()
        for i in range(SG):
            rig.extract_submit(per_slot, slot=g * SG + i)
        hp("extract_submit", t0)

    def exchange_and_match(g):
        """Waits for group g's extraction, exchanges the descriptors and queues the matching; match_finish(g) collects it.
        Everything between the extraction's completion and the match job is stream-ordered (no host synchronisation):
        slot streams -> torch's stream (export) -> collective -> slot streams (match)."""
        if gloo:                                                # rehearsal path only: host tensors
            for i in range(SG):
                s = g * SG + i
                rig.extract_wait(slot=s)
                rig.export_descriptors_dev(local_desc[g][i * per_slot].data_ptr(), local_cnt[g][i * per_slot:].data_ptr(), per_slot, slot=s)
            hd, hc = torch.zeros(recv_desc[g].shape, dtype=torch.uint8), torch.zeros(recv_cnt[g].shape, dtype=torch.int32)
            dist.all_to_all_single(hd, local_desc[g].cpu(), recv_splits, send_splits)
            dist.all_to_all_single(hc, local_cnt[g].cpu(), recv_splits, send_splits)
            recv_desc[g].copy_(hd)
            recv_cnt[g].copy_(hc)
            torch.cuda.synchronize()
        else:
            for i in range(SG):
                s = g * SG + i
                t0 = time.perf_counter()
                rig.extract_wait(slot=s)
                hp("extract_wait", t0)
                t0 = time.perf_counter()
                rig.export_descriptors_dev(local_desc[g][i * per_slot].data_ptr(), local_cnt[g][i * per_slot:].data_ptr(), per_slot,
                                           slot=s, then_stream=tstream)
                hp("export", t0)
            t0 = time.perf_counter()
            with torch.cuda.stream(cstream):
                dist.all_to_all_single(recv_desc[g], local_desc[g], recv_splits, send_splits)   # the exchange step (RCCL over xGMI)
                dist.all_to_all_single(recv_cnt[g], local_cnt[g], recv_splits, send_splits)
            hp("all_to_all", t0)
        t0 = time.perf_counter()
        for i in range(SG):                                     # this rank's frames, split over the group's slots
            rig.match_external_dev_submit(recv_desc[g].data_ptr(), recv_cnt[g].data_ptr(), nsets, sets[i * fps:(i + 1) * fps],
                                          slot=g * SG + i, after_stream=None if gloo else tstream)
        hp("match_submit", t0)

    def match_finish(g):
        t0 = time.perf_counter()
        for i in range(SG):
            rig.match_wait(slot=g * SG + i)
        hp("match_wait", t0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # per-kernel durations, accumulated from the HIP events the engine records on each slot's stream
    ksum = {k: 0.0 for k in TIMING_FIELD}

    ksamples = [0]

    def account(s):
        t = rig.timing(slot=s)
        if t["fast_us"] <= 0:      # a job replayed from its graph has no per-kernel events: not a sample
            return
        ksamples[0] += 1
        for k, f in TIMING_FIELD.items():
            ksum[k] += t[f]

    def run_steps(nsteps, timed, stage=None):
        """nsteps steps = nsteps*F frames per rank.
        N == 1: rolling submission, S jobs always in flight (a slot is resubmitted as soon as its previous job
        is collected), so step boundaries do not drain the pipeline; stage(slot), if given, runs in front of every job.
        N > 1: G slot groups rotate; extraction of steps k+1 .. k+G-2 is in flight while step k's descriptors
        are exchanged and matched."""
        if PAIRS:
            for _ in range(nsteps):
                pairs_round(timed)
            return
        if DIST:
            # step k runs on group k % G.  While step k's descriptors are exchanged, step k-1 is being matched and steps
            # k+1 .. k+G-2 are being extracted; a group is re-armed (next extraction) as soon as its matching is collected.
            def collect(g):
                match_finish(g)
                if timed:
                    for i in range(SG):
                        account(g * SG + i)
            if G == 1:
                for k in range(nsteps * IT):
                    extract_submit(0)
                    exchange_and_match(0)
                    collect(0)
                return
            nsteps = nsteps * IT               # rounds
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
            if stage:
                stage(s)
            rig.process_submit(fps, slot=s)
        for j in range(jobs):
            s = j % S
            rig.process_wait(slot=s)
            if timed:
                account(s)
            if j + S < jobs:
                if stage:
                    stage(s)
                rig.process_submit(fps, slot=s)

    def timed_region(nsteps, timed, stage=None):
        barrier()
        if hostprof is not None:
            hostprof.clear()
        t0 = time.perf_counter()
        run_steps(nsteps, timed, stage)
        t1 = time.perf_counter()
        barrier()
        dt = time.perf_counter() - t0
        if hostprof is not None and rank == 0:
            print("[bench] main thread, %d steps, %.1f ms (+%.1f ms closing barrier): " % (nsteps, (t1 - t0) * 1e3, (dt - (t1 - t0)) * 1e3) +
                  ", ".join("%s %.1f ms" % (k, v * 1e3) for k, v in sorted(hostprof.items(), key=lambda x: -x[1])), file=sys.stderr)
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if gloo else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    run_steps(args.warmup, False)
    # region length: one calibration region of --steps steps, then as many steps per region as --min-region-s needs
    # (all ranks agree: timed_region returns the max over ranks)
    steps_requested = args.steps
    if args.min_region_s > 0:
        dt_cal = timed_region(args.steps, False)
        need = int(np.ceil(1.1 * args.min_region_s / max(dt_cal / args.steps, 1e-9)))   # (+10 %: the calibration region runs a little slow)
        if dist is not None:
            tt = torch.tensor([need], dtype=torch.int64, device="cpu" if gloo else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            need = int(tt.item())
        args.steps = max(args.steps, need)
    dts = [timed_region(args.steps, True) for _ in range(max(1, args.repeats))]
    dt = float(np.median(dts))

    if args.dump_tracks and PAIRS:
        if rank == 0:
            np.savez(args.dump_tracks, frames=np.array(list(range(fps))), **{"t%d" % i: pairs_result[i][0] for i in range(fps)})
    elif args.dump_tracks:
        if world > 1:   # one file per rank
            args.dump_tracks = args.dump_tracks.replace(".npz", "_r%d.npz" % rank)
        np.savez(args.dump_tracks, frames=np.array(my_frames[:fps] if DIST else list(range(fps))),
                 **{"t%d" % i: rig.tracks(i, slot=0)[0] for i in range(fps)})

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = total_frames * IT * args.steps / dt
    launches = args.steps * IT * SG * len(dts)
    # workload constants for the algorithmic-byte formulas
    lv = [rig.level_size(l) for l in range(rig.nlevels)]
    Spx = sum(w * h for w, h in lv)
    S0, s_last = lv[0][0] * lv[0][1], lv[-1][0] * lv[-1][1]
    info = rig.info()
    nimg_launch = per_slot
    pairs_launch = fps * NCAMS * (NCAMS - 1) // 2
    Kc = np.mean([sum(len(rig.candidates(m, l, slot=0)[0]) for l in range(rig.nlevels)) for m in range(min(4, per_slot))])
    K = np.mean([rig.features(m, slot=0)[1].shape[0] for m in range(min(4, per_slot))])
    # Which kernel dominates is decided on ISOLATED durations (one job at a time, nothing else on the GPU): with several
    # jobs in flight every kernel's wall duration is stretched by whatever shares the GPU with it, and the ranking of
    # two close kernels flips from run to run.  The roofline figures themselves use the timed region, as specified.
    iso = None
    if not DIST:
        acc = {k: [] for k in TIMING_FIELD}
        rig.set_graph(0)           # launch by launch: every kernel between its own HIP events
        for _ in range(args.iso_jobs):
            rig.process_submit(fps, slot=0)
            rig.process_wait(slot=0)
            t = rig.timing(slot=0)
            for k, f in TIMING_FIELD.items():
                acc[k].append(t[f])
        iso = {k: float(np.median(v)) for k, v in acc.items()}
    rank_by = {k: (iso or ksum)[k] for k in GPU_KERNELS}
    dominant = max(rank_by, key=rank_by.get)
    sampled = max(1, ksamples[0])            # jobs of the timed regions that ran launch by launch (all of them unless graphs are on)
    avg_us = ksum[dominant] / sampled if ksamples[0] else (iso[dominant] if iso else 0.0)
    units = pairs_launch if dominant == "k_knn2" else nimg_launch
    unit_bytes = algorithmic_bytes(dominant, Spx, S0, s_last, K, Kc, info["bucket_total"])
    achieved = unit_bytes * units / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
    # PMC-derived HBM bytes per launch: only if profiles/traffic.json was measured on exactly these kernels
    traffic, traffic_note, valu_insts = None, "profiles/traffic.json absent", None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("kernels_sha256_16") != kernels_sha():
                traffic_note = "profiles/traffic.json was measured on other kernel sources (sha %s, now %s): not reported" % (
                    tj.get("kernels_sha256_16"), kernels_sha())
            elif tj.get("config") != args.config:
                traffic_note = "profiles/traffic.json is for config %s" % tj.get("config")
            else:
                traffic = int(tj[dominant]["bytes_per_image"] * nimg_launch)
                traffic_note = tj.get("_correction", "")
                # (the counters were read on launches of tj[...]["images_per_launch"] images: scaled to this run's launch size)
                lscale = nimg_launch / float(tj[dominant].get("images_per_launch", nimg_launch))
                valu_insts = tj[dominant].get("valu_insts_per_launch")
                valu_insts = valu_insts * lscale if valu_insts else valu_insts
        except Exception as e:      # noqa: BLE001
            traffic_note = "profiles/traffic.json unreadable: %s" % e
    out = {
        "metric": "multi-cam frames/sec (%d-cam %dx%d @%d kpts/cam, extract + intra-rig match)" % (NCAMS, W, H, NFEAT),
        "value": round(value, 2), "unit": "frames/s", "n_gpus": N, "steps": args.steps, "steps_requested": steps_requested,
        "region_s": round(dt, 4), "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_frame": round(dt / args.steps / (total_frames * IT) * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "repeats": len(dts), "value_min": round(total_frames * IT * args.steps / max(dts), 2),
        "value_max": round(total_frames * IT * args.steps / min(dts), 2),
        "config": {"workload": cfg["label"], "name": args.config,
                   "frames_per_rank_per_step": F * IT, "slots": S, "frames_per_launch": fps, "cameras": NCAMS, "nfeatures": NFEAT,
                   "sharding": "single GPU" if not DIST else
                   ("camera (c+f) mod N for extraction, one RCCL all-gather of every camera's {n, desc[K][32]} per round, camera pair (i,j) of frame f "
                    "matched on rank (i+j+f) mod N, accepted lists gathered on rank 0 for the serial track merge (SURVEY 8e as written; not pipelined)"
                    if PAIRS else
                    "camera (c+f) mod N for extraction, one RCCL all-to-all of descriptor sets per step (each set goes to rank f mod N only), frame f mod N for matching")},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_note": traffic_note,
                     "avg_launch_us": round(avg_us, 2), "algorithmic_bytes_per_launch": int(unit_bytes * units),
                     "images_per_launch": nimg_launch, "fast_candidates_per_image": int(Kc),
                     "note": "avg_launch_us is the HIP-event average over the timed regions, where %d jobs share the GPU; "
                             "isolated_* is the same kernel with one job in flight (median of %d)" % (S, args.iso_jobs)},
        "kernel_us_per_step": {k: round(v / sampled * launches / args.steps / len(dts), 2) for k, v in ksum.items()},
        "kernel_timing_samples": {"jobs_sampled": ksamples[0], "jobs_timed": launches,
                                  "note": "jobs that ran launch by launch with per-kernel HIP events inside the timed regions; the others were replayed "
                                          "from their captured HIP graph (mcorb_rig_set_graph)"},
    }
    # the whole path against the HBM roof: SURVEY 8(d)'s bytes per rig frame (every plane once per logical pass, the blur's
    # 2 S included although this build no longer moves them) and the bytes this build's kernels really have to move
    pairs_frame = NCAMS * (NCAMS - 1) // 2
    b_extract = (Spx - s_last) + (Spx - S0) + Spx + 2 * Spx + K * (512 + 32) + Kc * 12 + K * 28
    b_match = 2 * K * 32 + K * 16
    b_frame = NCAMS * b_extract + pairs_frame * b_match
    b_frame_build = NCAMS * sum(algorithmic_bytes(k, Spx, S0, s_last, K, Kc, info["bucket_total"]) for k in GPU_KERNELS if k != "k_knn2") \
        + pairs_frame * algorithmic_bytes("k_knn2", Spx, S0, s_last, K, Kc, info["bucket_total"])
    out["roofline"]["whole_path"] = {
        "bytes_per_frame_survey_8d": int(b_frame), "achieved": round(b_frame * value / 1e9, 1), "frac": round(b_frame * value / 1e9 / HBM_PEAK_GBS, 4),
        "bytes_per_frame_this_build": int(b_frame_build), "frac_this_build": round(b_frame_build * value / 1e9 / HBM_PEAK_GBS, 4),
        "unit": "GB/s", "note": "frames/s x bytes per rig frame / 8 TB/s; 8(d) counts a full-plane blur (2 S per image) that is fused "
                                "into the descriptor kernel here and never travels"}
    # per-kernel instruction-issue floors (SQ_INSTS_VALU / SQ_INSTS_MFMA per launch from profiles/traffic.json, same sha check):
    # a wave64 vector instruction occupies its SIMD's issue port for 4 cycles in these kernels whatever its class
    # (profiles/r03_valu_rates.txt: 2.2 for v_add / v_and / v_xor when two waves issue them back to back, 4.1 for everything
    # else; SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.008 quad-cycles in k_fast_cells), a 32x32x32 i8 multiply its matrix pipe for 32
    SIMDS, CLK_HZ = 1024, 2.4e9
    floors = {}
    if traffic is not None:
        for k in GPU_KERNELS:
            e = tj.get(k, {})
            if e.get("valu_insts_per_launch"):
                sc = nimg_launch / float(e.get("images_per_launch", nimg_launch))   # counters of a 128-image launch -> this launch size
                f = {"valu_insts": int(e["valu_insts_per_launch"] * sc), "valu_floor_us": round(e["valu_insts_per_launch"] * sc * 4 / SIMDS / CLK_HZ * 1e6, 1)}
                if e.get("mfma_insts_per_launch"):
                    f["mfma_insts"] = int(e["mfma_insts_per_launch"] * sc)
                    f["mfma_floor_us"] = round(e["mfma_insts_per_launch"] * sc * 32.6 / SIMDS / CLK_HZ * 1e6, 1)   # 32.6 cycles per scaled FP4 step (r04_fp4_probe.txt)
                floors[k] = f
    if valu_insts:
        # k_fast_cells' issue floor by phase: staging, pass 1 and the work-list expansion are made of the
        # 4.1-cycle class (packed 16-bit, v_perm, mbcnt: profiles/r03_valu_rates.txt), the arc score + NMS mix sustains 3.0 cycles
        # per instruction at the kernel's occupancy (the same file's mix_score row); 47 % of a cell-wave's 794 instructions are
        # score + NMS (245 + 130 of 794)
        share_score, cyc_a, cyc_b = 0.47, 4.1, 3.0
        cyc = (1.0 - share_score) * cyc_a + share_score * cyc_b
        floor_us = valu_insts * cyc / SIMDS / CLK_HZ * 1e6
        out["roofline"]["valu"] = {"insts_per_launch": int(valu_insts), "cycles_per_inst": round(cyc, 2),
                                   "cycles_by_phase": {"staging_pass1_expansion": cyc_a, "arc_score_nms": cyc_b, "share_arc_score_nms": share_score},
                                   "simds": SIMDS, "clock_ghz": 2.4,
                                   "floor_us": round(floor_us, 1), "floor_us_flat_4_cycles": round(valu_insts * 4 / SIMDS / CLK_HZ * 1e6, 1),
                                   "frac_in_flight": round(floor_us / avg_us, 3),
                                   "evidence": "profiles/r03_valu_rates.txt (tools/valu_rates.hip, 1/2/4/8 waves per SIMD)"}
    if DIST:
        out["exchange"] = {"collective": "all_gather_into_tensor + gather of the pair tables to rank 0" if PAIRS else "all_to_all_single (uneven splits)", "bytes_sent_per_rank_per_step": int(exchange_bytes * IT), "rounds_per_step": IT,
                           "send_splits_rank0": None if PAIRS else send_splits,
                           "ranks_seen": dist.get_world_size(), "backend": dist.get_backend()}
    if iso:
        ia = unit_bytes * units / (iso[dominant] * 1e-6) / 1e9
        out["roofline"].update({"isolated_launch_us": round(iso[dominant], 2), "isolated_achieved": round(ia, 2),
                                "isolated_frac": round(ia / HBM_PEAK_GBS, 5)})
        if "valu" in out["roofline"]:
            out["roofline"]["valu"]["frac_isolated"] = round(out["roofline"]["valu"]["floor_us"] / iso[dominant], 3)
        out["kernel_us_per_launch_isolated"] = {k: round(v, 2) for k, v in iso.items()}
        # every kernel against its own algorithmic bytes (isolated): where the path stands as a whole
        out["roofline_all_kernels_isolated"] = {
            k: dict({"us": round(iso[k], 1),
                     "GBps": round(algorithmic_bytes(k, Spx, S0, s_last, K, Kc, info["bucket_total"]) *
                                   (pairs_launch if k == "k_knn2" else nimg_launch) / (iso[k] * 1e-6) / 1e9, 1)}, **floors.get(k, {}))
            for k in GPU_KERNELS if iso[k] > 0}

    if N == 1 and not DIST and not args.no_staging:
        # the same steps with the frame hand-off in front of every batch: the reader's side of the boundary
        plane = W * H
        dt_st = timed_region(args.steps, False, stage=lambda s: rig.upload_staged(per_slot, slot=s))
        dt_u8 = timed_region(args.steps, False, stage=lambda s: rig.upload(slot_imgs[s], slot=s))
        out["value_with_staging"] = round(total_frames * args.steps / dt_st, 2)
        out["value_with_upload_u8"] = round(total_frames * args.steps / dt_u8, 2)
        out["pcie_gbs"] = round(total_frames * args.steps * NCAMS * plane / dt_st / 1e9, 2)
        out["staging_note"] = ("value_with_staging: DMA out of the pinned staging planes (mcorb_rig_upload_staged) before every batch; "
                               "value_with_upload_u8: mcorb_rig_upload_u8 from pageable memory (host copy + DMA); pcie_gbs: host->device "
                               "bytes/s in the staged region.  `value` itself keeps its inputs resident in HBM; fed from a host it is "
                               "bounded by these.")

    if N == 1 and not DIST and args.host_cores > 0 and hasattr(os, "sched_setaffinity"):
        # how much of `value` hangs on host cores (selection workers, slot drivers, track merges): the same steps from a
        # second rig whose threads -- created under the narrowed affinity mask -- and the caller share args.host_cores cores
        allowed = sorted(os.sched_getaffinity(0))
        if len(allowed) > args.host_cores:
            os.sched_setaffinity(0, set(allowed[:args.host_cores]))
            # With few cores the host's share of a job (keypoint records, accept lists, track merges) takes longer, so a slot
            # spends more of its cycle off the GPU: two more slots than jobs admitted to the GPU at once (mcorb_params.gpu_jobs)
            # keep the GPU at its four jobs.  (With all cores the plain four slots measured 4 % faster: scripts/gpujobs_sweep.sh.)
            S_main = S
            try:
                rig_main = rig
                S = S_main + HOST_CORES_EXTRA_SLOTS
                rig = mcorb.Rig(NCAMS, W, H, max_frames=fps, nslots=S, nfeatures=NFEAT, device_id=local, gpu_jobs=S_main)
                for s_ in range(S):
                    rig.upload(slot_imgs[s_ % len(slot_imgs)], slot=s_)
                run_steps(args.warmup, False)
                dt_hc = timed_region(args.steps, False)
                out["value_host_cores"] = {"cores": args.host_cores, "value": round(S * fps * args.steps / dt_hc, 2),
                                           "engine_workers": rig.info().get("host_threads"), "slots": S, "gpu_jobs": S_main,
                                           "note": "process confined to %d of the %d cores it may use (sched_setaffinity before the rig is "
                                                   "created), %d slots of which at most %d have their job on the GPU at a time (the others "
                                                   "are being post-processed on the host); `value` itself ran with all cores and %d slots"
                                                   % (args.host_cores, len(allowed), S, S_main, S_main)}
                rig.close()
            finally:
                rig = rig_main
                S = S_main
                os.sched_setaffinity(0, set(allowed))

    if N == 1 and not DIST and not args.no_cpu:
        ncpu = os.cpu_count()
        times, t_ext, results = cpu_baseline(cfg, list(range(4 + args.cpu_frames)), NCAMS)
        times, t_ext = times[4:], t_ext[4:]
        t1, _, _ = cpu_baseline(cfg, list(range(1 + max(4, args.cpu_frames // 8))), 1)
        t1 = t1[1:]
        out["cpu_baseline"] = {"value": round(1.0 / float(np.median(times)), 3), "unit": "frames/s", "cores": NCAMS,
                               "kind": "port", "value_1thread": round(1.0 / float(np.median(t1)), 3),
                               "note": "the repository's CPU restatement (oracle/), one thread per camera for extraction and ONE matcher thread -- "
                                       "not the reference's OpenCV build, whose BFMatcher::knnMatch (batchDistance) is parallel_for_-threaded: a reported "
                                       "baseline that flatters the GPU, never the target (the roofline fractions are)",
                               "sample": "%d rig frames (%s) after 4 warm-ups, CPU oracle, one thread per camera for extraction + "
                                         "matching on the caller thread; median %.1f ms (extract %.1f + match %.1f), p95 %.1f ms; "
                                         "single thread: %d frames, median %.1f ms; host has %d logical cores"
                                         % (len(times), cfg["label"], np.median(times) * 1e3, np.median(t_ext) * 1e3,
                                            np.median(times - t_ext) * 1e3, np.percentile(times, 95) * 1e3, len(t1),
                                            np.median(t1) * 1e3, ncpu)}
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
        out["single_frame_latency_ms"] = {k: lat[k] for k in ("upload_ms", "extract_match_ms", "readback_ms", "total_ms", "total_p95_ms", "separated")}
        out["single_frame_latency_ms"]["note"] = ("upload_ms = staging copy + DMA enqueue; the PCIe transfer itself runs at the head of "
                                                  "extract_match_ms (`separated` drains the stream between the two)")
    if N == 1 and not DIST and not args.no_extra_legs and args.config == "720p4":
        # two more driver-visible numbers, each from a fresh child process after this one's legs are done (its rig is closed):
        rig.close()
        fd = child_value(["--force-dist", "--repeats", "1"])
        out["value_force_dist"] = ({"value": fd["value"], "sharding": fd["config"]["sharding"], "exchange": fd.get("exchange"),
                                    "note": "the N > 1 code path (export -> RCCL all-to-all -> external match) at world size 1, same steps: the "
                                            "like-for-like denominator for value(N) / (N value(1)); `value` itself runs the fused path"}
                                   if "value" in fd else fd)
        c2 = child_value(["--config", "1080p8", "--repeats", "1"])
        out["configs2_1080p8"] = ({"value": c2["value"], "unit": c2["unit"], "ms_per_frame": c2["ms_per_frame"], "workload": c2["config"]["workload"],
                                   "frames_per_launch": c2["config"]["frames_per_launch"], "steps": c2["steps"],
                                   "note": "BASELINE configs[2] (8 cameras, 1920x1080, 28 camera pairs per rig frame) on ONE GPU; at --gpus 8 it is one camera per GPU"}
                                  if "value" in c2 else c2)
    print(json.dumps(out), flush=True)
    rig.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
