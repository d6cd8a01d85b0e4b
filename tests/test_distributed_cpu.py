"""world_size-2 gloo test of the multi-GPU data path (camera sharding -> exchange of
per-camera descriptors [all-gather, and the all-to-all bench.py uses] -> per-frame matching), with the CPU oracle
standing in for the GPU stages.  What is under test is the placement / indexing logic bench.py uses for N > 1
(mc-slam_amd/sharding.py) and that the exchange reproduces the single-process result."""
import os
import sys
from importlib import import_module

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NCAMS, W, H, NFEAT, NLEV, KCAP = 3, 200, 150, 150, 3, 256
FRAMES_PER_RANK = 2


def _extract(f, c):
    import oracle_lib as O
    synth = import_module("mc-slam_amd.synth")
    img = synth.synth_rig_frame_numpy(f, NCAMS, c, W, H)
    mono, k, d = O.OracleExtractor(NFEAT, 1.2, NLEV)(img)
    assert mono >= 0 and len(d) <= KCAP
    return d


def _free_port():
    """bind to port 0 and read it back: two test sessions on one box do not collide (advisor finding, round 3)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _worker_a2a(rank, world, port, q):
    """the exchange bench.py uses: every set goes only to the rank that matches its frame (uneven all-to-all)"""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist
    import oracle_lib as O
    shard = import_module("mc-slam_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = FRAMES_PER_RANK * world
    mine = shard.a2a_images_of_rank(rank, world, NCAMS, total)
    send, recv = shard.a2a_send_splits(rank, world, NCAMS, total), shard.a2a_recv_splits(rank, world, NCAMS, total)
    local = torch.zeros((len(mine), KCAP, 32), dtype=torch.uint8)
    cnt = torch.zeros(len(mine), dtype=torch.int32)
    for i, (f, c) in enumerate(mine):
        d = _extract(f, c)
        local[i, :len(d)] = torch.from_numpy(d)
        cnt[i] = len(d)
    block = torch.zeros((sum(recv), KCAP, 32), dtype=torch.uint8)
    counts = torch.zeros(sum(recv), dtype=torch.int32)
    dist.all_to_all_single(block, local, recv, send)      # the exchange step
    dist.all_to_all_single(counts, cnt, recv, send)
    block, counts = block.numpy(), counts.numpy()
    frames, sets = shard.a2a_match_sets(rank, world, NCAMS, total)
    out = {}
    for f, row in zip(frames, sets):
        out[f] = O.intra_matches([block[s, :counts[s]] for s in row])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist
    import oracle_lib as O
    shard = import_module("mc-slam_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = FRAMES_PER_RANK * world
    mine = shard.images_of_rank(rank, world, NCAMS, total)
    per = shard.sets_per_rank(world, NCAMS, total)
    local = torch.zeros((per, KCAP, 32), dtype=torch.uint8)
    cnt = torch.zeros(per, dtype=torch.int32)
    for i, (f, c) in enumerate(mine):
        d = _extract(f, c)
        local[i, :len(d)] = torch.from_numpy(d)
        cnt[i] = len(d)
    all_desc = [torch.zeros_like(local) for _ in range(world)]
    all_cnt = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(all_desc, local)                 # the exchange step
    dist.all_gather(all_cnt, cnt)
    block = torch.cat(all_desc).numpy()
    counts = torch.cat(all_cnt).numpy()
    frames, sets = shard.match_sets(rank, world, NCAMS, total)
    out = {}
    for f, row in zip(frames, sets):
        descs = [block[s, :counts[s]] for s in row]
        tr, mg = O.intra_matches(descs)
        out[f] = (tr, mg)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _worker_pairs(rank, world, port, q):
    """SURVEY 8(e) / north_star as written: one all-gather of every camera's descriptors, camera PAIR (i, j) of a frame matched on
    one rank (bench.py --exchange allgather --partition pairs), the accepted lists gathered on rank 0, which runs the serial
    merge.  The oracle's BruteForceMatch stands in for the GPU k-NN; the merge is the product's host function."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist
    import oracle_lib as O
    pkg = import_module("mc-slam_amd")
    shard = import_module("mc-slam_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = FRAMES_PER_RANK * world
    mine = shard.images_of_rank(rank, world, NCAMS, total)
    per = shard.sets_per_rank(world, NCAMS, total)
    local = torch.zeros((per, KCAP, 32), dtype=torch.uint8)
    cnt = torch.zeros(per, dtype=torch.int32)
    for i, (f, c) in enumerate(mine):
        d = _extract(f, c)
        local[i, :len(d)] = torch.from_numpy(d)
        cnt[i] = len(d)
    all_desc = [torch.zeros_like(local) for _ in range(world)]
    all_cnt = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(all_desc, local)                 # the exchange step: {n, desc[K][32]} of every camera to every rank
    dist.all_gather(all_cnt, cnt)
    block, counts = torch.cat(all_desc).numpy(), torch.cat(all_cnt).numpy()
    # this rank's pairs, in jobs of FRAMES_PER_RANK frames (what one engine slot takes)
    maxp = max(len(shard.pairs_of_rank(r, world, NCAMS, total)) for r in range(world))
    table = torch.zeros((maxp, 1 + 2 * KCAP), dtype=torch.int32)    # per pair: n, idx1[n], idx2[n]
    k = 0
    for pairs, psets in shard.pair_batches(rank, world, NCAMS, total, FRAMES_PER_RANK):
        for (f, i, j), (sa, sb) in zip(pairs, psets):
            i1, i2 = O.bruteforce_match(block[sa, :counts[sa]], block[sb, :counts[sb]])
            table[k, 0] = len(i1)
            table[k, 1:1 + len(i1)] = torch.from_numpy(i1.astype(np.int32))
            table[k, 1 + KCAP:1 + KCAP + len(i2)] = torch.from_numpy(i2.astype(np.int32))
            k += 1
    gathered = [torch.zeros_like(table) for _ in range(world)] if rank == 0 else None
    dist.gather(table, gathered, dst=0)              # the tables return to the rank that merges
    out = {}
    if rank == 0:
        idx = shard.gathered_set_index(world, NCAMS, total)
        lists = {}
        for r in range(world):
            for row, (f, i, j) in zip(gathered[r].numpy(), shard.pairs_of_rank(r, world, NCAMS, total)):
                n = int(row[0])
                lists[(f, i, j)] = (row[1:1 + n].astype(np.uint32), row[1 + KCAP:1 + KCAP + n].astype(np.uint32))
        for f in range(total):
            pl = [lists[(f, i, j)] for i in range(NCAMS - 1) for j in range(i + 1, NCAMS)]
            out[f] = pkg.merge_tracks(NCAMS, [counts[idx[(f, c)]] for c in range(NCAMS)], pl)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,worker", [(2, "allgather"), (2, "a2a"), (2, "pairs")])
def test_sharded_pipeline_equals_single_process(world, worker):
    import torch.multiprocessing as mp
    import oracle_lib as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    target = {"allgather": _worker, "a2a": _worker_a2a, "pairs": _worker_pairs}[worker]
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, out = q.get(timeout=240)
        results.update(out)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    total = FRAMES_PER_RANK * world
    assert sorted(results) == list(range(total))
    for f in range(total):
        descs = [_extract(f, c) for c in range(NCAMS)]
        tr, mg = O.intra_matches(descs)
        assert np.array_equal(results[f][0], tr) and results[f][1] == mg, "frame %d" % f


@pytest.mark.parametrize("world,ncams,fpr", [(1, 4, 8), (2, 4, 8), (4, 4, 8), (8, 4, 8), (8, 8, 4), (2, 3, 2)])
def test_placement_is_balanced_and_consistent(world, ncams, fpr):
    shard = import_module("mc-slam_amd.sharding")
    total = fpr * world
    per = shard.sets_per_rank(world, ncams, total)
    assert per == fpr * ncams                                     # fixed work per GPU: weak scaling
    idx = shard.gathered_set_index(world, ncams, total)
    assert sorted(idx.values()) == list(range(world * per))       # a bijection onto the gathered block
    seen = []
    for r in range(world):
        frames, sets = shard.match_sets(r, world, ncams, total)
        assert len(frames) == fpr and sets.shape == (fpr, ncams)
        seen += frames
        for f, row in zip(frames, sets):
            assert [idx[(f, c)] for c in range(ncams)] == row.tolist()
            if world >= ncams:                                    # one camera per GPU within a frame
                assert len({shard.owner(c, f, world) for c in range(ncams)}) == ncams
    assert sorted(seen) == list(range(total))


@pytest.mark.parametrize("world,ncams,fpr", [(1, 4, 8), (2, 4, 8), (4, 4, 8), (8, 4, 8), (8, 8, 4), (2, 3, 2), (3, 5, 3)])
def test_all_to_all_placement(world, ncams, fpr):
    """the all-to-all moves each set exactly once, to the rank that matches its frame; send / receive splits agree;
    the received block's index map is a bijection"""
    shard = import_module("mc-slam_amd.sharding")
    total = fpr * world
    sends = [shard.a2a_send_splits(r, world, ncams, total) for r in range(world)]
    for r in range(world):
        mine = shard.a2a_images_of_rank(r, world, ncams, total)
        assert sorted(mine) == sorted(shard.images_of_rank(r, world, ncams, total))
        assert [shard.dest(f, world) for f, _ in mine] == sorted(shard.dest(f, world) for f, _ in mine)   # grouped by destination
        assert sum(sends[r]) == len(mine) == fpr * ncams
        recv = shard.a2a_recv_splits(r, world, ncams, total)
        assert recv == [sends[src][r] for src in range(world)] and sum(recv) == fpr * ncams    # weak scaling: fixed volume
        frames, sets = shard.a2a_match_sets(r, world, ncams, total)
        assert frames == shard.frames_of_rank(r, world, total) and sets.shape == (fpr, ncams)
        assert sorted(sets.ravel().tolist()) == list(range(fpr * ncams))
        # simulate the transport: concatenate what every source sends here, look the sets up through the index map
        block = []
        for src in range(world):
            block += [fc for fc in shard.a2a_images_of_rank(src, world, ncams, total) if shard.dest(fc[0], world) == r]
        for f, row in zip(frames, sets):
            assert [block[i] for i in row] == [(f, c) for c in range(ncams)]


@pytest.mark.parametrize("world,ncams,fpr", [(1, 4, 8), (2, 4, 8), (4, 4, 8), (8, 4, 8), (8, 8, 4), (3, 5, 3)])
def test_pair_partition_placement(world, ncams, fpr):
    """every camera pair of every frame is matched on exactly one rank; the jobs a rank cuts its pairs into name at most one
    slot's worth of distinct sets; with the rotation by frame the load is even whenever a round holds a multiple of `world` frames"""
    shard = import_module("mc-slam_amd.sharding")
    total = fpr * world
    allp = []
    for r in range(world):
        mine = shard.pairs_of_rank(r, world, ncams, total)
        assert all(shard.pair_owner(i, j, f, world) == r and i < j for f, i, j in mine)
        allp += mine
        idx = shard.gathered_set_index(world, ncams, total)
        seen = []
        for pairs, psets in shard.pair_batches(r, world, ncams, total, fpr):
            assert psets.shape == (len(pairs), 2) and len(set(psets.ravel().tolist())) <= fpr * ncams
            assert len(pairs) <= fpr * ncams * (ncams - 1) // 2
            for (f, i, j), (a, b) in zip(pairs, psets):
                assert (a, b) == (idx[(f, i)], idx[(f, j)])
            seen += pairs
        assert seen == mine
    assert sorted(allp) == [(f, i, j) for f in range(total) for i in range(ncams - 1) for j in range(i + 1, ncams)]
    per = [len(shard.pairs_of_rank(r, world, ncams, total)) for r in range(world)]
    assert max(per) - min(per) <= 0 if total % world == 0 else True
    assert [shard.pair_slot(ncams, i, j) for i in range(ncams - 1) for j in range(i + 1, ncams)] == list(range(ncams * (ncams - 1) // 2))


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2`, typed the way the driver types `--gpus 1`: the parent spawns the ranks itself (before any
    GPU call), they rendezvous (gloo here: no GPU in this container), rank 0's line is relayed, and a failing rank makes the
    parent exit non-zero."""
    import json
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--spawn-selftest", "-1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["ranks_seen"] == 2 and line["sum"] == 3.0 and line["backend"] == "gloo" and line["n_gpus"] == 2
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--spawn-selftest", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "ranks failed" in p.stderr
