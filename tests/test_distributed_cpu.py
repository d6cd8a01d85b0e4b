"""world_size-2 gloo test of the multi-GPU data path (camera sharding -> all-gather of
per-camera descriptors -> per-frame matching), with the CPU oracle standing in for the GPU
stages.  What is under test is the placement / indexing logic bench.py uses for N > 1
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


@pytest.mark.parametrize("world", [2])
def test_sharded_pipeline_equals_single_process(world):
    import torch.multiprocessing as mp
    import oracle_lib as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
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
