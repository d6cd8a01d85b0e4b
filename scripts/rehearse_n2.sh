#!/bin/bash
# N = 2 rehearsal on a ONE-GPU box: two ranks share device 0, the collectives run over gloo on host tensors (RCCL refuses two ranks
# on one device), everything else -- placement, export, external / pair-list match jobs, gather, merge -- is the N > 1 code path
# of bench.py.  Both exchanges; the tracks every rank ends up with are compared with the fused single-process path's.
#   bash scripts/rehearse_n2.sh <outdir>
O=gpurun_out/${1:-n2}; mkdir -p $O
COMMON="--steps 2 --warmup 1 --repeats 1 --min-region-s 0 --no-cpu --no-latency --no-staging --host-cores 0 --iso-jobs 1"
python bench.py $COMMON --frames 16 --slots 1 --dump-tracks $O/fused.npz > $O/fused.json 2>$O/fused.err || { tail -3 $O/fused.err; exit 1; }
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $(python3 -c 'import socket; s=socket.socket(); s.bind(("127.0.0.1",0)); print(s.getsockname()[1])') bench.py --gpus 2 --single-device \
    --dist-backend gloo $COMMON --frames 8 --slots 4 --dump-tracks $O/a2a.npz > $O/a2a.json 2>$O/a2a.err || { tail -5 $O/a2a.err; exit 1; }
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $(python3 -c 'import socket; s=socket.socket(); s.bind(("127.0.0.1",0)); print(s.getsockname()[1])') bench.py --gpus 2 --single-device \
    --dist-backend gloo --exchange allgather --partition pairs $COMMON --frames 8 --slots 2 --dump-tracks $O/pairs.npz > $O/pairs.json 2>$O/pairs.err || { tail -5 $O/pairs.err; exit 1; }
python3 - <<PY
import numpy as np, json
f = np.load("$O/fused.npz")
ref = {int(fr): f["t%d" % i] for i, fr in enumerate(f["frames"])}
ok, n = True, 0
for name in ("a2a_r0", "a2a_r1", "pairs"):
    d = np.load("$O/%s.npz" % name)
    for i, fr in enumerate(d["frames"]):
        same = np.array_equal(d["t%d" % i], ref[int(fr)])
        ok &= same; n += 1
        if not same: print(name, "frame", int(fr), "differs")
for name in ("fused", "a2a", "pairs"):
    j = json.loads([l for l in open("$O/%s.json" % name) if l.startswith("{")][-1])
    print("%-6s n_gpus %d value %.0f frames/s  (%s)" % (name, j["n_gpus"], j["value"], j["config"]["sharding"][:60]))
print("N = 2 rehearsal: %d frame track tables compared with the fused path: %s" % (n, "all identical" if ok else "MISMATCH"))
PY
