#!/bin/bash
# matrix-pipe counters of k_knn2 (one slot); run on the GPU box from the repo root
O=gpurun_out/${1:-pmcm}; mkdir -p $O
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
ARGS="bench.py --steps 2 --warmup 1 --repeats 1 --min-region-s 0 --host-cores 0 --no-cpu --no-latency --no-staging --slots 1 --frames 32 --iso-jobs 1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/p -o p -- python3 $ARGS > $O/p.log 2>&1 || tail -5 $O/p.log
python3 scripts/pmc_summary.py $(find $O/p -name '*counter_collection.csv' | head -1) > $O/mfma.txt
grep -A8 "^k_knn2 " $O/mfma.txt
rm -rf $O/p
