#!/bin/bash
# SQ counter passes for one kernel under environment settings; run on the GPU box:  bash scripts/pmc_env.sh <outdir> <kernel> VAR=VALUE ...
O=gpurun_out/${1:-pmce}; K=$2; shift; shift
mkdir -p $O
for kv in "$@"; do export "$kv"; done
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
ARGS="bench.py --steps 2 --warmup 1 --repeats 1 --min-region-s 0 --host-cores 0 --no-cpu --no-latency --no-staging --slots 1 --frames 32 --iso-jobs 1"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 $ARGS > $O/p$i.log 2>&1 || echo "pass $i failed"
    f=$(find $O/p$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 scripts/pmc_summary.py $f | grep -A10 "^$K " | grep -B0 -A10 "^$K " | sed -n 1,9p
    rm -rf $O/p$i
done
