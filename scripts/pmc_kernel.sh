#!/bin/bash
# SQ counter passes for the mcorb kernels (1 slot, 128 images per launch); run on the GPU box from the repo root:
#   bash scripts/pmc_kernel.sh gpurun_out/pmc
set -e
O=${1:-gpurun_out/pmc}
mkdir -p $O
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $ROOT
ARGS="bench.py --steps 2 --warmup 1 --repeats 1 --min-region-s 0 --host-cores 0 --no-cpu --no-latency --no-staging --slots 1 --frames 32 --iso-jobs 1"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 $ARGS > $O/p$i.log 2>&1 || echo "pass $i failed (see $O/p$i.log)"
    f=$(find $O/p$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 scripts/pmc_summary.py $f > $O/p$i.txt
done
cat $O/p*.txt
