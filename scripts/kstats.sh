#!/bin/bash
# rocprofv3 kernel statistics of the isolated configuration (1 slot, 128 images per launch) + MFMA / issue counters
#   bash scripts/kstats.sh gpurun_out/ks
set -e
O=${1:-gpurun_out/ks}
mkdir -p $O
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $ROOT
ARGS="bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu --no-latency --no-staging --slots 1 --frames 32 --iso-jobs 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python3 $ARGS > $O/st.log 2>&1
f=$(find $O/st -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && cut -d, -f1-8 $O/kernel_stats.csv | head -20
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pm -o p -- python3 $ARGS > $O/pm.log 2>&1 || true
f=$(find $O/pm -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 scripts/pmc_summary.py $f > $O/pmc.txt && grep -A9 "^k_knn2 \|^k_expand" $O/pmc.txt
