#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/fetch_calib.hip); run on the GPU box from the repo root:
#   bash scripts/fetch_calib.sh gpurun_out/calib
set -e
O=${1:-gpurun_out/calib}
mkdir -p $O
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $ROOT
timeout -k 10 200 $ROOT/tools/_build/fetch_calib 1024 > $O/plain.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- $ROOT/tools/_build/fetch_calib 1024 > $O/f.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- $ROOT/tools/_build/fetch_calib 1024 > $O/w.log 2>&1
python3 scripts/pmc_summary.py $(find $O/f -name '*counter_collection.csv' | head -1) > $O/fetch.txt
python3 scripts/pmc_summary.py $(find $O/w -name '*counter_collection.csv' | head -1) > $O/write.txt
cat $O/plain.txt $O/fetch.txt $O/write.txt
