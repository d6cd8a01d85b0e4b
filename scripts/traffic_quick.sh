#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel (one slot, 128 images per launch) into gpurun_out/$1
O=gpurun_out/${1:-tq}; mkdir -p $O
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
ONE="--steps 3 --warmup 1 --repeats 1 --no-cpu --no-latency --no-staging --slots 1 --frames 32 --iso-jobs 1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_f -o f -- python3 bench.py $ONE > $O/p_f.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_w -o w -- python3 bench.py $ONE > $O/p_w.log 2>&1 &&
python3 scripts/pmc_summary.py $(find $O/p_f -name '*counter_collection.csv' | head -1) > $O/fetch.txt &&
python3 scripts/pmc_summary.py $(find $O/p_w -name '*counter_collection.csv' | head -1) > $O/write.txt
rm -rf $O/p_f $O/p_w
grep -A1 "^k_" $O/fetch.txt | grep -v "^--"
