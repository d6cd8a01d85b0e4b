#!/bin/bash
# Collects the round's bench line, rocprofv3 kernel statistics (default and one-slot), the FETCH_SIZE / WRITE_SIZE
# counter passes, the PCIe-inclusive rates and the single-frame latency into gpurun_out/$1 (run on the GPU box from the
# repo root).  Copy the summaries into profiles/ afterwards (see profiles/README.md).
set -e
O=gpurun_out/${1:-prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py 2>$O/bench.err | tail -1 > $O/bench_default.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_def -o d -- python3 bench.py --no-cpu --no-latency > $O/p_def.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_1s -o s -- python3 bench.py --slots 1 --frames 32 --no-cpu --no-latency > $O/p_1s.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_f -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-latency --slots 1 --frames 32 > $O/p_f.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_w -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-latency --slots 1 --frames 32 > $O/p_w.log 2>&1
python3 scripts/pmc_summary.py $O/p_f/f_counter_collection.csv > $O/fetch.txt
python3 scripts/pmc_summary.py $O/p_w/w_counter_collection.csv > $O/write.txt
timeout -k 10 200 python3 scripts/upload_rate.py > $O/upload_rate.txt 2>&1
timeout -k 10 100 python3 scripts/latency.py > $O/latency.json 2>&1
timeout -k 10 200 python3 scripts/bow_rate.py 2>/dev/null | tail -1 > $O/bow_rate.json
echo collected into $O
