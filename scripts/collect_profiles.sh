#!/bin/bash
# Collects the round's bench line, rocprofv3 kernel statistics (default and one-slot), the FETCH_SIZE / WRITE_SIZE
# counter passes, the SQ counter passes, the FETCH_SIZE calibration and the BoW rate into gpurun_out/$1 (run on the GPU box
# from the repo root).  Install the summaries into profiles/ afterwards: python3 scripts/install_profiles.py gpurun_out/$1 r02
set -e
O=gpurun_out/${1:-prof}
mkdir -p $O
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $ROOT
export MCORB_GRAPH=0   # per-kernel events and one launch per kernel everywhere in this collection (single-slot rigs would replay a graph)
ONE="--steps 3 --warmup 1 --repeats 1 --min-region-s 0 --host-cores 0 --no-cpu --no-latency --no-staging --slots 1 --frames 32 --iso-jobs 1"
timeout -k 10 400 python3 bench.py --no-extra-legs 2>$O/bench.err | tail -1 > $O/bench_default.json   # provisional (install_profiles.py reads the launch size from it)
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_def -o d -- python3 bench.py --no-cpu --no-latency --no-staging --repeats 1 > $O/p_def.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_1s -o s -- python3 bench.py $ONE > $O/p_1s.log 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_f -o f -- python3 bench.py $ONE > $O/p_f.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_w -o w -- python3 bench.py $ONE > $O/p_w.log 2>&1
python3 scripts/pmc_summary.py $(find $O/p_f -name '*counter_collection.csv' | head -1) > $O/fetch.txt
python3 scripts/pmc_summary.py $(find $O/p_w -name '*counter_collection.csv' | head -1) > $O/write.txt
echo "traffic done"
bash scripts/pmc_kernel.sh $O/sq > $O/sq.txt 2>&1 || true
bash scripts/fetch_calib.sh $O/calib > $O/calib.txt 2>&1 || true
timeout -k 10 200 python3 scripts/bow_rate.py 2>/dev/null | tail -1 > $O/bow_rate.json || true
timeout -k 10 200 python3 scripts/lf_rate.py 2>/dev/null | tail -1 > $O/lf_rate.json || true
timeout -k 10 200 python3 scripts/n1_rate.py 2>/dev/null | tail -1 > $O/n1_rate.json || true
MCORB_GRAPH=1 timeout -k 10 200 python3 scripts/latency.py 2>/dev/null | tail -1 > $O/latency.json || true
MCORB_GRAPH=0 timeout -k 10 200 python3 scripts/latency.py 2>/dev/null | tail -1 > $O/latency_nograph.json || true
MCORB_SELECT=host timeout -k 10 200 python3 scripts/latency.py 2>/dev/null | tail -1 > $O/latency_hostselect.json || true
timeout -k 10 200 tools/_build/fp4_probe > $O/fp4_probe.txt 2>&1 || true
timeout -k 10 100 tools/_build/lat_probe > $O/lat_probe.txt 2>&1 || true
(unset MCORB_GRAPH; bash scripts/lat_trace.sh ${1:-prof}_lt 2>&1 | grep -v rocprofv3 > $O/latency_trace.txt) || true
timeout -k 10 200 tools/_build/valu_rates > $O/valu_rates.txt 2>&1 || true
cp $(find $O/p_def -name '*kernel_stats.csv' | head -1) $O/stats_default.csv
cp $(find $O/p_1s -name '*kernel_stats.csv' | head -1) $O/stats_1slot.csv
# the bench line that is kept: run again with traffic.json regenerated from THIS run's counters, so that roofline.traffic and
# roofline.valu are filled in (bench.py drops them when the kernel sources' sha does not match)
python3 scripts/install_profiles.py $O ${2:-r04} > /dev/null
unset MCORB_GRAPH
timeout -k 10 400 python3 bench.py 2>$O/bench.err | tail -1 > $O/bench_default.json
# the raw traces stay on the box (gpurun copies back at most 64 MiB): summaries only
rm -rf $O/p_def $O/p_1s $O/p_f $O/p_w $O/calib $O/sq/p[0-9] $O/sq/p[0-9]/ 2>/dev/null
find $O -name '*.csv' -size +2M -delete
du -sh $O
echo collected into $O
