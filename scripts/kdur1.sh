#!/bin/bash
# like kdur.sh with ONE slot (one job on the GPU at a time): kernel durations beside the job's own copies
O=gpurun_out/${1:-kdur1}; mkdir -p $O; shift
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/p -o t -- python3 bench.py --no-cpu --no-latency --no-staging --host-cores 0 --repeats 1 --iso-jobs 0 --slots 1 --frames 32 --min-region-s 0.3 > $O/log 2>&1
python3 - <<PY
import csv,statistics,glob,json
f=glob.glob('$O/p/**/*kernel_trace.csv',recursive=True)[0]
d={}
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ","").replace("mcorb::","")
    d.setdefault(n,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for l in open('$O/log'):
    if l.startswith('{'): print("value", json.loads(l)["value"])
for n,v in sorted(d.items(),key=lambda x:-sum(x[1])):
    print("%-26s n %5d  min %7.1f  median %7.1f  max %8.1f us"%(n,len(v),min(v),statistics.median(v),max(v)))
PY
rm -rf $O/p
