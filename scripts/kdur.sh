#!/bin/bash
# median / min / max duration per kernel from a rocprofv3 kernel trace of the default bench (6 slots) -- run on the GPU box:
#   bash scripts/kdur.sh <outdir> [VAR=VALUE ...]
O=gpurun_out/${1:-kdur}; mkdir -p $O; shift
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/p -o t -- python3 bench.py --no-cpu --no-latency --no-staging --repeats 1 --iso-jobs 0 ${BARGS} > $O/log 2>&1
python3 - <<PY
import csv,statistics,glob
f=glob.glob('$O/p/**/*kernel_trace.csv',recursive=True)[0]
d={}
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ","").replace("mcorb::","")
    d.setdefault(n,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for n,v in sorted(d.items(),key=lambda x:-sum(x[1])):
    print("%-26s n %5d  min %7.1f  median %7.1f  max %8.1f  sum %9.1f us"%(n,len(v),min(v),statistics.median(v),max(v),sum(v)))
PY
rm -rf $O/p
