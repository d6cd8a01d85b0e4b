#!/bin/bash
# Is the bench being CPU-throttled by the box's cgroup quota?  Prints cpu.max and the cpu.stat delta around one bench run.
O=gpurun_out/${1:-thr}; mkdir -p $O
CG=/sys/fs/cgroup
cat $CG/cpu.max 2>/dev/null || cat $CG/cpu/cpu.cfs_quota_us $CG/cpu/cpu.cfs_period_us 2>/dev/null
nproc
cat $CG/cpu.stat 2>/dev/null > $O/stat0 || cat $CG/cpu/cpu.stat > $O/stat0
shift
python bench.py --no-cpu --no-latency --no-staging --repeats 3 --iso-jobs 0 "$@" > $O/b.json 2>$O/b.err
cat $CG/cpu.stat 2>/dev/null > $O/stat1 || cat $CG/cpu/cpu.stat > $O/stat1
paste $O/stat0 $O/stat1
python3 -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('repeats'))"
