#!/bin/bash
# slots x jobs admitted to the GPU at once (512-image jobs), all cores and 4 cores:  bash scripts/gpujobs_sweep.sh <outdir>
O=gpurun_out/${1:-gj}; mkdir -p $O
for cfg in "4 0" "6 4" "5 4" "8 4" "6 3" "6 5" "4 0" "6 4"; do
  set -- $cfg
  export MCORB_GPU_JOBS=$2
  timeout -k 10 300 python3 bench.py --no-cpu --no-latency --no-staging --no-extra-legs --repeats 1 --iso-jobs 0 --slots $1 --frames $(($1 * 128)) > $O/s$1_g$2.json 2> $O/s$1_g$2.err || { echo "slots $1 jobs $2 failed"; tail -3 $O/s$1_g$2.err; exit 1; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/s$1_g$2.json') if l.startswith('{')][-1])
h=d['value_host_cores']
print('slots $1 gpu_jobs $2: value %.0f  value_host_cores %.0f (%.2f)' % (d['value'], h['value'], h['value']/d['value']))"
done
