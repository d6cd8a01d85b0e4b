#!/bin/bash
# fewer jobs on the GPU at once:  bash scripts/gpujobs_sweep2.sh <outdir>
O=gpurun_out/${1:-gj}; mkdir -p $O
for cfg in "4 0" "4 3" "5 3" "3 0" "4 2" "4 0"; do
  set -- $cfg
  export MCORB_GPU_JOBS=$2
  timeout -k 10 300 python3 bench.py --no-cpu --no-latency --no-staging --no-extra-legs --host-cores 0 --repeats 2 --iso-jobs 0 --slots $1 --frames $(($1 * 128)) > $O/s$1_g$2.json 2> $O/s$1_g$2.err || { echo "slots $1 jobs $2 failed"; tail -3 $O/s$1_g$2.err; exit 1; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/s$1_g$2.json') if l.startswith('{')][-1])
print('slots $1 gpu_jobs $2: value %.0f (%.0f - %.0f)' % (d['value'], d['value_min'], d['value_max']))"
done
