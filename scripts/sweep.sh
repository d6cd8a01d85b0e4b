#!/bin/bash
# bench value for a list of environment settings, the whole list twice (forward, then backward):  bash scripts/sweep.sh <outdir> VAR=V1 VAR=V2 ...
O=gpurun_out/${1:-sweep}; mkdir -p $O; shift
one() {
  env "$1" python bench.py --no-cpu --no-latency --no-staging --host-cores 0 --repeats 2 --iso-jobs 4 > $O/x.json 2>$O/x.err
  python3 -c "
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1])
print('%-34s %8.1f (%.1f - %.1f)'%('$1', d['value'], d['value_min'], d['value_max']), {k:v['us'] for k,v in d['roofline_all_kernels_isolated'].items()})"
}
for kv in "$@"; do one $kv; done
for ((i=$#; i>0; i--)); do one ${!i}; done
