#!/bin/bash
# A/B on one box in ABBA order (a setting that always runs second would inherit whatever the first run warmed up):
#   bash scripts/abba.sh <outdir> VAR=VALUE      -> default, VAR, VAR, default, default, VAR
O=gpurun_out/${1:-abba}; mkdir -p $O; KV=$2
one() {
  local tag=$1; shift
  env "$@" python bench.py --no-cpu --no-latency --no-staging --host-cores 0 --repeats 3 > $O/$tag.json 2>$O/$tag.err
  python3 -c "
import json
d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1])
print('%-34s %8.1f (%.1f - %.1f)'%('$tag', d['value'], d['value_min'], d['value_max']), {k:v['us'] for k,v in d['roofline_all_kernels_isolated'].items()})"
}
one A1 MCORB_X=0; one B1 $KV; one B2 $KV; one A2 MCORB_X=0; one A3 MCORB_X=0; one B3 $KV
