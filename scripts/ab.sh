#!/bin/bash
# A/B on one box: bench value + isolated kernel times with and without an environment setting.
#   bash scripts/ab.sh <outdir> VAR=VALUE [VAR=VALUE ...]     (each setting is compared against the default, interleaved twice)
O=gpurun_out/${1:-ab}; mkdir -p $O; shift
one() {
  local tag=$1; shift
  env "$@" python bench.py --no-cpu --no-latency --no-staging --repeats 3 > $O/$tag.json 2>$O/$tag.err
  python3 -c "
import json
d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1])
print('%-34s %8.1f'%('$tag', d['value']), {k:v['us'] for k,v in d['roofline_all_kernels_isolated'].items()})"
}
for round in 1 2; do
  one default_$round MCORB_X=0
  for kv in "$@"; do one "${kv}_$round" "$kv"; done
done
