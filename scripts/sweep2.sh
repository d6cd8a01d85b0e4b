#!/bin/bash
# one bench line per configuration ("ENV=VAL ... -- bench args"), in the order given
#   bash scripts/sweep2.sh <outdir> "cfg1" "cfg2" ...
O=gpurun_out/${1:-sw2}; mkdir -p $O; shift
i=0
for cfg in "$@"; do
  i=$((i+1)); envs=${cfg%%--*}; args=${cfg#*--}
  env $envs python bench.py --no-cpu --no-latency --no-staging --host-cores 0 --no-extra-legs --repeats 3 $args > $O/c$i.json 2>$O/c$i.err
  python3 -c "
import json
d=json.loads(open('$O/c$i.json').read().strip().splitlines()[-1])
print('%-50s %8.1f (%.1f - %.1f)'%('''$cfg'''[:50], d['value'], d['value_min'], d['value_max']))"
done
