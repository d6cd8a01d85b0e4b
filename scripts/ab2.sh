#!/bin/bash
# A/B of two bench configurations on one box in ABBA order; a configuration = "ENV=VAL ... -- bench args"
#   bash scripts/ab2.sh <outdir> "A-config" "B-config"      e.g.  "MCORB_SELECT=host --" "GPU_MAX_HW_QUEUES=16 -- --slots 8"
O=gpurun_out/${1:-ab2}; mkdir -p $O
one() {
  local tag=$1 cfg=$2
  local envs=${cfg%%--*} args=${cfg#*--}
  env $envs python bench.py --no-cpu --no-latency --no-staging --host-cores 0 --no-extra-legs --repeats 3 $args > $O/$tag.json 2>$O/$tag.err
  python3 -c "
import json
d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1])
print('%-4s %-44s %8.1f (%.1f - %.1f)'%('$tag', '''$cfg'''[:44], d['value'], d['value_min'], d['value_max']))"
}
one A1 "$2"; one B1 "$3"; one B2 "$3"; one A2 "$2"; one A3 "$2"; one B3 "$3"
