#!/bin/bash
# bench value + cgroup throttling for a few host-side settings (driver wait mode, selection worker count)
O=gpurun_out/${1:-thrm}; mkdir -p $O
CG=/sys/fs/cgroup
run() {
  local tag=$1; shift
  local a=$(grep -E "^(nr_throttled|throttled_usec|usage_usec)" $CG/cpu.stat | awk '{print $2}' | tr '\n' ' ')
  env "$@" python bench.py --no-cpu --no-latency --no-staging --repeats 3 --iso-jobs 0 $BARGS > $O/$tag.json 2>$O/$tag.err
  local b=$(grep -E "^(nr_throttled|throttled_usec|usage_usec)" $CG/cpu.stat | awk '{print $2}' | tr '\n' ' ')
  python3 -c "
import json,sys
d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); a=[int(x) for x in '$a'.split()]; b=[int(x) for x in '$b'.split()]
print('%-28s value %8.1f  cpu %.1f s  throttled periods %d  throttled %.2f s'%('$tag', d['value'], (b[0]-a[0])/1e6, b[1]-a[1], (b[2]-a[2])/1e6), d.get('value_repeats') or '')"
}
if [ -n "$2" ]; then source $2; exit 0; fi
run default MCORB_X=0
run block MCORB_SYNC=block
run ht8 MCORB_HOST_THREADS=8
