#!/bin/bash
# the 4-host-core leg with different worker-pool sizes:  bash scripts/hostcores_sweep.sh <outdir>
O=gpurun_out/${1:-hc}; mkdir -p $O
for t in default 1 3 4 6; do
  if [ $t = default ]; then unset MCORB_HOST_THREADS; else export MCORB_HOST_THREADS=$t; fi
  timeout -k 10 300 python3 bench.py --no-cpu --no-latency --no-staging --no-extra-legs --repeats 1 --iso-jobs 0 > $O/t$t.json 2> $O/t$t.err || { echo "threads $t failed"; tail -3 $O/t$t.err; exit 1; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/t$t.json') if l.startswith('{')][-1])
h=d['value_host_cores']
print('MCORB_HOST_THREADS=$t value %.0f  value_host_cores %.0f (%.2f) workers %s' % (d['value'], h['value'], h['value']/d['value'], h.get('engine_workers')))"
done
