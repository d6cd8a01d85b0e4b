#!/bin/bash
# the 4-host-core leg with more jobs in flight (512-image jobs):  bash scripts/hostcores_slots.sh <outdir>
O=gpurun_out/${1:-hcs}; mkdir -p $O
for cfg in "4 512" "5 640" "6 768" "4 512"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --no-cpu --no-latency --no-staging --no-extra-legs --repeats 1 --iso-jobs 0 --slots $1 --frames $2 > $O/s$1.json 2> $O/s$1.err || { echo "slots $1 failed"; tail -3 $O/s$1.err; exit 1; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/s$1.json') if l.startswith('{')][-1])
h=d['value_host_cores']
print('slots $1: value %.0f  value_host_cores %.0f (%.2f)' % (d['value'], h['value'], h['value']/d['value']))"
done
