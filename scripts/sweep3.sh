#!/bin/bash
# images per job x slots, fused 720p4 path:  bash scripts/sweep3.sh <outdir>
O=gpurun_out/${1:-sweep3}; mkdir -p $O
for cfg in "4 512" "3 768" "4 1024" "2 512" "3 576" "4 512"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --no-cpu --no-latency --no-staging --no-extra-legs --host-cores 0 --repeats 2 --iso-jobs 0 --slots $1 --frames $2 > $O/s$1_f$2.json 2> $O/s$1_f$2.err || { echo "slots $1 frames $2 failed"; tail -3 $O/s$1_f$2.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads([l for l in open('$O/s$1_f$2.json') if l.startswith('{')][-1])
print('slots $1 frames/step $2 images/job', $2*4//$1, 'value %.0f (%.0f - %.0f)' % (d['value'], d['value_min'], d['value_max']))"
done
