#!/bin/bash
# A/B of whole-library build variants on one box: mc-slam_amd/_variants/libmcorb_<name>.so are swapped in one after the other
#   bash scripts/variant_ab.sh <outdir> <name> [<name> ...]
O=gpurun_out/${1:-vab}; mkdir -p $O; shift
cp mc-slam_amd/libmcorb.so $O/orig.so
for rep in 1 2; do
for v in "$@"; do
  cp mc-slam_amd/_variants/libmcorb_$v.so mc-slam_amd/libmcorb.so
  timeout -k 10 300 python3 bench.py --no-cpu --no-latency --no-staging --no-extra-legs --host-cores 0 --repeats 1 > $O/$v$rep.json 2> $O/$v$rep.err || { echo "$v failed"; tail -3 $O/$v$rep.err; cp $O/orig.so mc-slam_amd/libmcorb.so; exit 1; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/$v$rep.json') if l.startswith('{')][-1])
k=d['kernel_us_per_launch_isolated']
print('%-20s value %.0f | isolated us per 512 images: ' % ('$v', d['value']) + ', '.join('%s %.0f' % (a.replace('k_',''), b) for a, b in k.items()))"
done
done
cp $O/orig.so mc-slam_amd/libmcorb.so; rm $O/orig.so
