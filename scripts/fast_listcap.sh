#!/bin/bash
# k_fast_cells work-list capacity (LDS per wave -> waves per CU) re-swept on the current kernel:  bash scripts/fast_listcap.sh <outdir>
O=gpurun_out/${1:-flc}; mkdir -p $O
for n in ${CAPS:-768 640 704 832 896 768}; do
  export MCORB_FAST_LISTCAP=$n
  timeout -k 10 300 python3 bench.py --no-cpu --no-latency --no-staging --no-extra-legs --host-cores 0 --repeats 1 > $O/b$n.json 2> $O/b$n.err || { echo "cap $n failed"; tail -3 $O/b$n.err; exit 1; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/b$n.json') if l.startswith('{')][-1])
print('list cap $n: k_fast_cells isolated %.1f us per 512 images (%.1f per 128), value %.0f' % (d['kernel_us_per_launch_isolated']['k_fast_cells'], d['kernel_us_per_launch_isolated']['k_fast_cells']/4, d['value']))"
done
