#!/bin/bash
# k_knn2 with parts of its loop removed (MCORB_KNN_DBG bit 0: no LDS fill / barrier, bit 2: no top-2 folds, bit 4: no barrier;
# results are wrong, only the durations mean something): isolated kernel time per variant from the bench's HIP events
O=gpurun_out/${1:-knndbg}; mkdir -p $O
for d in ${KNN_DBG_LIST:-0 16 1 4 5 0}; do
  MCORB_KNN_DBG=$d python bench.py --no-cpu --no-latency --no-staging --host-cores 0 --repeats 1 --min-region-s 0.2 > $O/d$d.json 2>$O/d$d.err
  python3 -c "
import json
d=json.loads(open('$O/d$d.json').read().strip().splitlines()[-1])
print('DBG=$d', d['value'], {k:v['us'] for k,v in d['roofline_all_kernels_isolated'].items()})"
done
