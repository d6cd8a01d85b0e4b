#!/bin/bash
# GPU tests + a short bench line (per-kernel isolated times) into gpurun_out/$1; run on the GPU box from the repo root.
O=gpurun_out/${1:-quick}
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/t.log 2>&1; tail -3 $O/t.log
python bench.py --no-cpu --no-latency --no-staging --repeats 3 > $O/b.json 2>$O/b.err && python3 -c "
import json,sys
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print(d['value'], {k:v['us'] for k,v in d['roofline_all_kernels_isolated'].items()})"
