#!/bin/bash
# world-size-1 rehearsal of the sharded bench path over RCCL for several slot-group shapes (GPU box), fused path last
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29512
mkdir -p gpurun_out/dm
run() {
  local g=$1 s=$2; shift 2
  MCORB_BENCH_GROUPS=$g env "$@" python bench.py --force-dist --slots $s --no-cpu --no-latency --no-staging --repeats 2 2>gpurun_out/dm/err_${g}_$s.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('G=$g S=$s $*', d['value'], d['config']['frames_per_rank_per_step'], d['ms_per_step'])"
}
run 6 12 X=0
run 6 18 X=0
run 5 10 X=0
run 8 16 X=0
run 4 12 X=0
python bench.py --no-cpu --no-latency --no-staging --repeats 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fused', d['value'])"
