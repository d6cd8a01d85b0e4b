#!/bin/bash
# the sharded bench path (world size 1 over RCCL) against the single-GPU path on the same box
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29518
python bench.py --force-dist --no-cpu --no-latency --no-staging --repeats 3 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dist ', d['value'], d['value_min'], d['value_max'], d['config']['frames_per_rank_per_step'], d['config']['slots'], d['ms_per_step'], d.get('exchange'))"
python bench.py --no-cpu --no-latency --no-staging --repeats 3 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused', d['value'], d['value_min'], d['value_max'])"
