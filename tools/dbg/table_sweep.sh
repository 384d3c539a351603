#!/bin/bash
# workers x batch x subtree size x group rows sweep of the table workload (one GPU, one contiguous 6144-row block): framework proofs/s
for cfg in "4 32 64 128" "4 32 64 256" "4 32 64 512" "4 32 128 256" "4 32 32 256" "3 48 64 192" "3 48 64 384" "5 24 64 192" "2 64 64 256" "4 40 64 320"; do
  set -- $cfg
  python bench.py --steps 6 --warmup 1 --config2-leaves 0 --degree-sweep "" --no-leaves-leg --no-cpu-baseline --no-verify --workers $1 --table-batch $2 --subtree $3 --group-rows $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('workers $1 batch $2 subtree $3 group $4:', round(d['value'],1), 'proofs/s', d['config']['work_plan_waves'], round(d['config']['device_memory_used_bytes']/1e9), 'GB')"
done
