#!/bin/bash
# does the number of hardware queues HIP maps its streams onto (GPU_MAX_HW_QUEUES, default 4) bound the workers? one contiguous
# 8192-row block under the native scheduler, 4 / 6 workers with 4 (default) and 8 hardware queues
for cfg in "4 32 4" "4 32 8" "6 24 8" "6 20 8"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 python bench.py --steps 8 --warmup 1 --config2-leaves 0 --degree-sweep "" --no-leaves-leg --no-cpu-baseline --no-verify --workers $1 --table-batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES=$3 workers $1 batch $2:', round(d['value'],1), 'proofs/s', round(d['config']['device_memory_used_bytes']/1e9), 'GB')"
done
