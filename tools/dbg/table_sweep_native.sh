#!/bin/bash
# workers x batch sweep under the native scheduler (no GIL between the workers: more, narrower workers are an option the Python unit loop
# did not have), one contiguous 8192-row block, same proofs in flight (workers x batch = 128) unless noted
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r04h
for cfg in "4 32" "5 24" "6 20" "8 16" "3 48" "2 64" "6 24" "4 32"; do
  set -- $cfg
  python bench.py --steps 8 --warmup 1 --config2-leaves 0 --degree-sweep "" --no-leaves-leg --no-cpu-baseline --no-verify --workers $1 --table-batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('native workers $1 batch $2:', round(d['value'],1), 'proofs/s', d['config']['work_plan_waves'], round(d['config']['device_memory_used_bytes']/1e9), 'GB')"
done
