#!/bin/bash
# workers x batch sweep of the table workload (one GPU): framework proofs/s per configuration
for cfg in "4 32 64" "6 32 64" "8 16 64" "3 64 128" "2 64 128" "4 48 64" "6 24 32"; do
  set -- $cfg
  python bench.py --steps 2 --warmup 1 --no-leaves-leg --no-verify --workers $1 --table-batch $2 --subtree $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('workers $1 batch $2 subtree $3:', round(d['value'],1), 'proofs/s', round(d['ms_per_step']), 'ms/step')"
done
