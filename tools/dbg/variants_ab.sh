#!/bin/bash
# round 5, table build (one block of 5120 rows, side legs off): the proof-of-work sweep width (MP2G_POW_LANES: lanes of one launch
# over all proofs of a batch; 1048576 = rounds 1-4, 262144 = the default now) and workers x batch at the same or more memory
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
QUIET="--no-leaves-leg --no-verify --config2-leaves 0 --degree-sweep= --no-cpu-baseline"
one() { python3 $R/bench.py --steps 5 --warmup 2 --rows 1024 $QUIET "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   ', round(d['value'],1), 'proofs/s', d['config']['device_memory_used_bytes']>>30, 'GiB')"; }
{
for rep in 1 2; do for lanes in 1048576 262144 131072 524288; do echo "MP2G_POW_LANES=$lanes"; MP2G_POW_LANES=$lanes one; done; done
for cfg in "4 32" "3 48" "4 48" "3 64" "5 32" "4 40"; do set -- $cfg; echo "workers $1 batch $2"; one --workers $1 --table-batch $2; done
} 2>&1 | tee $O/variants_ab.txt
