#!/bin/bash
# round 5: units of a wave shrinking towards its end (csrc/forest.hip) against units of one size (MP2G_FOREST_FIXED_UNITS=1):
# the driver's block (20 480 rows) twice each, then a 2^16-row block with the shrinking units (fixed: profiles/r05/bench_r05_block_2p16_rows.json)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
QUIET="--no-leaves-leg --no-verify --config2-leaves 0 --degree-sweep= --no-cpu-baseline"
one() { python3 $R/bench.py --warmup 2 --rows 1024 $QUIET "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   ', round(d['value'],1), 'proofs/s', d['table_rows_total'], 'rows')"; }
{
for rep in 1 2; do
  echo "fixed units, 20480 rows"; MP2G_FOREST_FIXED_UNITS=1 one --steps 20
  echo "shrinking units, 20480 rows"; one --steps 20
done
echo "shrinking units, 65536 rows"; one --steps 64
} 2>&1 | tee $O/units_ab.txt
