#!/bin/bash
# Round 6, VERDICT r05 item 7: the small levels of a tree in one launch per <= 6 levels (csrc/merkle.hip merkle_subtree_wave_kernel,
# -DMP2G_EXPERIMENT_MERKLE_FUSED: bash tools/dbg/build_variant.sh merkle_fused "-DMP2G_EXPERIMENT_MERKLE_FUSED" merkle.hip) against
# one launch per level (the product): a lone commitment, a lone proof, the table block. (The committed run, profiles/r06/
# merkle_fused_ab.txt, was made while the fused form was the default and the per-level form the variant: same two libraries.)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
QUIET="--no-leaves-leg --no-verify --config2-leaves 0 --degree-sweep= --no-cpu-baseline"
: > $O/merkle_fused_ab.txt
for mode in fused per_level fused per_level; do
  if [ $mode = fused ]; then export MP2G_LIB=$R/build_dbg/merkle_fused/libmp2gpu.so; else unset MP2G_LIB; fi
  echo "== $mode" >> $O/merkle_fused_ab.txt
  python3 $R/tools/dbg/merkle_fused_ab.py >> $O/merkle_fused_ab.txt 2>> $O/merkle_fused_ab.err
  for b in 1 4; do
    python3 $R/bench.py --workload leaves --batch $b --streams 1 --steps 20 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 > /tmp/lp.json
    python3 -c "import json; d=json.load(open('/tmp/lp.json')); print('lone proofs B=$b:', round(d['ms_per_step'],2), 'ms per step;', {k: round(sum(v.values()),2) for k, v in d['stage_ms'].items()})" >> $O/merkle_fused_ab.txt
  done
  python3 $R/bench.py --steps 20 --warmup 5 --rows 1024 $QUIET 2>> $O/merkle_fused_ab.err | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('table block:', round(d['value'],1), 'proofs/s on', d['config']['rows_per_rank'], 'rows')" >> $O/merkle_fused_ab.txt
done
cat $O/merkle_fused_ab.txt
