#!/bin/bash
# Round 6: the leaf sponge with the next chunk's limbs requested before the current permutation (-DMP2G_EXPERIMENT_LEAF_PREFETCH:
# 127 VGPRs under amdgpu_waves_per_eu(4, 4), 190 spill instructions; 161 VGPRs = three waves without the cap) against the product:
# the kernel alone at one and at four generations of blocks, and the table block.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
QUIET="--no-verify --config2-leaves 0 --degree-sweep= --no-cpu-baseline"
: > $O/leaf_prefetch_ab.txt
for mode in product prefetch product prefetch; do
  if [ $mode = prefetch ]; then export MP2G_LIB=$R/build_dbg/leafpf/libmp2gpu.so; else unset MP2G_LIB; fi
  echo "== $mode" >> $O/leaf_prefetch_ab.txt
  python3 $R/tools/dbg/merkle_fused_ab.py 2>> $O/leaf_prefetch_ab.err | head -1 >> $O/leaf_prefetch_ab.txt
  python3 $R/tools/dbg/commit_only.py 2>> $O/leaf_prefetch_ab.err | tail -3 >> $O/leaf_prefetch_ab.txt
  python3 $R/bench.py --steps 20 --warmup 5 --rows 1024 $QUIET 2>> $O/leaf_prefetch_ab.err | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('table block:', round(d['value'],1), 'proofs/s on', d['config']['rows_per_rank'], 'rows')" >> $O/leaf_prefetch_ab.txt
done
cat $O/leaf_prefetch_ab.txt
