#!/bin/bash
# the two-sponges-per-lane experiment on the leaf sponge (merkle.hip, MP2G_LEAF_ILP2 = 0 one sponge per lane / 1 two sponges, 2 waves
# per SIMD / 2 two sponges held to 3 waves per SIMD): kernel durations under rocprofv3 for 2^20 leaves of 135 limbs, then the table bench
R=$GRAFT_REPO_ROOT
# the experiment lives in a variant library (the product has no switch)
export MP2G_LIB=$($R/tools/dbg/build_variant.sh ilp2 "-DMP2G_EXPERIMENT_LEAF_ILP2" merkle.hip | tail -1)
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2; do
  export MP2G_LEAF_ILP2=$m
  rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/tools/dbg/sponge_ab.py > /tmp/sp.txt 2>&1
  echo "== MP2G_LEAF_ILP2=$m"; grep commit /tmp/sp.txt
  grep -E "leaf_hash_poly" /tmp/tr/*/*_kernel_stats.csv | cut -d, -f1-4 | sed 's/(unsigned long const[^"]*"//'
done
cd $R
for rep in 1 2; do for m in 0 1 2; do
  MP2G_LEAF_ILP2=$m python bench.py --steps 2 --warmup 1 --config2-leaves 0 --degree-sweep "" --no-leaves-leg --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MP2G_LEAF_ILP2=$m table:', round(d['value'],1), 'proofs/s')"
done; done
