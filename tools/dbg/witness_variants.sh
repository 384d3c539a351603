#!/bin/bash
# A/B of the witness kernel's footprint (lanes per proof block, VGPR cap) on the table workload: variant libraries built beside
# the product (build_dbg/wit_<lanes>_<bounds>/libmp2gpu.so, selected with MP2G_LIB; the product library is not touched)
R=$GRAFT_REPO_ROOT
cd $R
for v in "512 512" "512 1024" "256 256" "256 1024" "1024 1024"; do
  set -- $v
  bash tools/dbg/build_variant.sh wit_$1_$2 "-DWIT_LANES_N=$1 -DWIT_BOUNDS=$2" witness_dev.hip > /dev/null
  export MP2G_LIB=$R/build_dbg/wit_$1_$2/libmp2gpu.so
  python tools/dbg/witness_dev_timing.py 2>/dev/null | grep "B=32"
  python bench.py --steps 2 --warmup 1 --no-leaves-leg --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $1 bounds $2:', round(d['value'],1), 'proofs/s')"
done
