#!/bin/bash
# A/B of two library builds on the roofline leg alone and on the batched 2^12 shape: product vs build_dbg/$1/libmp2gpu.so, alternating
V=${1:-prev}
for rep in 1 2 3; do
  for lib in mapreduce-plonky2_amd/libmp2gpu.so build_dbg/$V/libmp2gpu.so; do
    MP2G_LIB=$GRAFT_REPO_ROOT/$lib python3 bench.py --workload ntt --steps 1000 --warmup 50 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['roofline']['launch_ms']*1e3,2), round(d['roofline']['frac'],4))"
  done
done
