#!/bin/bash
# A/B of two builds of the library on the bench: the product's and build_dbg/$1/libmp2gpu.so (MP2G_LIB), alternating, 2 repetitions
V=${1:-cmad}
for rep in 1 2; do
  for lib in mapreduce-plonky2_amd/libmp2gpu.so build_dbg/$V/libmp2gpu.so; do
    MP2G_LIB=$GRAFT_REPO_ROOT/$lib python3 bench.py --steps 5 --no-cpu-baseline --cpu-budget 4 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), d.get('verified'), round(d['roofline']['launch_ms']*1e3,1), round(d['ntt_batched_2p12']['GBps']), round(d['sponge']['permutations_per_s']/1e9,3), d['stage_ms']['base 2^13 x 64']['quotient'], d['stage_ms']['base 2^13 x 64']['wires_commit'])"
  done
done
