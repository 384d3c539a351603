#!/bin/bash
# the multi-rank code paths of bench.py on a one-GPU box (ranks share the device, collectives over gloo): plumbing, not a measurement
export MP2G_BENCH_BACKEND=gloo
for wl in leaves tree recursion; do
  extra=""
  [ $wl = recursion ] && extra="--batch 32 --trees 2"
  [ $wl = leaves ] && extra="--cpu-budget 4"
  echo "== $wl"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --workload $wl $extra 2>&1 | tail -2 | cut -c1-900
done
