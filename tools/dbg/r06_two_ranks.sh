#!/bin/bash
# round 6: the driver's command shape at N = 2 with the REAL block size (2^17 rows per rank), the two ranks sharing the one GPU there is
# (gloo: a plumbing rehearsal of the SCALE run -- lean blocks, the pre-timing probe, the join over two real block roots -- not a measurement)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
T0=$(date +%s)
MP2G_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r06/bench_2ranks_2p17_gloo.json 2> gpurun_out/r06/bench_2ranks_2p17_gloo.err
echo "rc=$? wall=$(( $(date +%s) - T0 )) s" >> gpurun_out/r06/bench_2ranks_2p17_gloo.err
tail -3 gpurun_out/r06/bench_2ranks_2p17_gloo.err; tail -c 600 gpurun_out/r06/bench_2ranks_2p17_gloo.json
