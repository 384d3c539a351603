#!/bin/bash
# round 6, closing GPU call: the driver's command on a fresh box, an 8-rank gloo run with lean blocks (plumbing at a non-trivial size), the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
T0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_final.json 2> gpurun_out/r06/bench_final.err
echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s" >> gpurun_out/r06/bench_final.err
tail -2 gpurun_out/r06/bench_final.err
T0=$(date +%s)
MP2G_BENCH_BACKEND=gloo timeout 1200 python3 bench.py --gpus 8 --rows 1024 --steps 1 --warmup 1 --workers 1 --table-batch 16 --lean --no-cpu-baseline > gpurun_out/r06/bench_8ranks_gloo.json 2> gpurun_out/r06/bench_8ranks_gloo.err
echo "8 ranks rc=$? wall=$(( $(date +%s) - T0 )) s" >> gpurun_out/r06/bench_8ranks_gloo.err
tail -2 gpurun_out/r06/bench_8ranks_gloo.err
T0=$(date +%s)
timeout 1800 python3 -m pytest tests -x -q -m gpu > gpurun_out/r06/gpu_tests_final.log 2>&1
echo "tests rc=$? wall=$(( $(date +%s) - T0 )) s" >> gpurun_out/r06/gpu_tests_final.log
tail -c 300 gpurun_out/r06/gpu_tests_final.log
