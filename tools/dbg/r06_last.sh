#!/bin/bash
# round 6, last GPU call: the whole GPU suite and the driver's command on the round's final code
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
T0=$(date +%s)
timeout 1800 python3 -m pytest tests -x -q -m gpu > gpurun_out/r06/gpu_tests_last.log 2>&1
echo "tests rc=$? wall=$(( $(date +%s) - T0 )) s" >> gpurun_out/r06/gpu_tests_last.log
tail -c 300 gpurun_out/r06/gpu_tests_last.log
T0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_last.json 2> gpurun_out/r06/bench_last.err
echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s" >> gpurun_out/r06/bench_last.err
tail -2 gpurun_out/r06/bench_last.err
