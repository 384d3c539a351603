#!/bin/bash
# round 6, fourth GPU call: the driver's command on a FRESH box first (cold imports included, as the driver will run it), then the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
T0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_rehearsal2.json 2> gpurun_out/r06/bench_rehearsal2.err
echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s" >> gpurun_out/r06/bench_rehearsal2.err
tail -3 gpurun_out/r06/bench_rehearsal2.err
T0=$(date +%s)
timeout 1800 python3 -m pytest tests -x -q -m gpu > gpurun_out/r06/gpu_tests.log 2>&1
echo "tests rc=$? wall=$(( $(date +%s) - T0 )) s" >> gpurun_out/r06/gpu_tests.log
tail -c 400 gpurun_out/r06/gpu_tests.log
