#!/bin/bash
# round 6, first GPU call: the bench tests that changed, the instruction-cost rows, the driver's command as a rehearsal
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_table.py -x -q -k "bench_gpus_4 or bench_gpus_8 or bench_default or native_build_equals or pipelined" > gpurun_out/r06/tests_first.log 2>&1
echo "tests rc=$?" >> gpurun_out/r06/tests_first.log
tools/ubench/ubench > gpurun_out/r06/ubench.txt 2>&1
tools/ubench/ubench_madsum > gpurun_out/r06/ubench_madsum.txt 2>&1
/usr/bin/time -v python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_rehearsal.json 2> gpurun_out/r06/bench_rehearsal.err
echo "bench rc=$?" >> gpurun_out/r06/bench_rehearsal.err
tail -c 600 gpurun_out/r06/tests_first.log; tail -5 gpurun_out/r06/bench_rehearsal.err
