#!/bin/bash
# round 6, second GPU call: the witness-tape tests, then the driver's command as a rehearsal (timed with bash's own clock)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_witness_tape.py tests/test_gpu_c_abi.py -x -q > gpurun_out/r06/tests_second.log 2>&1
echo "tests rc=$?" >> gpurun_out/r06/tests_second.log
T0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_rehearsal.json 2> gpurun_out/r06/bench_rehearsal.err
echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s" >> gpurun_out/r06/bench_rehearsal.err
free -g >> gpurun_out/r06/bench_rehearsal.err; nproc >> gpurun_out/r06/bench_rehearsal.err
tail -c 400 gpurun_out/r06/tests_second.log; tail -8 gpurun_out/r06/bench_rehearsal.err
