#!/bin/bash
# round 6, third GPU call: parity of the fused tree levels / new opcodes in the padded circuits, the bench line's new fields, the two A/Bs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_ntt_merkle.py tests/test_gpu_fri.py tests/test_gpu_witness_tape.py tests/test_gpu_table.py -x -q -k "not bench_gpus and not resume_dir and not python_build" > gpurun_out/r06/tests_third.log 2>&1
echo "tests rc=$?" >> gpurun_out/r06/tests_third.log
tail -c 300 gpurun_out/r06/tests_third.log
bash tools/dbg/merkle_fused_ab.sh > /dev/null 2>&1
bash tools/dbg/ntt_priority_ab.sh > /dev/null 2>&1
cat gpurun_out/r06/merkle_fused_ab.txt gpurun_out/r06/ntt_priority_ab.txt
