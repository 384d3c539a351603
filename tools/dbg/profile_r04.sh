# Round-4 evidence run (one gpurun call on the final code): the bench line (default = table workload), rocprofv3 kernel stats of the
# table workload (4 workers and a single worker: un-overlapped kernel durations), of the roofline leg alone and of the batched
# 8192 x 2^12 shape alone, TCC traffic of the 2^22 NTT (separate --pmc passes, no trace domains mixed in), SQ wait counters of the
# NTT passes (shift-twiddle scheme and, with MP2G_NTT_NOSHIFT=1, the classic tables), the recursion workload, host-witness A/B, the
# ungrouped work-plan schedule A/B. Outputs under gpurun_out/r04/.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err   # the driver's command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof4 -- python3 $R/bench.py --steps 2 --warmup 1 --rows 512 --no-leaves-leg --no-verify --config2-leaves 0 --degree-sweep "" --no-cpu-baseline > $O/prof4.json 2> $O/prof4.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 $R/bench.py --steps 1 --warmup 1 --rows 512 --workers 1 --no-leaves-leg --no-verify --config2-leaves 0 --degree-sweep "" --no-cpu-baseline > $O/prof1.json 2> $O/prof1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ntt -- python3 $R/bench.py --workload ntt --steps 20 --warmup 2 > $O/ntt.json 2> $O/ntt.err
MP2G_NTT_NOSHIFT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ntt_classic -- python3 $R/bench.py --workload ntt --steps 20 --warmup 2 > $O/ntt_classic.json 2> $O/ntt_classic.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ntt12 -- python3 $R/tools/dbg/ntt_batched.py > $O/ntt12.txt 2> $O/ntt12.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/traffic_$c -- python3 $R/tools/dbg/traffic_run.py > /dev/null 2> $O/traffic_$c.err
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq_wait -- python3 $R/tools/dbg/traffic_run.py > /dev/null 2> $O/sq_wait.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/sq_valu -- python3 $R/tools/dbg/traffic_run.py > /dev/null 2> $O/sq_valu.err
python3 $R/bench.py --workload recursion --batch 128 --trees 8 --steps 3 --warmup 1 > $O/recursion.json 2> $O/recursion.err
python3 $R/bench.py --steps 2 --warmup 1 --host-witness --no-leaves-leg --no-verify --config2-leaves 0 --degree-sweep "" --no-cpu-baseline > $O/bench_host_witness.json 2> $O/bench_host_witness.err
python3 $R/bench.py --workload leaves > $O/leaves.json 2> $O/leaves.err
python3 $R/bench.py --steps 20 --warmup 5 --python-build --group-rows 1 --no-leaves-leg --no-verify --config2-leaves 0 --degree-sweep "" --no-cpu-baseline > $O/bench_ungrouped.json 2> $O/bench_ungrouped.err   # one work-plan item per worker unit: round 3's schedule
python3 $R/tools/dbg/witness_dev_timing.py > $O/witness_dev_timing.txt 2>&1
# keep only the summaries (the raw traces exceed the merge limit)
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*_agent_info.csv" -delete
find $O -name "*.db" -delete
ls -la $O | head -40
tail -c 600 $O/bench.json
