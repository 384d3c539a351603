#!/bin/bash
# Round-6 evidence run (one gpurun call): rocprofv3 summaries the bench line's kernel figures can be recomputed from, on this round's build.
#   ntt     kernel trace of the roofline leg alone + FETCH_SIZE / WRITE_SIZE passes (separate, nothing but --kernel-trace beside them)
#   sponge  kernel trace of the leaf kernel alone (2^20 leaves x 17 permutations) + one SQ counter pass (instructions per permutation)
#   ubench_pmc  SQ_INSTS_VALU / GRBM_GUI_ACTIVE of tools/ubench's single-instruction streams: cycles per instruction of every opcode class
#   table   kernel trace of a four-worker and a one-worker table build at natural degrees (1024-row steps)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
QUIET="--no-verify --config2-leaves 0 --degree-sweep= --no-cpu-baseline"
keep_small() { find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*.db" -delete; du -sh $O; }
for part in "$@"; do
case $part in
ntt)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ntt -- python3 $R/bench.py --workload ntt --steps 50 --warmup 1000 > $O/ntt.json 2> $O/ntt.err
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/traffic_$c -- python3 $R/tools/dbg/traffic_run.py > /dev/null 2> $O/traffic_$c.err
    python3 $R/tools/dbg/pmc_summary.py $O/traffic_$c $O/traffic_${c}_summary.json "tools/dbg/traffic_run.py: 6 calibration calls of scale_powers_kernel (exactly 32768 KB read), 10 forward 2^22 NTTs"
    rm -rf $O/traffic_$c
  done ;;
sponge)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sponge_trace -- python3 $R/tools/dbg/commit_only.py > $O/sponge_trace.txt 2> $O/sponge_trace.err
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sponge_pmc -- python3 $R/tools/dbg/commit_only.py > $O/sponge_pmc.txt 2> $O/sponge_pmc.err
  python3 $R/tools/dbg/pmc_summary.py $O/sponge_pmc $O/sponge_pmc_summary.json "tools/dbg/commit_only.py: 13 commits of 135 x 2^17 values (2^20 leaves x 17 permutations per leaf-kernel launch)"
  rm -rf $O/sponge_pmc ;;
ubench_pmc)  # the single-instruction streams priced in SHADER CYCLES per instruction (clock-free: GRBM_GUI_ACTIVE / SQ_INSTS_VALU), not in time
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/ubench_pmc -- $R/tools/ubench/ubench > $O/ubench_under_pmc.txt 2> $O/ubench_pmc.err
  python3 $R/tools/dbg/pmc_summary.py $O/ubench_pmc $O/ubench_pmc_summary.json "tools/ubench/ubench: every row's kernel launched twice (timeit: one untimed, one timed)"
  rm -rf $O/ubench_pmc ;;
table)
  for wk in 4 1; do
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$wk -- python3 $R/bench.py --steps 2 --warmup 1 --rows 1024 --workers $wk $QUIET > $O/prof$wk.json 2> $O/prof$wk.err
  done ;;
esac
done
for tag in prof_ntt sponge_trace prof4 prof1; do
  S=$(ls -t $O/$tag/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$S" ] && cp $S $O/${tag}_kernel_stats.csv
done
keep_small
ls $O | head -60
