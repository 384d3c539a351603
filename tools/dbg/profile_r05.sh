# Round-5 evidence run, in parts (each part one gpurun call or several chained): outputs under gpurun_out/r05/.
# usage: profile_r05.sh <part> ...   parts: bench | trace | k13 | pmc | sponge | sweep | ntt
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
QUIET="--no-leaves-leg --no-verify --config2-leaves 0 --degree-sweep= --no-cpu-baseline"
# only summaries travel back (gpurun merges at most 64 MiB): raw traces and per-dispatch counter rows are reduced on the box and deleted
keep_small() { find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*.db" -delete; du -sh $O; }
for part in "$@"; do
case $part in
bench)   # the driver's command
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json ;;
trace)   # kernel durations of a table step, four workers and one (un-overlapped), natural degrees
  for wk in 4 1; do
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$wk -- python3 $R/bench.py --steps 2 --warmup 1 --rows 1024 --workers $wk $QUIET > $O/prof$wk.json 2> $O/prof$wk.err
    python3 $R/tools/dbg/trace_by_grid.py $(ls -t $O/prof$wk/*/*_kernel_trace.csv | head -1) 80 > $O/prof${wk}_by_grid.txt
  done ;;
k13)     # the reference-equivalent regime: every base circuit padded to 2^13 rows with the reference's leaf gate set
  for wk in 4 1; do
    timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k13_prof$wk -- python3 $R/bench.py --pad-base-bits 13 --table-batch 16 --steps 1 --warmup 1 --rows 1024 --workers $wk $QUIET > $O/k13_prof$wk.json 2> $O/k13_prof$wk.err
    python3 $R/tools/dbg/trace_by_grid.py $(ls -t $O/k13_prof$wk/*/*_kernel_trace.csv | head -1) 80 > $O/k13_prof${wk}_by_grid.txt
  done ;;
pmc)     # is the chip ALU-saturated in a 4-worker step? chip-wide SQ counters, program directly after --
  # (counter collection serialises the dispatches: the per-kernel sums are exact, the step's wall time under --pmc means nothing;
  #  the saturation figure is total VALU wave-instructions of the step / (wall time of the UNPROFILED step x the chip's issue rate))
  timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/pmc4 -- python3 $R/bench.py --steps 1 --warmup 0 --rows 512 --workers 4 $QUIET > $O/pmc4.json 2> $O/pmc4.err
  python3 $R/tools/dbg/pmc_summary.py $O/pmc4 $O/pmc4_summary.json "bench.py --steps 1 --warmup 0 --rows 512 --workers 4: one block of 512 rows = 2560 framework proofs (+ prover creation)"
  rm -rf $O/pmc4 ;;
sponge)  # the leaf sponge alone: instructions per permutation, cycles per instruction
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sponge_pmc -- python3 $R/tools/dbg/commit_only.py > $O/sponge_pmc.txt 2> $O/sponge_pmc.err
  python3 $R/tools/dbg/pmc_summary.py $O/sponge_pmc $O/sponge_pmc_summary.json "tools/dbg/commit_only.py: 13 commits of 135 x 2^17 values (2^20 leaves x 17 permutations per leaf-kernel launch)"
  rm -rf $O/sponge_pmc
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sponge_trace -- python3 $R/tools/dbg/commit_only.py > $O/sponge_trace.txt 2> $O/sponge_trace.err
  $R/tools/ubench/ubench > $O/ubench.txt 2>&1 ;;
sweep)   # workers x batch with the pipelined forest, and the synchronous one beside it
  for cfg in "4 32 0" "4 32 1" "3 48 0" "2 64 0" "2 64 1" "1 32 0" "1 32 1" "1 64 0"; do
    set -- $cfg
    MP2G_FOREST_SYNC=$3 python3 $R/bench.py --steps 10 --warmup 2 --rows 1024 --workers $1 --table-batch $2 $QUIET 2> $O/sweep.err | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('workers $1 batch $2 sync $3:', round(d['value'],1), 'proofs/s')" >> $O/sweep.txt
  done; cat $O/sweep.txt ;;
ntt)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ntt -- python3 $R/bench.py --workload ntt --steps 50 --warmup 1000 > $O/ntt.json 2> $O/ntt.err   # as bench.py's roofline leg: 1000 untimed transforms (the clocks settle), 50 timed
  for c in FETCH_SIZE WRITE_SIZE; do   # separate passes, no trace domains beside the counters (MI355X_MICROARCH.md, HBM / rocprofv3)
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/traffic_$c -- python3 $R/tools/dbg/traffic_run.py > /dev/null 2> $O/traffic_$c.err
    python3 $R/tools/dbg/pmc_summary.py $O/traffic_$c $O/traffic_${c}_summary.json "tools/dbg/traffic_run.py: 6 calibration calls of scale_powers_kernel (exactly 32768 KB read), 10 forward 2^22 NTTs"
    rm -rf $O/traffic_$c
  done ;;
esac
done
keep_small
ls $O | head -60
