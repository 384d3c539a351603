#!/bin/bash
# Round 6, VERDICT r05 item 6: a worker's memory-phase kernels starve behind the other workers' sponge workgroups (a 0.5 ms transform
# averaged 8.5 ms, max 111 ms, in the four-worker trace of round 5). A/B: the product library against the variant whose transforms run
# on a HIGH priority stream per context (csrc/ntt.h MP2G_EXPERIMENT_NTT_PRIORITY, forked from / joined to the worker's stream by
# events), alternating, on the 20480-row block; then a four-worker kernel trace of each for the transforms' average / max durations.
# Exit criterion: >= +2 % on the block or max transform duration < 10 ms; else dropped.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
QUIET="--no-leaves-leg --no-verify --config2-leaves 0 --degree-sweep= --no-cpu-baseline"
V=$R/build_dbg/nttprio/libmp2gpu.so
: > $O/ntt_priority_ab.txt
for rep in 1 2; do
  for mode in product priority; do
    if [ $mode = priority ]; then export MP2G_LIB=$V MP2G_NTT_PRIORITY=1; else unset MP2G_LIB MP2G_NTT_PRIORITY; fi
    python3 $R/bench.py --steps 20 --warmup 5 --rows 1024 $QUIET 2>> $O/ntt_priority_ab.err | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$mode rep $rep:', round(d['value'],1), 'proofs/s on', d['config']['rows_per_rank'], 'rows')" >> $O/ntt_priority_ab.txt
  done
done
for mode in product priority; do
  if [ $mode = priority ]; then export MP2G_LIB=$V MP2G_NTT_PRIORITY=1; else unset MP2G_LIB MP2G_NTT_PRIORITY; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prio_$mode -- python3 $R/bench.py --steps 2 --warmup 1 --rows 1024 --workers 4 $QUIET > $O/prio_$mode.json 2> $O/prio_$mode.err
  S=$(ls -t $O/prio_$mode/*/*_kernel_stats.csv | head -1)
  cp $S $O/ntt_priority_${mode}_kernel_stats.csv
  echo "== $mode: transforms in the four-worker trace (name, calls, total ns, avg ns, %, min, max)" >> $O/ntt_priority_ab.txt
  grep -i "ntt_\|transpose" $S | head -12 >> $O/ntt_priority_ab.txt
  rm -rf $O/prio_$mode
done
cat $O/ntt_priority_ab.txt
