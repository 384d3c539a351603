#!/bin/bash
# Round 6: two ranks sharing one GPU (8 streams, 8 hardware queues) built 903 proofs/s where one rank's 4 x 48 builds 882-890: is it the
# streams or the queues? workers x GPU_MAX_HW_QUEUES on the 20 480-row block, one process.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
QUIET="--no-verify --config2-leaves 0 --degree-sweep= --no-cpu-baseline"
: > $O/workers_sweep.txt
for cfg in "4 4" "4 8" "6 8" "8 8" "8 4" "6 8 32" "8 8 32" "4 4"; do
  set -- $cfg
  B=${3:-48}
  GPU_MAX_HW_QUEUES=$2 python3 $R/bench.py --steps 20 --warmup 5 --rows 1024 --workers $1 --table-batch $B $QUIET 2>> $O/workers_sweep.err | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('workers $1 queues $2 batch $B:', round(d['value'],1), 'proofs/s,', round(d['config']['device_memory_used_bytes']/1e9,1), 'GB')" >> $O/workers_sweep.txt
done
cat $O/workers_sweep.txt
