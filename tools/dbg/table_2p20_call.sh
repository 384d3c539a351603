#!/bin/bash
# one call of the 2^20-row table build across calls (bench.py --resume-dir): the block roots kept so far travel in
# profiles/r05/table_2p20_store/ (gpurun_out/ does not travel TO the box), this call's additions come back under gpurun_out/store/
# usage: table_2p20_call.sh <max-seconds> [extra bench.py arguments]
set -u
MAXS=${1:-2600}; shift || true
mkdir -p gpurun_out/store
cp -r profiles/r05/table_2p20_store/. gpurun_out/store/ 2>/dev/null
TAG=$(date +%H%M%S)
MP2G_PROGRESS_FILE=gpurun_out/table_2p20_progress_$TAG.txt python3 bench.py --steps 128 --rows 1024 --warmup 2 --table-blocks 8 \
  --resume-dir gpurun_out/store --max-seconds "$MAXS" --record gpurun_out/table_2p20_rows.json "$@" \
  > gpurun_out/table_2p20_call_$TAG.json 2> gpurun_out/table_2p20_call_$TAG.err
echo "exit $?"; tail -c 600 gpurun_out/table_2p20_call_$TAG.json; tail -5 gpurun_out/table_2p20_call_$TAG.err
