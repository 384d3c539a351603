#!/bin/bash
# rocprofv3 kernel trace + stats of the default (table) workload; summary -> gpurun_out/prof_table/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_table
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $R/bench.py --steps 2 --warmup 1 --no-leaves-leg --no-verify "$@" > $OUT/bench.json 2> $OUT/bench.err
find $OUT/raw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/raw -name "*kernel_trace.csv" -size +30M -delete
find $OUT/raw -name "*_agent_info.csv" -delete
head -45 $OUT/kernel_stats.csv | cut -c1-200
tail -c 300 $OUT/bench.json
