# Round-2 evidence run (one gpurun call): bench line, rocprofv3 kernel stats (4 streams and single stream), TCC traffic of
# the 2^22 NTT (separate --pmc passes, no trace domains mixed in), tree workload. Outputs under gpurun_out/r02/.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof4 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify > $O/prof4.json 2> $O/prof4.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-verify > $O/prof1.json 2> $O/prof1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ntt -- python3 $R/bench.py --workload ntt --steps 20 --warmup 2 > $O/ntt.json 2> $O/ntt.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/traffic_$c -- python3 $R/tools/dbg/traffic_run.py > /dev/null 2> $O/traffic_$c.err
done
python3 $R/bench.py --workload tree --steps 2 --warmup 1 > $O/tree.json 2> $O/tree.err
python3 $R/bench.py --workload recursion --batch 128 --trees 8 --steps 3 --warmup 1 > $O/recursion.json 2> $O/recursion.err
python3 $R/bench.py --hasher poseidon --no-cpu-baseline > $O/bench_poseidon.json 2> $O/bench_poseidon.err
# keep only the summaries (the raw traces exceed the merge limit)
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*_agent_info.csv" -delete
ls -la $O $O/prof4/* $O/traffic_FETCH_SIZE/* 2>/dev/null | head -40
tail -c 600 $O/bench.json
