# kernel shares of the real-recursion workload (witness check on): rocprofv3 kernel stats of two 32-leaf trees in flight
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02rec
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --workload recursion --batch 128 --trees 8 --steps 1 --warmup 1 > $O/recursion_prof.json 2> $O/recursion_prof.err
find $O -name "*kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
tail -c 300 $O/recursion_prof.json
