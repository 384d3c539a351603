R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ntt -- python3 $R/bench.py --workload ntt --steps 20 --warmup 2 > $O/ntt.json 2> $O/ntt.err
cat $O/ntt.json | cut -c1-900; grep ntt_ $O/prof_ntt/*/*_kernel_stats.csv | cut -d, -f1-5
python3 $R/bench.py --workload ntt --steps 20 --warmup 2 2>/dev/null | cut -c400-800
