#!/bin/bash
# the shift-twiddle scheme of ntt.hip (w_64 = 2^3: wave-uniform shifts after the first radix-8 round, one merged table in the second)
# against the classic per-round tables (MP2G_NTT_NOSHIFT=1): correctness first (the NTT tests under both), then the 2^22 transform and
# the batched shapes between HIP events, alternating, and the kernel durations under rocprofv3
R=$GRAFT_REPO_ROOT
cd $R
echo "== tests (shift scheme)"; timeout 900 python -m pytest tests/test_gpu_ntt_merkle.py -x -q -m gpu 2>&1 | tail -3
echo "== tests (classic tables)"; MP2G_NTT_NOSHIFT=1 timeout 900 python -m pytest tests/test_gpu_ntt_merkle.py -x -q -m gpu -k "ntt" 2>&1 | tail -2
for rep in 1 2 3; do
  echo "== rep $rep: 2^22 (shift | classic)"
  python tools/dbg/ntt22.py "({}, {'MP2G_NTT_NOSHIFT': '1'})"
done
echo "== batched shapes: shift"; python tools/dbg/ntt_batched.py
echo "== batched shapes: classic"; MP2G_NTT_NOSHIFT=1 python tools/dbg/ntt_batched.py
cd /tmp && export TMPDIR=/tmp
for mode in shift classic; do
  rm -rf /tmp/tr
  if [ $mode = classic ]; then export MP2G_NTT_NOSHIFT=1; else unset MP2G_NTT_NOSHIFT; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/bench.py --workload ntt --steps 200 --warmup 50 > /dev/null 2>&1
  echo "== rocprof ($mode)"; grep -E "ntt_(rows|cols)" /tmp/tr/*/*_kernel_stats.csv | sed "s/(mp2g::NttArgs[^\"]*\"//" | cut -d, -f1-4
  mkdir -p $R/gpurun_out/ntt_shift; cp /tmp/tr/*/*_kernel_stats.csv $R/gpurun_out/ntt_shift/kernel_stats_$mode.csv 2>/dev/null
done
