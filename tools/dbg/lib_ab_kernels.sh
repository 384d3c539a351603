#!/bin/bash
# per-kernel A/B of two library builds: rocprofv3 kernel stats of a single-stream bench run with each (MP2G_LIB); kernels matching
# the regex $2 are listed (default: the gate and permutation-quotient kernels)
V=${1:-carryacc}
export PAT=${2:-gate_constraints|quotient_perm}
cd /tmp && export TMPDIR=/tmp
for lib in mapreduce-plonky2_amd/libmp2gpu.so build_dbg/$V/libmp2gpu.so; do
  export MP2G_LIB=$GRAFT_REPO_ROOT/$lib
  rm -rf /tmp/ab_prof
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-verify > /dev/null 2>&1
  echo "== $lib"
  python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/ab_prof/*/*_kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows:
    n=r['Name']
    import re, os
    if re.search(os.environ['PAT'], n):
        k=n.split('<')[1].split('>')[0] if '<' in n else 'light/perm'
        print(f"  {n.split('(')[0][-45:]:45s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
        tot+=int(r['TotalDurationNs'])
print("  total gate+perm ms", tot/1e6)
PY
done
