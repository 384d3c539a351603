#!/bin/bash
# Round 6: the multiply-reduce with the product's last carry word entering through the reduction's carry-in (gl_reduce128w_split, the
# product's form) against the split form of rounds 1-5 (-DGL_MUL_SPLIT_TAIL): tools/ubench, time and cycles
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ubench ubench_oldtail ubench ubench_oldtail; do
  echo "== $v"; $R/tools/ubench/$v 2>&1 | grep -E "mismatch|gl_mul |mulw\(32chain\)|poseidon"
done | tee $O/ubench_tail.txt
for v in ubench ubench_oldtail; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/${v}_pmc3 -- $R/tools/ubench/$v > /dev/null 2> $O/${v}_pmc3.err
  python3 $R/tools/dbg/pmc_summary.py $O/${v}_pmc3 $O/${v}_pmc3_summary.json "tools/ubench/$v" > /dev/null
  rm -rf $O/${v}_pmc3
  python3 -c "
import json; k=json.load(open('$O/${v}_pmc3_summary.json'))['kernels']
for n in ('kperm','void k<12>','void k<0>'):
    print('$v', n, 'VALU wave-insts', int(k[n]['SQ_INSTS_VALU']), 'cycles/inst', round(k[n]['GRBM_GUI_ACTIVE']/8*1024/k[n]['SQ_INSTS_VALU'],3), 'GUI cycles', int(k[n]['GRBM_GUI_ACTIVE']/8))" | tee -a $O/ubench_tail.txt
done
