#!/bin/bash
# Round 6: v_cndmask_b32 selecting by an SGPR pair costs 4.2 cycles as a stream, by VCC in the VOP2 encoding (to be measured here);
# the weak reductions with the multiply-add's carry in VCC (-DGL_REDUCE_VCC) against the product's, in tools/ubench (time and cycles)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench/ubench > $O/ubench2.txt 2>&1
$R/tools/ubench/ubench_vcc > $O/ubench_vcc.txt 2>&1
for v in ubench ubench_vcc; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/${v}_pmc2 -- $R/tools/ubench/$v > /dev/null 2> $O/${v}_pmc2.err
  python3 $R/tools/dbg/pmc_summary.py $O/${v}_pmc2 $O/${v}_pmc2_summary.json "tools/ubench/$v"
  rm -rf $O/${v}_pmc2
done
grep -E "inst |mulw\(32chain\)|red\(32chain\)|gl_mul |poseidon|mismatch" $O/ubench2.txt
echo == vcc; grep -E "mulw\(32chain\)|red\(32chain\)|gl_mul |gl_mul_small|poseidon|mismatch" $O/ubench_vcc.txt
