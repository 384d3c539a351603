# SQ counters of the two 2^22 NTT kernels (separate --pmc passes of <= 8 SQ counters, only --kernel-trace beside them).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ntt_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/dbg/traffic_run.py > /dev/null 2> $O/p$i.err
  tail -2 $O/p$i.err
done
python3 - <<'PY'
import csv, glob, os, statistics
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/ntt_pmc"
res = {}
for f in glob.glob(O + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "ntt_" not in k or "nat" in k: continue
        res.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in res.items():
    print(k)
    for c, xs in sorted(v.items()):
        print(f"   {c:28s} {statistics.median(xs):16.0f}  (n={len(xs)})")
PY
