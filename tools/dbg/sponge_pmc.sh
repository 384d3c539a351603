# SQ counters of the Poseidon2 permutation micro-benchmark (tools/ubench: kperm = 2^22 lanes x 16 permutations)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sp1 /tmp/sp2
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/sp1 -- $R/tools/ubench/ubench > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sp2 -- $R/tools/ubench/ubench > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, statistics
res = {}
for f in glob.glob("/tmp/sp*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "kperm" not in k and "k<0" not in k and "k<7" not in k: continue
        res.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in res.items():
    print(k)
    for c, xs in sorted(v.items()):
        print(f"   {c:28s} {statistics.median(xs):16.0f}  (n={len(xs)})")
PY
