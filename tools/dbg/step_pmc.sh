# VALU issue efficiency of every kernel of a prover step (or of the python script given as arguments): cycles per VALU
# instruction and where the waves wait
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st1
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/st1 -- python3 ${@:-$R/bench.py --workload leaves --steps 1 --warmup 0 --streams 1 --batch 64 --no-cpu-baseline --no-verify} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, statistics
res = {}
for f in glob.glob("/tmp/st1/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void mp2g::", "").replace("mp2g::", "")
        res.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
rows = []
for k, v in res.items():
    tot = {c: sum(xs) for c, xs in v.items()}
    gui = tot.get("GRBM_GUI_ACTIVE", 0) / 8
    iv = tot.get("SQ_INSTS_VALU", 0)
    wc = tot.get("SQ_WAVE_CYCLES", 1)
    rows.append((gui, k, len(v["SQ_WAVES"]), iv, gui * 1024 / iv if iv else 0, tot.get("SQ_WAIT_ANY", 0) / wc, tot.get("SQ_WAIT_INST_ANY", 0) / wc, tot.get("SQ_ACTIVE_INST_ANY", 0) / wc))
tg = sum(r[0] for r in rows)
print(f"{'kernel':58s} calls  share  cyc/VALU  parked  issue-stall  active")
for gui, k, n, iv, cpi, wa, wi, ac in sorted(rows, reverse=True)[:34]:
    print(f"{k[:58]:58s} {n:5d} {100*gui/tg:5.1f}%  {cpi:7.2f}  {100*wa:5.1f}%  {100*wi:5.1f}%  {100*ac:5.1f}%")
PY
