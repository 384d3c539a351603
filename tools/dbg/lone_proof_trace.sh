# where a lone proof's 5 ms go: kernel trace of B = 1 proves on one stream (busy time vs gaps, top kernels)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/lp -- python3 $R/bench.py --batch 1 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/lp/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the timed region: the last 4 steps = find the pow_kernel launches (one per prove): 2 proves per step
pw = [i for i, r in enumerate(rows) if "pow_kernel" in r["Kernel_Name"]]
# take the span between the 5th-last and the last pow kernel: 4 proves... use the last 8 proves
i0, i1 = pw[-9], pw[-1]
seg = rows[i0 + 1:i1 + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"8 proves (4 base 2^13 + 4 wrap 2^12): wall {(t1 - t0) / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms ({100 * busy / (t1 - t0):.0f} %), {len(seg)} launches = {len(seg) / 8:.0f} per prove, mean gap {(t1 - t0 - busy) / len(seg) / 1e3:.1f} us")
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"].split("(")[0].replace("void mp2g::", "").replace("mp2g::", "")
    agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
for k, (t, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:22]:
    print(f"  {t / 8e3:8.1f} us/prove  {n / 8:6.1f} launches/prove  avg {t / n / 1e3:7.1f} us  {k[:70]}")
PY
