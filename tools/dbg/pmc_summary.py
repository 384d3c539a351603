"""Reduce a rocprofv3 --pmc run (counter_collection.csv: one row per dispatch and counter) to a small JSON: per kernel the number of
dispatches and the SUM of every counter over its dispatches, plus the grand totals -- what is kept from a PMC pass of a whole
proving step (the raw CSV is hundreds of MB).   usage: pmc_summary.py <dir with */*_counter_collection.csv> <out.json> [note]"""
import csv, glob, json, sys
by, tot, n_rows = {}, {}, 0
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void mp2g::", "").replace("mp2g::", "")
            c, v = r["Counter_Name"], float(r["Counter_Value"])
            e = by.setdefault(k, {"dispatches": set(), "counters": {}})
            e["dispatches"].add(r["Dispatch_Id"])
            e["counters"][c] = e["counters"].get(c, 0.0) + v
            tot[c] = tot.get(c, 0.0) + v
            n_rows += 1
out = {"note": sys.argv[3] if len(sys.argv) > 3 else "", "rows_read": n_rows, "totals": tot,
       "kernels": {k: {"dispatches": len(e["dispatches"]), **e["counters"]} for k, e in sorted(by.items(), key=lambda kv: -kv[1]["counters"].get("SQ_INSTS_VALU", 0))}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(f"{sys.argv[2]}: {len(by)} kernels, totals {tot}")
