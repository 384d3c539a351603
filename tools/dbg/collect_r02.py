"""Turn the raw outputs of tools/dbg/profile_r02.sh (gpurun_out/r02/) into the committed summaries under profiles/r02/:
kernel-stats CSVs, the bench JSON lines, the NTT traffic from the TCC counters (FETCH_SIZE calibrated on scale_powers_kernel in
the same run), the 2^22 NTT's two launches matched in the kernel trace, the kernel shares of a single-stream step."""
import csv, glob, json, os, shutil, statistics, sys


def newest(pattern):
    """gpurun merges into gpurun_out/ without deleting: an earlier run's files (other pid prefix) may sit beside the new ones"""
    return max(glob.glob(pattern), key=os.path.getmtime)


ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(ROOT, "gpurun_out", "r02"), os.path.join(ROOT, "profiles", "r02")
os.makedirs(dst, exist_ok=True)
shutil.copy(newest(f"{src}/prof4/runc/*_kernel_stats.csv"), f"{dst}/bench_r02_kernel_stats.csv")
shutil.copy(newest(f"{src}/prof1/runc/*_kernel_stats.csv"), f"{dst}/bench_r02_single_stream_kernel_stats.csv")
if glob.glob(f"{src}/prof_ntt/runc/*_kernel_stats.csv"):
    shutil.copy(newest(f"{src}/prof_ntt/runc/*_kernel_stats.csv"), f"{dst}/bench_r02_ntt_kernel_stats.csv")
summary = {}
for a, b in (("bench.json", "bench_r02.json"), ("tree.json", "bench_r02_tree.json"), ("recursion.json", "bench_r02_recursion.json"),
             ("bench_poseidon.json", "bench_r02_poseidon.json"), ("prof4.json", "bench_r02_under_rocprof.json"), ("ntt.json", "bench_r02_ntt_under_rocprof.json")):
    if not os.path.exists(f"{src}/{a}"):
        continue
    line = [l for l in open(f"{src}/{a}").read().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    open(f"{dst}/{b}", "w").write(json.dumps(d, indent=1) + "\n")
    summary[b] = {k: d.get(k) for k in ("value", "ms_per_step", "verified", "framework_proofs_per_s")}
    if "roofline" in d:
        summary[b]["ntt_us"] = d["roofline"]["launch_ms"] * 1e3
        summary[b]["frac"] = d["roofline"]["frac"]
    if "cpu_baseline" in d:
        summary[b]["cpu"] = d["cpu_baseline"]["value"]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(f"{src}/traffic_{c}/runc/*_counter_collection.csv")
    by = {}
    for r in csv.DictReader(open(f)):
        by.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    for k, v in by.items():
        res.setdefault(k, {})[c] = statistics.median(v)
cal = 32768.0 / res["mp2g::scale_powers_kernel"]["FETCH_SIZE"]
cols_name = next(k for k in res if "ntt_cols" in k and "kernel<10" in k)
rows_name = next(k for k in res if "ntt_rows" in k and "nat" not in k and "kernel<12, 0>" in k)
cols, rows = res[cols_name], res[rows_name]
fetch = (cols["FETCH_SIZE"] + rows["FETCH_SIZE"]) * cal * 1024
write = (cols["WRITE_SIZE"] + rows["WRITE_SIZE"]) * 1024
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, WRITE_SIZE) --kernel-trace --output-format csv -- python3 tools/dbg/traffic_run.py   (tools/dbg/profile_r02.sh)",
           "units": "KB per dispatch as reported (median over the dispatches of each kernel); FETCH_SIZE scaled by the factor calibrated in this same run on scale_powers_kernel, which reads exactly 32768 KB with 8 B/lane loads (MI355X_MICROARCH.md: FETCH_SIZE under-reports 8 B/lane patterns by 2x)",
           "fetch_calibration_factor": cal, "kernels": {k: v for k, v in res.items() if "ntt" in k or "scale_powers" in k or "tw4" in k},
           "ntt_2p22_forward_bitrev": {"fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes": fetch + write, "algorithmic_bytes": 16 << 22,
                                       "note": f"two launches ({cols_name.split('::')[-1]}, {rows_name.split('::')[-1]}); pass A also streams the 32 MiB 4-step twiddle table (one multiply per point instead of two)"},
           "note_natural_order_pass": "ntt_rows_nat_kernel<12,0> (natural-order output of a 2^22 transform, used by the calibration call only, never by prove()) writes ~4x its 32 MiB: with one row per tile its stores are 8-byte scatters. The prover's natural-order transforms are 2^15 / 2^16 points (16 rows per tile: 128-byte stores)."},
          open(f"{dst}/ntt_traffic.json", "w"), indent=1)
summary["ntt_traffic_bytes"] = fetch + write
for tag in ("prof4", "prof1"):
    f = newest(f"{src}/{tag}/runc/*_kernel_trace.csv")
    rows_ = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    pairs = []
    for i, r in enumerate(rows_):
        if "ntt_cols" in r["Kernel_Name"] and "kernel<10" in r["Kernel_Name"]:
            for s in rows_[i + 1:i + 6]:
                if "ntt_rows" in s["Kernel_Name"] and "nat" not in s["Kernel_Name"] and "kernel<12, 0>" in s["Kernel_Name"] and s["Stream_Id"] == r["Stream_Id"]:
                    pairs.append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(s["End_Timestamp"]) - int(s["Start_Timestamp"])) / 1e3))
                    break
    summary[f"ntt_pair_{tag}"] = {"n": len(pairs), "cols_us": statistics.mean(p[0] for p in pairs), "rows_us": statistics.mean(p[1] for p in pairs)}
rows_ = list(csv.DictReader(open(f"{dst}/bench_r02_single_stream_kernel_stats.csv")))
# the shares are those of the prover's step: the roofline leg's own launches (1050 2^22 transforms = the matched pairs above) and the
# sponge / digest legs of the bench are taken out
leg = summary["ntt_pair_prof1"]
leg_ns = {"ntt_cols_v2_kernel<10, 3>": leg["n"] * leg["cols_us"] * 1e3, "ntt_rows_v2_kernel<12, 0>": leg["n"] * leg["rows_us"] * 1e3}
for r in rows_:
    for k, v in leg_ns.items():
        if k in r["Name"]:
            r["TotalDurationNs"] = max(0.0, float(r["TotalDurationNs"]) - v)
rows_ = [r for r in rows_ if "hash_no_pad_batch" not in r["Name"] and "row_digest" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows_)
agg = {}
for r in rows_:
    n = r["Name"]
    key = ("gate constraints" if "gate_constraints_lde" in n else "leaf sponge" if "leaf_hash" in n else "merkle levels" if "merkle_level" in n else
           "NTT/LDE" if "ntt_" in n or "scale_powers" in n else "PoW" if "pow_kernel" in n else "quotient perm" if "quotient_perm" in n else
           "Z/partial products" if "zpp" in n else "transcript" if "ch_" in n else "sponge bench/digest" if "hash_no_pad_batch" in n or "row_digest" in n else "other")
    agg[key] = agg.get(key, 0) + float(r["TotalDurationNs"])
summary["single_stream_shares_pct"] = {k: round(100 * v / tot, 1) for k, v in sorted(agg.items(), key=lambda x: -x[1])}
json.dump(summary, open(f"{dst}/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
