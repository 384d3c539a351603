"""Turn the raw outputs of tools/dbg/profile_r04.sh (gpurun_out/r04/) into the committed summaries under profiles/r04/: the bench
JSON lines, kernel-stats CSVs (table workload with 4 workers and with one; the roofline leg alone; the batched prover shapes alone),
the NTT traffic from the TCC counters (FETCH_SIZE calibrated on scale_powers_kernel in the same run), the SQ wait / VALU counters of
the two NTT passes, the 2^22 NTT's two launches matched in the kernel trace, kernel shares of a single-worker table step."""
import csv, glob, json, os, shutil, statistics

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(ROOT, "gpurun_out", "r04"), os.path.join(ROOT, "profiles", "r04")
os.makedirs(dst, exist_ok=True)
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)
for tag, name in (("prof4", "table_4workers"), ("prof1", "table_1worker"), ("prof_ntt", "ntt_2p22"), ("prof_ntt_classic", "ntt_2p22_classic_tables"), ("prof_ntt12", "ntt_batched")):
    shutil.copy(newest(f"{src}/{tag}/runc/*_kernel_stats.csv"), f"{dst}/{name}_kernel_stats.csv")
summary = {}
for a, b in (("bench.json", "bench_r04.json"), ("prof4.json", "bench_r04_under_rocprof_4workers.json"), ("prof1.json", "bench_r04_under_rocprof_1worker.json"),
             ("recursion.json", "bench_r04_recursion.json"), ("bench_host_witness.json", "bench_r04_host_witness.json"), ("leaves.json", "bench_r04_leaves.json"),
             ("ntt.json", "bench_r04_ntt_under_rocprof.json"), ("ntt_classic.json", "bench_r04_ntt_classic_tables_under_rocprof.json"),
             ("bench_ungrouped.json", "bench_r04_ungrouped_items.json")):
    if not os.path.exists(f"{src}/{a}"):
        continue
    line = [l for l in open(f"{src}/{a}").read().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    open(f"{dst}/{b}", "w").write(json.dumps(d, indent=1) + "\n")
    summary[b] = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "verified", "framework_proofs_per_s", "rows_per_s", "table_2p20_rows_extrapolated_s") if d.get(k) is not None}
    if "config2" in d:
        summary[b]["config2_framework_proofs_per_s"] = d["config2"].get("value")
    if "by_base_degree" in d:
        summary[b]["by_base_degree_proofs_per_s"] = {k: v.get("value") for k, v in d["by_base_degree"].items()}
    if "roofline" in d:
        summary[b]["ntt_us"], summary[b]["frac"] = d["roofline"]["launch_ms"] * 1e3, d["roofline"]["frac"]
    if "cpu_baseline" in d:
        summary[b]["cpu"] = d["cpu_baseline"]["value"]
for f in ("ntt12.txt", "witness_dev_timing.txt"):
    shutil.copy(f"{src}/{f}", f"{dst}/{f}")
# ---- counters of the 2^22 NTT (tools/dbg/traffic_run.py: 6 calibration calls, 10 forward transforms)
res = {}
for tag, counters in (("traffic_FETCH_SIZE", ["FETCH_SIZE"]), ("traffic_WRITE_SIZE", ["WRITE_SIZE"]), ("sq_wait", ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES"]),
                      ("sq_valu", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"])):
    f = newest(f"{src}/{tag}/runc/*_counter_collection.csv")
    by = {}
    for r in csv.DictReader(open(f)):
        by.setdefault((r["Kernel_Name"].split("(")[0], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in by.items():
        res.setdefault(k, {})[c] = statistics.median(v)
cal = 32768.0 / res["mp2g::scale_powers_kernel"]["FETCH_SIZE"]
cols_name = next(k for k in res if "ntt_cols" in k and "kernel<10" in k)
rows_name = next(k for k in res if "ntt_rows" in k and "nat" not in k and "kernel<12, 0" in k)
cols, rows = res[cols_name], res[rows_name]
fetch = (cols["FETCH_SIZE"] + rows["FETCH_SIZE"]) * cal * 1024
write = (cols["WRITE_SIZE"] + rows["WRITE_SIZE"]) * 1024
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (and, in separate passes, WRITE_SIZE; SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES; SQ_INSTS_VALU SQ_ACTIVE_INST_VALU) --kernel-trace "
                      "--output-format csv -- python3 tools/dbg/traffic_run.py   (tools/dbg/profile_r04.sh)",
           "units": "KB per dispatch as reported (median over the dispatches of each kernel); FETCH_SIZE scaled by the factor calibrated in this same run on scale_powers_kernel, "
                    "which reads exactly 32768 KB with 8 B/lane loads (MI355X_MICROARCH.md: FETCH_SIZE under-reports such patterns by 2x)",
           "fetch_calibration_factor": cal, "kernels": {k: v for k, v in res.items() if "ntt" in k or "scale_powers" in k},
           "ntt_2p22_forward_bitrev": {"fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes": fetch + write, "algorithmic_bytes": 16 << 22,
                                       "note": f"two launches ({cols_name.split('::')[-1]}, {rows_name.split('::')[-1]}); pass A also streams the 32 MiB 4-step twiddle table (one multiply per point instead of two: "
                                               "dropping it costs 3 us, DESIGN.md section 4)"},
           "sq": {n.split("::")[-1]: {"wait_any_over_wave_cycles": k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], "valu_insts_per_point": k["SQ_INSTS_VALU"] * 64 / (1 << 22) if "SQ_INSTS_VALU" in k else None}
                  for n, k in ((cols_name, cols), (rows_name, rows))}},
          open(f"{dst}/ntt_traffic.json", "w"), indent=1)
summary["ntt_traffic_bytes"] = fetch + write
summary["ntt_sq_wait_any_frac"] = {"cols": cols["SQ_WAIT_ANY"] / cols["SQ_WAVE_CYCLES"], "rows": rows["SQ_WAIT_ANY"] / rows["SQ_WAVE_CYCLES"]}
# ---- the 2^22 pair in the roofline-leg trace
f = newest(f"{src}/prof_ntt/runc/*_kernel_trace.csv")
rows_ = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
pairs = []
for i, r in enumerate(rows_):
    if "ntt_cols" in r["Kernel_Name"] and "kernel<10" in r["Kernel_Name"]:
        for s in rows_[i + 1:i + 4]:
            if "ntt_rows" in s["Kernel_Name"] and "kernel<12, 0" in s["Kernel_Name"]:
                pairs.append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(s["End_Timestamp"]) - int(s["Start_Timestamp"])) / 1e3,
                              (int(s["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
                break
summary["ntt_pair_roofline_leg"] = {"n": len(pairs), "cols_us": statistics.mean(p[0] for p in pairs), "rows_us": statistics.mean(p[1] for p in pairs),
                                    "first_start_to_last_end_us": statistics.mean(p[2] for p in pairs)}
# ---- the batched prover shapes alone (tools/dbg/ntt_batched.py: 11 launches per shape)
f = newest(f"{src}/prof_ntt12/runc/*_kernel_trace.csv")
by = {}
for r in csv.DictReader(open(f)):
    if "ntt_" in r["Kernel_Name"]:
        by.setdefault((r["Kernel_Name"].split("(")[0].split("::")[-1], r["Grid_Size_X"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
summary["ntt_batched_isolated"] = {}
for (k, grid), v in by.items():
    us = statistics.median(v)
    pts = {"ntt_rows_v2_kernel<12, 0, true>": 8192 << 12, "ntt_rows_v2_kernel<12, 0, false>": 8192 << 12, "ntt_rows_v2_kernel<13, 0, false>": 4096 << 13,
           "ntt_rows_v2_kernel<10, 2, false>": 32768 << 10}.get(k)
    summary["ntt_batched_isolated"][f"{k} grid {grid}"] = {"median_us": us, "launches": len(v), "GBps": 16.0 * pts / us / 1e3 if pts else None, "frac_of_8TBps": 16.0 * pts / us / 1e3 / 8000 if pts else None}
# ---- kernel shares of a single-worker table step (un-overlapped durations); the kernel legs of the bench taken out by name
rows_ = list(csv.DictReader(open(f"{dst}/table_1worker_kernel_stats.csv")))
rows_ = [r for r in rows_ if "hash_no_pad_batch" not in r["Name"] and "ntt_cols_v2_kernel<10" not in r["Name"]]
agg = {}
for r in rows_:
    n = r["Name"]
    key = ("witness replay (latency kernel, <= 1 block per proof)" if "witness_exec" in n else "gate constraints" if "gate_constraints_lde" in n else "witness check" if "gate_check" in n or "zpp_wrap" in n else
           "leaf sponge" if "leaf_hash" in n else "merkle levels" if "merkle_level_kernel" in n else "merkle top levels (latency)" if "merkle_level_wave" in n else
           "NTT/LDE" if "ntt_" in n or "scale_powers" in n else "PoW" if "pow_kernel" in n else "quotient perm" if "quotient_perm" in n else
           "Z/partial products" if "zpp" in n else "transcript (latency)" if "ch_" in n else "multiset digest" if "row_digest" in n or "map_to_curve" in n or "sum_" in n else "other")
    agg[key] = agg.get(key, 0) + float(r["TotalDurationNs"])
tot = sum(agg.values())
summary["table_1worker_shares_pct"] = {k: round(100 * v / tot, 1) for k, v in sorted(agg.items(), key=lambda x: -x[1])}
thr = sum(v for k, v in agg.items() if "latency" not in k)
summary["table_1worker_throughput_kernel_seconds"] = thr / 1e9
json.dump(summary, open(f"{dst}/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
