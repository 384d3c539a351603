"""Turn the outputs of tools/dbg/profile_r06.sh (gpurun_out/r06/) into the committed summaries under profiles/r06/: the rocprofv3
kernel-stats CSVs of the roofline leg, of the leaf kernel alone and of a four- / one-worker table build on this round's code, the 2^22
NTT's HBM traffic (ntt_traffic.json) and the leaf sponge's counters (sponge_counters.json: what bench.py's `roofline_alu` reads, newest
round first), and summary.json. Missing inputs are skipped."""
import csv, glob, json, os, re, shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(ROOT, "gpurun_out", "r06"), os.path.join(ROOT, "profiles", "r06")
os.makedirs(dst, exist_ok=True)
summary_path = os.path.join(dst, "summary.json")
summary = json.load(open(summary_path)) if os.path.exists(summary_path) else {}


def last_json_line(path):
    if not os.path.exists(path):
        return None
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def kernel_rows(path):
    out = {}
    for r in csv.DictReader(open(path)):
        k = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void mp2g::", "").replace("mp2g::", "")
        out[k] = (int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"]))
    return out


for tag, name in (("prof_ntt", "ntt_2p22"), ("sponge_trace", "sponge_alone"), ("prof4", "table_4workers"), ("prof1", "table_1worker")):
    f = f"{src}/{tag}_kernel_stats.csv"
    if not os.path.exists(f):
        continue
    shutil.copy(f, f"{dst}/{name}_kernel_stats.csv")
    rows = kernel_rows(f)
    line = last_json_line(f"{src}/{tag if tag != 'prof_ntt' else 'ntt'}.json")
    e = {"top": [[k, v[0], round(v[1] / 1e6, 2), round(v[2] / 1e3, 2), round(v[3], 2)] for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1])[:10]], "columns": "kernel, calls, total ms, avg us, %"}
    if line is not None:
        open(f"{dst}/bench_r06_under_rocprof_{name}.json", "w").write(json.dumps(line, indent=1) + "\n")
        if line.get("value"):
            e["proofs_per_s_under_rocprof"] = line["value"]
        if name.startswith("table"):
            leaf = next((v for k, v in rows.items() if k.startswith("leaf_hash_poly_major_kernel<0>")), None)
            if leaf and line["config"].get("leaf_sponge_permutations_process_total"):
                e["leaf_sponge_in_step_perms_per_s"] = line["config"]["leaf_sponge_permutations_process_total"] / (leaf[1] / 1e9)
        if name == "ntt_2p22":
            a = next(v for k, v in rows.items() if k.startswith("ntt_cols_v2_kernel<10"))
            b = next(v for k, v in rows.items() if k.startswith("ntt_rows_v2_kernel<12, 0, true"))
            e["kernels_avg_us"] = [a[2] / 1e3, b[2] / 1e3]
            e["bench_launch_us_between_events"] = line["roofline"]["launch_ms"] * 1e3
            e["frac_of_hbm_peak_from_the_trace"] = (16 << 22) / ((a[2] + b[2]) / 1e9) / 8e12
    summary[name] = e

if os.path.exists(f"{src}/sponge_pmc_summary.json"):
    shutil.copy(f"{src}/sponge_pmc_summary.json", f"{dst}/sponge_pmc_summary.json")
    p = json.load(open(f"{src}/sponge_pmc_summary.json"))
    k = next(v for n, v in p["kernels"].items() if n.startswith("leaf_hash_poly_major_kernel<0>"))
    perms = k["dispatches"] * 17 * (1 << 20)
    old = json.load(open(os.path.join(ROOT, "profiles", "r05", "sponge_counters.json")))
    out = {"kernel": "leaf_hash_poly_major_kernel<0>", "launch": "2^20 leaves x 135 limbs = 17 permutations per lane (tools/dbg/commit_only.py)", "dispatches": k["dispatches"],
           "command": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 tools/dbg/commit_only.py (tools/dbg/profile_r06.sh sponge)",
           "counters_summed_over_dispatches": {c: v for c, v in k.items() if c != "dispatches"},
           "valu_insts_per_perm": k["SQ_INSTS_VALU"] * 64 / perms, "salu_insts_per_perm": k.get("SQ_INSTS_SALU", 0) * 64 / perms,
           "gui_cycles_per_launch": k["GRBM_GUI_ACTIVE"] / 8 / k["dispatches"],
           "cycles_per_valu_wave_inst_achieved": k["GRBM_GUI_ACTIVE"] / 8 * 1024 / k["SQ_INSTS_VALU"]}
    st = f"{src}/sponge_trace_kernel_stats.csv"
    if os.path.exists(st):
        leaf = next(v for n, v in kernel_rows(st).items() if n.startswith("leaf_hash_poly_major_kernel<0>"))
        out["avg_launch_us_kernel_trace"] = leaf[2] / 1e3
        out["isolated_perms_per_s_kernel_trace"] = 17 * (1 << 20) / (leaf[2] / 1e9)
        out["sclk_hz"] = out["gui_cycles_per_launch"] / (leaf[2] / 1e9)
        out["peak_valu_wave_insts_per_s"] = 1024 * out["sclk_hz"] / 2.0
        out["cycles_per_valu_wave_inst_of_the_mix"] = 2.0
    ub = os.path.join(dst, "ubench.txt")
    if os.path.exists(ub):
        m = re.search(r"add32\s+[\d.]+ ms\s+([\d.]+) Gop/s", open(ub).read())
        if m:
            out["add32_wave_insts_per_s_measured"] = float(m.group(1)) * 1e9 / 64
    for name in ("table_1worker", "table_4workers"):
        if "leaf_sponge_in_step_perms_per_s" in summary.get(name, {}):
            out["in_step_perms_per_s_" + name] = summary[name]["leaf_sponge_in_step_perms_per_s"]
    if "in_step_perms_per_s_table_1worker" in out:
        out["in_step_perms_per_s"] = out["in_step_perms_per_s_table_1worker"]
        out["in_step_source"] = ("profiles/r06/table_1worker_kernel_stats.csv: permutations queued by the process (mp2g_stat_leaf_permutations) / the leaf kernel's summed duration, ONE worker "
                                 "(un-overlapped launches; with four workers a launch's duration includes the time it shares the chip)")
    # the chip-wide instruction count of a build was taken in round 5 (step_counters_4workers.json); the build's kernels are unchanged
    for key in ("step_valu_wave_insts_per_framework_proof", "step_source"):
        if key in old:
            out[key] = old[key]
    json.dump(out, open(f"{dst}/sponge_counters.json", "w"), indent=1)
    summary["sponge_counters"] = {k_: out[k_] for k_ in ("valu_insts_per_perm", "cycles_per_valu_wave_inst_achieved", "isolated_perms_per_s_kernel_trace", "sclk_hz", "in_step_perms_per_s") if k_ in out}

if os.path.exists(f"{src}/traffic_FETCH_SIZE_summary.json") and os.path.exists(f"{src}/traffic_WRITE_SIZE_summary.json"):
    fs = json.load(open(f"{src}/traffic_FETCH_SIZE_summary.json"))["kernels"]
    ws = json.load(open(f"{src}/traffic_WRITE_SIZE_summary.json"))["kernels"]
    per = lambda d, k, c: d[k][c] / d[k]["dispatches"]
    sp = next(k for k in fs if "scale_powers" in k)
    cal = 32768.0 / per(fs, sp, "FETCH_SIZE")
    cols = next(k for k in fs if "ntt_cols" in k and "kernel<10" in k)
    rows_k = next(k for k in fs if "ntt_rows" in k and "nat" not in k and "kernel<12, 0" in k)
    fetch = (per(fs, cols, "FETCH_SIZE") + per(fs, rows_k, "FETCH_SIZE")) * cal * 1024
    write = (per(ws, cols, "WRITE_SIZE") + per(ws, rows_k, "WRITE_SIZE")) * 1024
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, WRITE_SIZE) --kernel-trace --output-format csv -- python3 tools/dbg/traffic_run.py (tools/dbg/profile_r06.sh ntt)",
               "units": "KB per dispatch as reported (mean over the dispatches of each kernel); FETCH_SIZE scaled by the factor calibrated in this same run on scale_powers_kernel (reads exactly 32768 KB with 8 B / lane loads: MI355X_MICROARCH.md's correction for that pattern)",
               "fetch_calibration_factor": cal,
               "ntt_2p22_forward_bitrev": {"fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes": fetch + write, "algorithmic_bytes": 16 << 22,
                                           "note": f"two launches ({cols}, {rows_k}); pass A also streams the 32 MiB 4-step twiddle table"}},
              open(f"{dst}/ntt_traffic.json", "w"), indent=1)
    summary["ntt_traffic_bytes"] = fetch + write
json.dump(summary, open(summary_path, "w"), indent=1)
print(json.dumps(summary, indent=1)[:3500])
