"""Turn the outputs of tools/dbg/profile_r05.sh / icache_ab.sh / table_2p20_call.sh (gpurun_out/r05/, gpurun_out/) into the committed
summaries under profiles/r05/: bench lines, kernel-stats CSVs and per-grid views of the table build (natural degrees and the
reference-equivalent k = 13 regime, four workers and one), the chip-wide VALU totals of a step, the leaf sponge's counters
(sponge_counters.json: what bench.py's `roofline_alu` reads) and summary.json. Missing inputs are skipped: the parts run in
separate gpurun calls."""
import csv, glob, json, os, re, shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(ROOT, "gpurun_out", "r05"), os.path.join(ROOT, "profiles", "r05")
os.makedirs(dst, exist_ok=True)
summary_path = os.path.join(dst, "summary.json")
summary = json.load(open(summary_path)) if os.path.exists(summary_path) else {}


def newest(pattern):
    g = glob.glob(pattern)
    return max(g, key=os.path.getmtime) if g else None


def last_json_line(path):
    if not os.path.exists(path):
        return None
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def kernel_rows(path):
    """rocprofv3 --stats kernel_stats.csv -> {short name: (calls, total ns, avg ns, percentage)}"""
    out = {}
    for r in csv.DictReader(open(path)):
        k = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void mp2g::", "").replace("mp2g::", "")
        out[k] = (int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"]))
    return out


GROUPS = (("leaf sponge", ("leaf_hash_poly_major",)), ("other sponges / tree levels", ("merkle_level", "leaf_hash_ext", "leaf_hash_row", "hash_no_pad")),
          ("witness replay", ("witness_exec", "wires_from_rows")), ("transcript", ("ch_wave", "challenger")), ("gate constraints", ("gate_constraints", "gate_check")),
          ("NTT / LDE", ("ntt_", "scale_powers")), ("proof of work", ("pow_kernel",)), ("permutation argument + quotient", ("zpp_", "quotient_perm", "zs_")),
          ("FRI (openings, fold, queries)", ("fri_", "openings", "compose", "divide", "combine", "fold_", "query")), ("lookup", ("lookup", "lut_")),
          ("Ecgfp5 digests", ("row_digest", "map_to_curve", "sum_ranges", "sum_kernel", "scalar_mul")), ("copies", ("forest_copy", "copy_rows", "Memcpy", "memset", "fill")))


def shares(rows):
    tot = sum(v[1] for v in rows.values())
    out, seen = {}, set()
    for name, pats in GROUPS:
        t = sum(v[1] for k, v in rows.items() if k not in seen and any(p in k for p in pats))
        seen.update(k for k in rows if any(p in k for p in pats))
        out[name] = round(100 * t / tot, 2)
    out["everything else"] = round(100 * sum(v[1] for k, v in rows.items() if k not in seen) / tot, 2)
    return out, tot


for tag, name in (("prof4", "table_4workers"), ("prof1", "table_1worker"), ("k13_prof4", "table_k13_4workers"), ("k13_prof1", "table_k13_1worker"), ("sponge_trace", "sponge_alone"),
                  ("prof_ntt", "ntt_2p22")):
    f = newest(f"{src}/{tag}/*/*_kernel_stats.csv")
    if not f:
        continue
    shutil.copy(f, f"{dst}/{name}_kernel_stats.csv")
    if os.path.exists(f"{src}/{tag}_by_grid.txt"):
        shutil.copy(f"{src}/{tag}_by_grid.txt", f"{dst}/{name}_by_grid.txt")
    rows = kernel_rows(f)
    line = last_json_line(f"{src}/{tag}.json")
    if line is not None:
        open(f"{dst}/bench_r05_under_rocprof_{name}.json", "w").write(json.dumps(line, indent=1) + "\n")
    if name.startswith("table"):
        sh, tot = shares(rows)
        e = {"kernel_time_s": tot / 1e9, "shares_percent": sh, "top": [[k, v[0], round(v[1] / 1e6, 1), round(v[3], 2)] for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1])[:12]]}
        if line is not None:
            e["proofs_per_s_under_rocprof"] = line["value"]
            leaf = next((v for k, v in rows.items() if k.startswith("leaf_hash_poly_major_kernel<0>")), None)
            if leaf and line["config"].get("leaf_sponge_permutations_process_total"):
                e["leaf_sponge_in_step_perms_per_s"] = line["config"]["leaf_sponge_permutations_process_total"] / (leaf[1] / 1e9)
        summary[name] = e

for a, b in (("bench.json", "bench_r05.json"), ("bench_final.json", "bench_r05_final.json")):
    d = last_json_line(f"{src}/{a}")
    if d is None:
        continue
    open(f"{dst}/{b}", "w").write(json.dumps(d, indent=1) + "\n")
    summary[b] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "verified": d["verified"],
                  "by_base_degree": {k: v.get("value") for k, v in (d.get("by_base_degree") or {}).items()},
                  "config2": (d.get("config2") or {}).get("value"), "roofline_frac": d["roofline"]["frac"], "ntt_us": d["roofline"]["launch_ms"] * 1e3,
                  "roofline_alu": d.get("roofline_alu"), "commit_135x2p15": {k: d["commit_135x2p15"][k] for k in ("seconds", "lde_GBps", "merkle_permutations_per_s")} if "commit_135x2p15" in d else None}

for f in ("sweep.txt", "icache_ab.txt", "ubench.txt", "sponge_trace.txt", "variants_ab.txt"):
    if os.path.exists(f"{src}/{f}"):
        shutil.copy(f"{src}/{f}", f"{dst}/{f}")

# ---- chip-wide VALU totals of a table step (per-kernel sums are exact under --pmc; the dispatches are serialised, so the wall time is not)
if os.path.exists(f"{src}/pmc4_summary.json"):
    shutil.copy(f"{src}/pmc4_summary.json", f"{dst}/step_counters_4workers.json")
    p = json.load(open(f"{src}/pmc4_summary.json"))
    line = last_json_line(f"{src}/pmc4.json")
    summary["step_counters"] = {"SQ_INSTS_VALU_total": p["totals"].get("SQ_INSTS_VALU"), "framework_proofs_in_the_run": (line or {}).get("table_rows_total", 512) * 5,
                                "note": p["note"]}

# ---- the leaf sponge alone: instructions per permutation, achieved issue rate, the chip's plain-instruction issue rate
if os.path.exists(f"{src}/sponge_pmc_summary.json"):
    shutil.copy(f"{src}/sponge_pmc_summary.json", f"{dst}/sponge_pmc_summary.json")
    p = json.load(open(f"{src}/sponge_pmc_summary.json"))
    k = next(v for n, v in p["kernels"].items() if n.startswith("leaf_hash_poly_major_kernel<0>"))
    perms = k["dispatches"] * 17 * (1 << 20)
    out = {"kernel": "leaf_hash_poly_major_kernel<0>", "launch": "2^20 leaves x 135 limbs = 17 permutations per lane (tools/dbg/commit_only.py)", "dispatches": k["dispatches"],
           "command": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -- python3 tools/dbg/commit_only.py",
           "counters_summed_over_dispatches": {c: v for c, v in k.items() if c != "dispatches"},
           "valu_insts_per_perm": k["SQ_INSTS_VALU"] * 64 / perms, "salu_insts_per_perm": k.get("SQ_INSTS_SALU", 0) * 64 / perms,
           "gui_cycles_per_launch": k["GRBM_GUI_ACTIVE"] / 8 / k["dispatches"],  # the counter sums the 8 XCDs
           "cycles_per_valu_wave_inst_achieved": k["GRBM_GUI_ACTIVE"] / 8 * 1024 / k["SQ_INSTS_VALU"]}
    st = newest(f"{src}/sponge_trace/*/*_kernel_stats.csv")
    if st:
        leaf = next(v for n, v in kernel_rows(st).items() if n.startswith("leaf_hash_poly_major_kernel<0>"))
        out["avg_launch_us_kernel_trace"] = leaf[2] / 1e3
        out["isolated_perms_per_s_kernel_trace"] = 17 * (1 << 20) / (leaf[2] / 1e9)
        out["sclk_hz"] = out["gui_cycles_per_launch"] / (leaf[2] / 1e9)  # shader cycles of a launch / its duration (the trace and the counter pass are separate runs of the same program)
    ub = f"{src}/ubench.txt"
    if os.path.exists(ub):
        m = re.search(r"add32\s+[\d.]+ ms\s+([\d.]+) Gop/s", open(ub).read())
        if m:
            out["add32_wave_insts_per_s_measured"] = float(m.group(1)) * 1e9 / 64
    # the chip's VALU issue peak: a wave64 instruction takes 2 cycles on a SIMD-32 (MI355X_MICROARCH.md, 'v_fma_f32 (wave64): 2 cyc'), 1024 SIMDs
    if "sclk_hz" in out:
        out["peak_valu_wave_insts_per_s"] = 1024 * out["sclk_hz"] / 2.0
        out["cycles_per_valu_wave_inst_of_the_mix"] = 2.0
    for name in ("table_1worker", "table_4workers"):
        if "leaf_sponge_in_step_perms_per_s" in summary.get(name, {}):
            out["in_step_perms_per_s_" + name] = summary[name]["leaf_sponge_in_step_perms_per_s"]
    if "in_step_perms_per_s_table_1worker" in out:
        out["in_step_perms_per_s"] = out["in_step_perms_per_s_table_1worker"]
        out["in_step_source"] = ("profiles/r05/table_1worker_kernel_stats.csv: permutations queued by the process (mp2g_stat_leaf_permutations) / the leaf kernel's summed duration, ONE worker "
                                 "(un-overlapped launches; with four workers a launch's duration includes the time it shares the chip)")
    if "SQ_INSTS_VALU_total" in summary.get("step_counters", {}):
        # every VALU wave-instruction of a 512-row table build (all kernels; step_counters_4workers.json) per framework proof: with the
        # build's proofs/s it is the rate at which the whole chip issues VALU instructions during a build
        line = last_json_line(f"{src}/pmc4.json")
        share = 1.0
        if line and line["config"].get("leaf_sponge_permutations_process_total"):
            share = line["config"]["leaf_sponge_permutations"] / line["config"]["leaf_sponge_permutations_process_total"]  # the timed block's part of the process (the rest: prover creation)
        out["step_valu_wave_insts_per_framework_proof"] = summary["step_counters"]["SQ_INSTS_VALU_total"] * share / summary["step_counters"]["framework_proofs_in_the_run"]
        out["step_source"] = "profiles/r05/step_counters_4workers.json: SQ_INSTS_VALU summed over every dispatch of a 512-row table build (2560 framework proofs), scaled by the timed block's share of the process's sponge work"
    json.dump(out, open(f"{dst}/sponge_counters.json", "w"), indent=1)
    summary["sponge_counters"] = {k_: out[k_] for k_ in ("valu_insts_per_perm", "cycles_per_valu_wave_inst_achieved", "isolated_perms_per_s_kernel_trace", "sclk_hz", "in_step_perms_per_s") if k_ in out}
    if "SQ_INSTS_VALU_total" in summary.get("step_counters", {}):
        sc = summary["step_counters"]
        sc["valu_wave_insts_per_framework_proof"] = sc["SQ_INSTS_VALU_total"] / sc["framework_proofs_in_the_run"]

# ---- HBM traffic of the 2^22 NTT (TCC counters, separate passes; FETCH_SIZE calibrated in the same run on scale_powers_kernel, which
# reads exactly 32768 KB with 8 B / lane loads: the guide's correction for that access pattern)
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
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, WRITE_SIZE) --kernel-trace --output-format csv -- python3 tools/dbg/traffic_run.py (tools/dbg/profile_r05.sh ntt)",
               "units": "KB per dispatch as reported (mean over the dispatches of each kernel); FETCH_SIZE scaled by the factor calibrated in this same run on scale_powers_kernel",
               "fetch_calibration_factor": cal,
               "ntt_2p22_forward_bitrev": {"fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes": fetch + write, "algorithmic_bytes": 16 << 22,
                                           "note": f"two launches ({cols}, {rows_k}); pass A also streams the 32 MiB 4-step twiddle table"}},
              open(f"{dst}/ntt_traffic.json", "w"), indent=1)
    summary["ntt_traffic_bytes"] = fetch + write

# ---- the completed 2^20-row table
rec = os.path.join(ROOT, "gpurun_out", "table_2p20_rows.json")
if os.path.exists(rec):
    shutil.copy(rec, f"{dst}/table_2p20_rows.json")
    r = json.load(open(rec))
    summary["table_2p20_rows"] = {k: r[k] for k in ("table_rows_total", "framework_proofs", "gpu_seconds", "value", "join_levels", "gpu_seconds_join_levels")}
json.dump(summary, open(summary_path, "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
