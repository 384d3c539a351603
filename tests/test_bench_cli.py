"""bench.py's command line on a box without GPUs: `--gpus N` starts its own ranks only when N devices are visible, and says so
otherwise (before anything touches a GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_devices_is_refused_with_a_message():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MP2G_BENCH_BACKEND")}
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box: no device visible to this child
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "--gpus 2 needs 2 visible GPUs" in r.stderr + r.stdout


def test_default_workload_is_the_table_build():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "table (default, the headline)" in " ".join(r.stdout.split())


def test_resume_dir_is_a_single_gpu_mode(tmp_path):
    """`--resume-dir` builds the table's blocks one after another on one GPU; asked for together with several GPUs it says so before
    any rank is started or any GPU touched"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MP2G_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--resume-dir", str(tmp_path / "s")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "--resume-dir builds the blocks one after another on ONE GPU" in r.stderr + r.stdout
    assert not (tmp_path / "s").exists()


def test_block_plan_rows_per_rank():
    """the rows of a rank's block: without --rows the block is 2^17 rows whatever --steps says, so that 8 ranks build the metric's
    2^20-row table (+ 7 separator rows) and 1 / 2 / 4 ranks the same block per rank; the warm-up block never exceeds 5120 rows"""
    sys.path.insert(0, ROOT)
    import bench
    n, per_step, warm = bench.block_plan(None, 20, 5)  # the driver's command
    assert n == 1 << 17 and per_step == 6554 and warm == 5120 and 19 * per_step < n <= 20 * per_step
    assert 8 * n == 1 << 20 and 8 * n + 7 == 1048583
    assert bench.block_plan(None, 5, 1)[0] == 1 << 17 and bench.block_plan(None, 5, 0)[2] == 0
    assert bench.block_plan(1024, 20, 5) == (20480, 1024, 5120) and bench.block_plan(1024, 20, 9)[2] == 5120
    assert bench.block_plan(8, 2, 1) == (16, 8, 8)
    assert bench.block_plan(None, 4, 1, default_rows=16) == (16, 4, 4)
    os.environ["MP2G_BENCH_BLOCK_ROWS"] = "24"
    try:
        assert bench.block_plan(None, 5, 2) == (24, 5, 10)
    finally:
        del os.environ["MP2G_BENCH_BLOCK_ROWS"]


def test_cpu_throughput_sweep_reports_the_best_mode():
    """bench.py's cpu_baseline as THROUGHPUT: the sampled framework proofs proved P side by side x T OpenMP threads each by the oracle;
    the value reported is the best mode's. Here on a small circuit (the GPU run hands it the timed block's captured witnesses)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import bench
    import circuits as C
    import oracle as O
    ckt = C.build(5, C.ALL_KINDS, 1)
    ofp = O.standard_params(5, (ckt.num_constants + C.NUM_ROUTED, C.NUM_WIRES, 20, 16), pow_bits=4, num_queries=2)
    cd = O.rand_field(4, 2)
    chain = lambda name: [(f"{name} step {i}", ckt, ofp, cd, (lambda: ckt.wires), ckt.pi_hash) for i in range(2)]
    samples = [chain("cells_leaf"), chain("cells_full"), chain("row_leaf"), chain("row_full")]
    best, sweep = bench.cpu_throughput(samples, seconds_per_mode=0.3, modes=[(2, 1), (1, 2)])
    assert len(sweep["modes"]) == 2 and all(m["framework_proofs"] >= m["concurrent_proofs"] and m["proofs_per_s"] > 0 for m in sweep["modes"])
    assert best["proofs_per_s"] == max(m["proofs_per_s"] for m in sweep["modes"]) and sweep["memory_cap"]["max_concurrent"] >= 1
    # the default modes: P x T = the host's threads, capped by memory
    _, sweep = bench.cpu_throughput(samples, seconds_per_mode=0.05)
    cores = os.cpu_count()
    assert [(m["concurrent_proofs"], m["threads_per_proof"]) for m in sweep["modes"]][0] == (cores, 1)


def test_cpu_farm_runs_the_throughput_modes_as_processes():
    """the farm behind cpu_baseline's throughput modes: a helper process (started before the GPU is touched in a real run) spawns P
    worker processes per mode, each maps the spooled samples and proves framework proofs with T OpenMP threads; here two modes on a
    small circuit, and the fallback to threads when no farm is there"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    import circuits as C
    import oracle as O
    ckt = C.build(5, C.ALL_KINDS, 1)
    ofp = O.standard_params(5, (ckt.num_constants + C.NUM_ROUTED, C.NUM_WIRES, 20, 16), pow_bits=4, num_queries=2)
    cd = O.rand_field(4, 2)
    chain = lambda name: [(f"{name} step {i}", ckt, ofp, cd, (lambda: ckt.wires), ckt.pi_hash) for i in range(2)]
    samples = [chain("cells_leaf"), chain("cells_full"), chain("row_leaf"), chain("row_full")]
    farm = bench.CpuFarm()
    try:
        assert farm.p is not None
        best, sweep = bench.cpu_throughput_modes(samples, farm, 0.3)
        assert "separate processes" in sweep["how"] and len(sweep["modes"]) >= 1
        cores = os.cpu_count()
        assert sweep["modes"][0]["threads_per_proof"] in (1, 4) and sweep["modes"][0]["concurrent_proofs"] * sweep["modes"][0]["threads_per_proof"] <= cores
        assert all(m["workers_failed"] == 0 and m["framework_proofs"] >= m["concurrent_proofs"] and m["proofs_per_s"] > 0 for m in sweep["modes"])
        assert best["proofs_per_s"] == max(m["proofs_per_s"] for m in sweep["modes"])
    finally:
        farm.close()
    best, sweep = bench.cpu_throughput_modes(samples, None, 0.1)
    assert "threads of one process" in sweep["how"] and best["proofs_per_s"] > 0
