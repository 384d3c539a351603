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
