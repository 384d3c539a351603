"""BASELINE configs[3] on the GPU: the table-creation flow of mapreduce-plonky2_amd/table.py (cells-tree + row-tree framework
proofs scheduled by the batched work plan) through the HIP prover, its off-circuit side against the oracle, and bench.py's
default workload including its self-launch on two ranks."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import circuits as C
import oracle as O
from table_oracle import OracleTableWitness, o_sum

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = importlib.import_module("mapreduce-plonky2_amd.table")
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
FW = importlib.import_module("mapreduce-plonky2_amd.framework")
IX = importlib.import_module("mapreduce-plonky2_amd.indexing")


def test_curve_sum_ranges_vs_oracle(ctx, mp2):
    w = mp2.map_to_curve_batch(ctx, O.rand_field((150, 9), 77))
    ranges = [(0, 150), (0, 1), (3, 3), (5, 70), (64, 129), (149, 150), (10, 75), (0, 64), (1, 66)]
    gw, gwei = mp2.curve_sum_ranges(ctx, w, ranges)
    for (lo, hi), a, b in zip(ranges, gw, gwei):
        ow, owei = o_sum(w[lo:hi]) if hi > lo else (np.zeros(5, dtype=np.uint64), np.array([0] * 10 + [1], dtype=np.uint64))
        assert np.array_equal(a, ow) and np.array_equal(b, owei), (lo, hi)
    with pytest.raises(mp2.Mp2gError, match="range"):
        mp2.curve_sum_ranges(ctx, w, [(0, 151)])
    # no ranges, no points
    assert mp2.curve_sum_ranges(ctx, w, np.zeros((0, 2), dtype=np.uint32))[0].shape == (0, 5)


def test_row_digests_sum_to_the_table_digest(ctx, mp2):
    """the per-row terms: each equals the oracle's one-row digest, and their sum is compute_table_row_digest"""
    rng = np.random.default_rng(5)
    ids = O.rand_field(5, 0xC0FFEE04)
    values = rng.integers(0, 1 << 32, size=(70, 5, 8), dtype=np.uint32)
    w, wei = mp2.row_digests(ctx, ids, values, values[:, 0:1])
    tw, twei = mp2.compute_table_row_digest(ctx, ids, values, values[:, 0:1])
    assert np.array_equal(mp2.curve_sum(ctx, w), tw)
    for r in (0, 1, 69):
        ow, owei = np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
        O.lib().orc_row_digest_batch(0, O.p(ids), O.sz(5), O.p(O.arr(values[r:r + 1], np.uint32)), O.p(O.arr(values[r:r + 1, 0:1], np.uint32)), O.sz(1), O.sz(1), O.p(ow), O.p(owei))
        assert np.array_equal(w[r], ow) and np.array_equal(wei[r], owei)


def test_table_witness_matches_the_oracle(ctx, mp2):
    table = T.SyntheticTable(7, 4, seed=0xC0FFEE04, block=2)
    assert all(table.secondary_int(i) < table.secondary_int(i + 1) for i in range(6)) and int(table.values[0, 0, 0]) >> 16 == 2
    root, nodes, spans = T.balanced_bst(7)
    g, o = T.TableWitness(ctx, table, spans), OracleTableWitness(table, spans)
    assert np.array_equal(g.cell_digest, o.cell_digest) and np.array_equal(g.unique, o.unique)
    assert np.array_equal(g.row_w, o.row_w) and np.array_equal(g.row_own, o.row_own)
    for k in spans:
        assert np.array_equal(g.row_digest[k], o.row_digest[k]) and np.array_equal(g.root_digest_w[k], o.root_digest_w[k])


def test_prover_serves_narrower_batches(ctx, mp2):
    """mp2g_prover_set_active: a prover created for 6 proofs proving 2 gives what a prover created for 2 gives, and goes back to 6"""
    ckt = C.build(7, C.VERIFIER_KINDS, 3)
    wide, narrow = FW.CircuitProver(ctx, ckt, 6, bind_public_inputs=True), FW.CircuitProver(ctx, ckt, 2, bind_public_inputs=True)
    d_w = FW.tile_witness(ctx, ckt, 6, 11)
    ph = O.rand_field((6, 4), 12)
    d_ph = ctx.to_device(ph)
    wide.prove(d_w, d_ph)
    full = wide.results()
    wide.pr.set_active(2)
    wide.prove(d_w, d_ph)
    narrow.prove(d_w, d_ph)
    a, b = wide.results(), narrow.results()
    assert a[0].shape[0] == 2 and all(np.array_equal(x, y) for x, y in zip(a, b))
    assert all(np.array_equal(x[:2], y) for x, y in zip(full, a))
    wide.pr.set_active(6)
    wide.prove(d_w, d_ph)
    assert all(np.array_equal(x, y) for x, y in zip(full, wide.results()))
    with pytest.raises(mp2.Mp2gError):
        wide.pr.set_active(7)
    wide.free(); narrow.free()


@pytest.fixture(scope="module")
def params(ctx):
    prover = FW.GpuProver(ctx, capacity=8)
    p = T.TableParams(prover, FW.circuit_fri_params, IX.empty_poseidon_hash(ctx))
    yield p
    prover.free()


def test_table_build_of_eleven_rows(ctx, mp2, params):
    """11 rows x (4 cells-tree proofs + 1 row-tree proof) = 55 real framework proofs, scheduled by the batched work plan over two
    workers; the row tree has leaves, partial and full nodes. The root exposes the off-circuit tree hash, the table's multiset
    digest, min / max, and passes the oracle's verifier; a cells proof made by the Python builder equals the batch path's."""
    n = 11
    table = T.SyntheticTable(n, 4, seed=0xC0FFEE04)
    root, nodes, spans = T.balanced_bst(n)
    kinds = sorted({sum(c is not None for c in nodes[k]) for k in nodes})
    assert kinds == [0, 1, 2]
    ctx2 = mp2.Context(0)
    provers = [params.cells.prover, FW.GpuProver(ctx2, capacity=8)]
    build = T.TableBuild(params, [R.ProofSession(p) for p in provers], batch=8, subtree_size=4, host_threads=8)
    wit = T.TableWitness(ctx, table, spans)
    proof, name = build.run(table, wit, root, nodes)
    assert name == "row_full" and build.n_proofs == 5 * n
    pis = proof[3]
    want = T.expected_root_public_inputs(ctx, table, wit, root, nodes, spans)
    assert np.array_equal(pis[:T.ROWS_IO], want)
    assert np.array_equal(pis[T.ROWS_IO:], np.asarray(params.rows.set_digest, dtype=np.uint64))
    assert np.array_equal(pis[4:15], mp2.compute_table_row_digest(ctx, table.col_ids, table.values, table.values[:, 0:1])[1])
    wckt, wcap, wdig = params.rows.chains["row_full"][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(pis, 4), *proof[:3]) == 0
    # every row's cells root exposes the row's cells-tree hash (indexing.cells hashes are part of expected_root_public_inputs) and 4 cells
    for r in (0, n - 1):
        cp = build.cells_roots[r][0][3]
        assert int(cp[26]) == 4 and int(cp[27]) == 0 and np.array_equal(cp[4:15], wit.cell_digest[r, 3])
    # scheduling does not change a proof: one worker, batches of three, the plain (node by node) work plan -> the same root, word for word
    plain = T.TableBuild(params, [R.ProofSession(provers[0])], batch=3, subtree_size=1, host_threads=8)
    proof2, name2 = plain.run(table, wit, root, nodes)
    assert name2 == name and plain.n_proofs == 5 * n and all(np.array_equal(a, b) for a, b in zip(proof, proof2))
    # ... and neither does dropping every proof as soon as its parent is proved (what a 2^17-row block does): only the root stays
    lean = T.TableBuild(params, [R.ProofSession(p) for p in provers], batch=8, subtree_size=4, host_threads=8, keep_proofs=False)
    proof3, _ = lean.run(table, wit, root, nodes)
    assert all(np.array_equal(a, b) for a, b in zip(proof, proof3)) and list(lean.row_proofs) == [root] and not lean.cells_roots
    # the builder path (eager Python circuit) of one cells leaf = the witness-program path
    flat = T._u64cat([table.col_ids[1]], table.values[0, 1], [0], wit.cell_digest[0, 0], T.NEUTRAL_FIELDS)
    one = params.cells.generate_proof("cells_leaf", [], [], flat)
    (two,) = params.cells.generate_proofs_batch("cells_leaf", [([], [], flat)])
    assert all(np.array_equal(a, b) for a, b in zip(one, two))
    provers[1].free()
    ctx2.close()


def test_row_node_refuses_a_foreign_cells_proof(ctx, mp2, params):
    """a row-tree proof is verified in place of the cells-tree proof: its digest is not in the cells circuit set (the reference's
    set_circuit_membership_target fails the same way)"""
    table = T.SyntheticTable(1, 4, seed=7)
    root, nodes, spans = T.balanced_bst(1)
    wit = T.TableWitness(ctx, table, spans)
    build = T.TableBuild(params, [R.ProofSession(params.cells.prover)], batch=8, subtree_size=1, host_threads=4)
    row_proof, name = build.run(table, wit, root, nodes)
    with pytest.raises(KeyError):
        params.cells.membership(params.rows.vds[name][1])
    # and a cells proof with a tampered public input makes the row node's witness inconsistent: prove() refuses it
    cells = build.cells_roots[0]
    bad = (cells[0][0], cells[0][1], cells[0][2], cells[0][3].copy())
    bad[3][0] ^= np.uint64(1)
    name, job = build.row_job(table, wit, nodes, 0, (bad, cells[1]), {})
    with pytest.raises(Exception, match="witness"):
        params.rows.generate_proofs_batch(name, [job])


def _bench(args, env=None, timeout=2400):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_default_workload_is_the_table_build():
    """python bench.py (no --workload): the table build as ONE contiguous block of steps x rows rows, self-verifying, with roofline and
    cpu_baseline objects (all-thread / one-thread medians, CPU model), and the side legs of the driver's line at small sizes: BASELINE
    configs[2] (a 64-leaf tree of real proofs here), the table rate at a padded base degree, the prove()-only loop"""
    line = _bench(["--rows", "8", "--steps", "2", "--warmup", "1", "--workers", "2", "--table-batch", "8", "--subtree", "8", "--cpu-budget", "1",
                   "--config2-leaves", "64", "--degree-sweep", "12", "--sweep-rows", "8", "--sweep-runs", "2", "--leaves-leg"])
    assert line["config"]["workload"].startswith("table:") and line["unit"] == "proofs/s" and line["n_gpus"] == 1
    assert line["config"]["rows_per_rank"] == 16 and line["config"]["row_tree_depth"] == 4 and line["steps"] == 2
    assert abs(line["value"] * line["ms_per_step"] * 2 / 1e3 - 5 * 16) < 1e-6  # one 16-row block, 5 framework proofs per row
    assert line["config"]["shapes"]["row_full"] == [14, 13, 12]
    assert line["value"] > 0 and line["verified"] >= 16 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0
    med = line["cpu_baseline"]["medians"]
    assert med["runs"] == 5 and med["one_thread"]["median_s"] >= med["all_threads"]["median_s"] > 0 and line["cpu_baseline"]["cpu_model"]
    assert line["leaves_prove_only"]["value"] > 0 and len(line["config"]["root_public_inputs"]) == T.ROWS_IO + 4
    assert line["config2"]["framework_proofs"] == 127 and line["config2"]["value"] > 0 and line["config2"]["root_verified"]
    k12 = line["by_base_degree"]["12"]
    assert k12["median_of"] == 2 and k12["rows"] == 8 and line["commit_135x2p15"]["merkle_permutations"] == (1 << 18) * 18 - 16 and line["commit_135x2p15"]["lde_GBps"] > 0
    assert line["sponge"]["median_of"] == 7 and line["roofline"]["traffic_source"] and line["config2"]["device_memory_used_bytes"] > 0 and line["config"]["device_memory_used_bytes"] > 0
    assert k12["value"] > 0 and k12["root_verified"] and all(ch[0] >= 12 for ch in k12["shapes"].values()) and k12["shapes"]["cells_leaf"][0] == 12 and k12["shapes"]["cells_leaf"][-1] == 12
    assert line["config"]["device_memory_used_bytes"] > 0 and line["config"]["host_orchestration"]["scheduler"].startswith("native")
    # round 6: the CPU baseline is a throughput (the best of the process-farm modes and the latency mode), the leaf kernel is timed alone
    # and priced against the mix peak, the 2^22 NTT is stated against the VALU roof as well
    cb = line["cpu_baseline"]
    assert cb["value"] >= cb["latency_mode"]["value"] > 0 and cb["mode"] and cb["oracle_build"]["march"] in ("native", "x86-64-v3")
    assert "separate processes" in cb["throughput_sweep"]["how"] and all(m["workers_failed"] == 0 for m in cb["throughput_sweep"]["modes"])
    alu = line["roofline_alu"]
    assert alu["isolated"]["perms_per_s"] > 0 and abs(alu["isolated"]["frac"] - alu["frac_alone_clock_free"]) < 1e-9 and 0.5 < alu["frac_alone_clock_free"] < 1
    assert 1.5e9 < alu["sclk_hz_this_run_inferred"] < 3e9 and alu["mix_model"]["cycles_per_valu_inst_additive"] > 2 and 0 < alu["table_build"]["frac"] < 1
    assert 0 < line["roofline"]["valu"]["frac_of_valu_peak"] < 1 and line["commit_135x2p15"]["leaf_sponge_ms"] > 0 and line["commit_135x2p15"]["tree_levels_ms"] > 0


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: two ranks (gloo rendezvous, both on this box's GPU), one block of rows each, the
    separator row between them proved by rank 0 over both block roots; the run itself asserts that the root's digest is the whole
    table's and its min the first block's"""
    line = _bench(["--gpus", "2", "--rows", "8", "--steps", "1", "--warmup", "1", "--workers", "2", "--table-batch", "8", "--subtree", "4", "--no-leaves-leg",
                   "--no-cpu-baseline"], env={"MP2G_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["config"]["workload"].startswith("table:")
    assert "1 join level" in line["config"]["sharding"]
    # 2 x 8 rows + the separator row, 5 proofs each
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 5 * 17) < 1e-6


def test_bench_resume_dir_builds_the_table_across_calls(tmp_path):
    """`bench.py --resume-dir`: the table as 2 blocks of 8 rows built in TWO calls (the first is held to one block), block roots kept as
    ProofWithVK bytes in the proof store, the second call re-checks the stored root, builds the other block, proves the separator row
    and verifies the root -- which must be the root `--gpus 2` gets for the same table with one rank per block"""
    import glob
    common = ["--rows", "8", "--steps", "1", "--warmup", "1", "--workers", "2", "--table-batch", "8", "--subtree", "4", "--table-blocks", "2",
              "--resume-dir", str(tmp_path / "store")]
    first = _bench(common + ["--max-seconds", "0"])
    assert first["blocks_built_this_call"] == [0] and first["blocks_missing"] == [1]
    assert len(glob.glob(str(tmp_path / "store" / "row_tree_*.bin"))) == 1
    rec = _bench(common)
    assert rec["blocks_built_this_call"] == [1] and rec["table_rows_total"] == 17 and rec["framework_proofs"] == 5 * 17 and rec["join_levels"] == 1
    assert os.path.exists(str(tmp_path / "store" / "table_record.json")) and len(glob.glob(str(tmp_path / "store" / "row_tree_*.bin"))) == 3
    again = _bench(common)  # a third call finds everything there: no block built, the same root
    assert again["blocks_built_this_call"] == [] and again["root_proof_with_vk_fnv1a64"] == rec["root_proof_with_vk_fnv1a64"]
    two = _bench(["--gpus", "2", "--rows", "8", "--steps", "1", "--warmup", "1", "--workers", "2", "--table-batch", "8", "--subtree", "4", "--no-leaves-leg",
                  "--no-cpu-baseline", "--no-verify"], env={"MP2G_BENCH_BACKEND": "gloo"})
    assert two["config"]["root_public_inputs"] == rec["root_public_inputs"]


def test_shared_prover_scratch_changes_no_proof(tmp_path):
    """the provers of a context take their working buffers from one shared scratch (csrc/ctx.h; the default) or own them
    (MP2G_SHARE_SCRATCH=0): the same 8-row table built both ways ends in the same root proof byte for byte, and the shared build plans and
    uses less device memory"""
    common = ["--rows", "8", "--steps", "1", "--warmup", "1", "--workers", "2", "--table-batch", "8", "--subtree", "4", "--table-blocks", "1"]
    shared = _bench(common + ["--resume-dir", str(tmp_path / "a")])
    own = _bench(common + ["--resume-dir", str(tmp_path / "b")], env={"MP2G_SHARE_SCRATCH": "0"})
    assert shared["root_proof_with_vk_fnv1a64"] == own["root_proof_with_vk_fnv1a64"] and shared["root_public_inputs"] == own["root_public_inputs"]
    assert shared["table_rows_total"] == 8 and shared["join_levels"] == 0


@pytest.mark.parametrize("n_cols,rows", [(1, 2), (6, 3)])
def test_other_column_counts(ctx, mp2, params, n_cols, rows):
    """the cells tree follows ryhope's sbbst for any number of value columns: one column = a lone leaf; six columns = leaves 1, 3, 5, a
    full node 2, a partial node 6 whose child is 5, the root 4 over 2 and 6 (sbbst.rs:301-333: the missing right child is replaced by
    the first in-range node down its left spine). Off-circuit data against the oracle, root public inputs against the off-circuit tree."""
    table = T.SyntheticTable(rows, n_cols, seed=0xC0FFEE04 + n_cols)
    root, nodes, spans = T.balanced_bst(rows)
    wit = T.TableWitness(ctx, table, spans)
    o = OracleTableWitness(table, spans)
    assert np.array_equal(wit.cell_digest, o.cell_digest) and all(np.array_equal(wit.row_digest[k], o.row_digest[k]) for k in spans)
    build = T.TableBuild(params, [R.ProofSession(params.cells.prover)], batch=8, subtree_size=2, host_threads=4)
    proof, name = build.run(table, wit, root, nodes)
    assert build.n_proofs == rows * (n_cols + 1)
    pis = proof[3]
    assert np.array_equal(pis[:T.ROWS_IO], T.expected_root_public_inputs(ctx, table, wit, root, nodes, spans))
    assert np.array_equal(pis[4:15], mp2.compute_table_row_digest(ctx, table.col_ids, table.values, table.values[:, 0:1])[1])
    cells_root = build.cells_roots[0]
    assert cells_root[1] == ("cells_leaf" if n_cols == 1 else "cells_full") and int(cells_root[0][3][26]) == n_cols


def test_table_build_of_128_rows(ctx, mp2, params):
    """a block of 128 rows (640 real framework proofs) the way bench.py runs it: three workers with provers of capacity 32, the row tree
    cut into spun-off subtrees of <= 16 rows by the batched work plan (two waves: eight bottom subtrees, then the top of the tree).
    Root = off-circuit tree hash, the block's multiset digest, min / max; the root proof passes the oracle's verifier."""
    n = 128
    table = T.SyntheticTable(n, 4, seed=0xC0FFEE04, block=6)
    root, nodes, spans = T.balanced_bst(n)
    ctxs = [mp2.Context(0) for _ in range(3)]
    provers = [FW.GpuProver(c, capacity=32) for c in ctxs]
    build = T.TableBuild(params, [R.ProofSession(p) for p in provers], batch=32, subtree_size=16, host_threads=8)
    wit = T.TableWitness(ctx, table, spans)
    proof, name = build.run(table, wit, root, nodes)
    assert build.n_proofs == 5 * n and len(build.row_proofs) == n and len(build.cells_roots) == n
    kinds = {}
    for k, (_, nm) in build.row_proofs.items():
        kinds[nm] = kinds.get(nm, 0) + 1
    assert kinds == {"row_leaf": 64, "row_full": 63, "row_partial": 1}
    pis = proof[3]
    assert np.array_equal(pis[:T.ROWS_IO], T.expected_root_public_inputs(ctx, table, wit, root, nodes, spans))
    assert np.array_equal(pis[4:15], mp2.compute_table_row_digest(ctx, table.col_ids, table.values, table.values[:, 0:1])[1])
    assert int(table.values[0, 0, 0]) >> 16 == 6 and np.array_equal(pis[26:34], table.values[0, 0].astype(np.uint64)) and np.array_equal(pis[34:42], table.values[n - 1, 0].astype(np.uint64))
    wckt, wcap, wdig = params.rows.chains[name][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(pis, 4), *proof[:3]) == 0
    for p in provers:
        p.free()
    for c in ctxs:
        c.close()


def test_bench_gpus_4_two_join_levels():
    """four ranks (gloo, sharing this box's GPU): the blocks meet in two join levels -- ranks 1 and 3 hand their roots to 0 and 2, which prove
    the separator rows 1 and 5, then rank 2 hands the joined tree to rank 0, which proves separator 3 over two joined trees. The run asserts
    the root's digest = the digest of all 4 x 6 + 3 rows and its min = block 0's; every rank checks its block root and its sampled proofs."""
    line = _bench(["--gpus", "4", "--rows", "6", "--steps", "1", "--warmup", "1", "--workers", "1", "--table-batch", "8", "--subtree", "4", "--no-leaves-leg",
                   "--no-cpu-baseline"], env={"MP2G_BENCH_BACKEND": "gloo", "MP2G_HANDOFF_PROBE_FAIL": "3"})
    assert line["n_gpus"] == 4 and "2 join level" in line["config"]["sharding"]
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 5 * (4 * 6 + 3)) < 1e-6 and line["verified"] >= 4 * 13
    # the pre-timing probe: rank 3's direct send "failed" (the test knob), its peer's recv was answered by the staged tensor, and every
    # rank chose the staged hand-off for both levels before t0 -- the build still ends on the whole table's root
    assert line["config"]["handoff"]["mode"] == "staged" and "STAGED" in line["config"]["sharding"] and line["table_rows_total"] == 27


def test_bench_gpus_8_three_join_levels():
    """eight ranks (gloo, sharing this box's GPU) -- the shape of the driver's 8-GPU scaling run: eight blocks, seven separator rows,
    three join levels (1 -> 0, 3 -> 2, 5 -> 4, 7 -> 6; 2 -> 0, 6 -> 4; 4 -> 0). The run asserts the root's digest = the digest of all
    8 x 2 + 7 rows and its min = block 0's; every rank checks its block root and its sampled proofs; all eight ranks answer the
    backend's all_reduce before anything is timed."""
    line = _bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--workers", "1", "--table-batch", "4", "--subtree", "4", "--no-leaves-leg",
                   "--no-cpu-baseline"], env={"MP2G_BENCH_BACKEND": "gloo", "MP2G_BENCH_BLOCK_ROWS": "2"})
    assert line["n_gpus"] == 8 and line["config"]["join_levels"] == 3 and "3 join level" in line["config"]["sharding"]
    assert line["config"]["rccl_ranks"] == 8 and line["config"]["ranks_on_host"] == 8 and line["config"]["backend"] == "gloo"
    # the driver's command shape: no --rows, so the rank's block is the default block (2^17 rows; scaled to 2 rows here), a step is
    # 1/steps of it, and 8 ranks build 8 blocks + 7 separator rows; every join pair was probed before t0
    assert line["config"]["rows_per_rank"] == 2 and line["config"]["rows_per_step"] == 1 and line["steps"] == 2 and line["table_rows_total"] == 8 * 2 + 7
    assert line["config"]["handoff"]["mode"] == "host" and "probed before t0" in line["config"]["sharding"]
    assert abs(line["value"] * line["ms_per_step"] * 2 / 1e3 - 5 * (8 * 2 + 7)) < 1e-6 and line["verified"] >= 8 * 10


def test_native_build_equals_the_python_build(ctx, mp2, params):
    """table.NativeTableBuild (the scheduler in C++: mp2g_forest_*, worker threads, level batching, child proofs in a device pool)
    against table.TableBuild (the Python unit loop) on the same 37-row block: the same root proof word for word, the same kept row
    proofs and cells roots, 5 framework proofs per row; the root exposes the off-circuit tree and verifies; an unsatisfied witness
    fails the native build as it fails the Python one (plonky2's prove() panics)."""
    n = 37
    table = T.SyntheticTable(n, 4, seed=0xC0FFEE04, block=2)
    root, nodes, spans = T.balanced_bst(n)
    samples, keep = T.sample_nodes(nodes, spans)
    ctxs = [mp2.Context(0) for _ in range(2)]
    provers = [FW.GpuProver(c, capacity=8) for c in ctxs]
    wit = T.TableWitness(ctx, table, spans)
    py = T.TableBuild(params, [R.ProofSession(p) for p in provers], batch=8, subtree_size=8, host_threads=4)
    want, want_name = py.run(table, wit, root, nodes)
    nb = T.NativeTableBuild(params, provers, batch=8, subtree_size=8, group_rows=16)
    got, got_name = nb.run(table, wit, root, nodes, keep=samples)
    assert got_name == want_name and nb.n_proofs == 5 * n == py.n_proofs
    assert all(np.array_equal(a, b) for a, b in zip(got, want)), "root proof"
    for k in keep | {root}:
        assert nb.row_proofs[k][1] == py.row_proofs[k][1]
        assert all(np.array_equal(a, b) for a, b in zip(nb.row_proofs[k][0], py.row_proofs[k][0])), f"row {k}"
        assert all(np.array_equal(a, b) for a, b in zip(nb.cells_roots[k][0], py.cells_roots[k][0])), f"cells root of row {k}"
    pis = got[3]
    assert np.array_equal(pis[:T.ROWS_IO], T.expected_root_public_inputs(ctx, table, wit, root, nodes, spans))
    wckt, wcap, wdig = params.rows.chains[got_name][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(pis, 4), *got[:3]) == 0
    assert [w[0] for w in nb.wave_log] == [w[0] for w in py.wave_log], "the same waves of work-plan items"
    # a second block through the same object (new forest, same chains)
    again, _ = nb.run(table, wit, root, nodes, keep=samples)
    assert all(np.array_equal(a, b) for a, b in zip(again, want)) and nb.n_proofs == 10 * n
    # a witness that violates a constraint: the value limb of one cell out of u32 range
    bad = T.SyntheticTable(n, 4, seed=0xC0FFEE04, block=2)
    bad_wit = T.TableWitness(ctx, bad, spans)
    bad_wit.unique = bad_wit.unique.copy()
    bad_wit.cell_digest = bad_wit.cell_digest.copy()
    bad.col_ids = bad.col_ids.copy()
    vals = bad.values.astype(np.uint64)
    vals[5, 2, 3] = 1 << 33
    bad.values = vals
    with pytest.raises(mp2.Mp2gError, match="witness"):
        nb.run(bad, bad_wit, root, nodes)
    # ... and the failed build left the workers usable: the queued batches were rolled back, the good block proves again
    once_more, _ = nb.run(table, wit, root, nodes, keep=samples)
    assert all(np.array_equal(a, b) for a, b in zip(once_more, want))
    nb.free()
    for p in provers:
        p.free()
    for c in ctxs:
        c.close()


def test_pipelined_units_equal_the_synchronous_ones(ctx, mp2, params, monkeypatch):
    """the forest's workers queue the next batches of a unit while one runs (csrc/forest.hip: a ring of three batch records per worker,
    two input buffers per chain, witness flags and slot hand-backs confirmed one batch late); MP2G_FOREST_SYNC=1 keeps a
    synchronisation behind every batch. Same proofs either way, also through a pool that only holds the frontier because slots come
    back (a worker that finds it empty waits for them instead of failing), and with 6 value columns (pool sized from the width)"""
    pa = params
    for n_cols, seed, n in ((4, 0xC0FFEE04, 29), (6, 0xC0FFEE0A, 11)):
        table = T.SyntheticTable(n, n_cols, seed=seed, block=8)
        root, nodes, spans = T.balanced_bst(n)
        wit = T.TableWitness(ctx, table, spans)
        ctxs = [mp2.Context(0) for _ in range(2)]
        provers = [FW.GpuProver(c, capacity=4) for c in ctxs]
        roots = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("MP2G_FOREST_SYNC", mode)
            nb = T.NativeTableBuild(pa, provers, batch=4, subtree_size=4, group_rows=8)
            roots[mode], _ = nb.run(table, wit, root, nodes)
            assert nb.n_proofs == (n_cols + 1) * n
            nb.free()
        assert all(np.array_equal(a, b) for a, b in zip(roots["0"], roots["1"]))
        # a pool of 40 / 56 slots for 145 / 77 proofs: two workers x (a unit's 8 rows x 3 / 5 live cells proofs + queued batches) do not fit at once
        monkeypatch.setenv("MP2G_FOREST_SYNC", "0")
        tight = T.NativeTableBuild(pa, provers, batch=4, subtree_size=4, group_rows=8, pool_slots=40 if n_cols == 4 else 56)
        got, _ = tight.run(table, wit, root, nodes)
        assert all(np.array_equal(a, b) for a, b in zip(got, roots["1"]))
        tight.free()
        for p_ in provers:
            p_.free()
        for c in ctxs:
            c.close()


def test_bench_python_build():
    """python bench.py --python-build: the same line by the Python unit loop (the default is the native scheduler: the test of the
    default workload above asserts that): block root checked against the off-circuit tree and the oracle's verifier, one framework
    proof of every circuit kind re-proved by the oracle bit for bit, the scheduler named in the line"""
    line = _bench(["--rows", "8", "--steps", "2", "--warmup", "1", "--workers", "2", "--table-batch", "8", "--subtree", "8", "--python-build", "--no-cpu-baseline",
                   "--config2-leaves", "0", "--degree-sweep", "", "--no-leaves-leg"])
    assert line["config"]["host_orchestration"]["scheduler"].startswith("python") and line["verified"] >= 16
    assert 0 <= line["config"]["host_orchestration"]["host_glue_share"] < 1
    assert abs(line["value"] * line["ms_per_step"] * 2 / 1e3 - 5 * 16) < 1e-6


def test_bench_gpus_2_python_build():
    """two ranks (gloo, sharing this box's GPU) with `--python-build`: table.TableBuild's Python unit loop over mp2g_chain_run instead of the
    native scheduler (the default, which the other multi-rank tests run: there the block roots leave the forests' device pools as host
    proofs and the separator rows above them are proved through the Python unit loop over the same chains); the run asserts the joined
    root's digest (the whole table's) and min"""
    line = _bench(["--gpus", "2", "--rows", "8", "--steps", "1", "--warmup", "1", "--workers", "2", "--table-batch", "8", "--subtree", "4", "--no-leaves-leg",
                   "--no-cpu-baseline", "--python-build"], env={"MP2G_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["config"]["host_orchestration"]["scheduler"].startswith("python")
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 5 * 17) < 1e-6 and line["verified"] >= 2 * 13


def test_forest_refuses_what_it_cannot_prove(ctx, mp2, params):
    """mp2g_forest_* error behaviour (csrc/forest.hip): a node registered twice, a unit naming an unknown node, a node whose child was
    never proved, a pool too small for a batch, a proof asked for before it exists -- each returns the library's error with a message
    and leaves the forest usable; released roots give their slots back"""
    nb = T.NativeTableBuild(params, [FW.GpuProver(ctx, capacity=4)], batch=4, subtree_size=4, group_rows=4, pool_slots=6)
    n = 2
    table = T.SyntheticTable(n, 4, seed=0xC0FFEE04, block=4)
    root, nodes, spans = T.balanced_bst(n)
    wit = T.TableWitness(ctx, table, spans)
    F = mp2.Forest([nb.provers[0].ctx], nb.desc, nb.chains, max(nb.pw_cells, nb.pw_rows), 6)
    nb.forest = F
    nb.register(table, wit, root, nodes, keep={root})
    with pytest.raises(mp2.Mp2gError, match="registered twice"):
        nb.register(table, wit, root, nodes, keep={root})
    with pytest.raises(mp2.Mp2gError, match="unknown node"):
        F.prove([[12345]])
    with pytest.raises(mp2.Mp2gError, match="not proved"):
        F.prove([[nb.cell_id(0, 2)]])  # the cells-tree full node of row 0 before its leaves
    with pytest.raises(mp2.Mp2gError, match="not proved"):
        F.proof_words(root)
    # 2 rows x 4 cells-tree nodes: the four leaves of both rows fill 4 of 6 slots, the two full nodes need 2 more while the leaves are
    # still alive, then slots come back; the whole block passes through 6 slots only because children are released as parents are proved
    F.prove([[nb.cell_id(k, c) for k in range(n) for c in range(1, 5)] + list(range(n))])
    assert F.proved == 5 * n
    words = F.proof_words(root)
    pis = words[:T.ROWS_IO + 4]
    assert np.array_equal(pis[:T.ROWS_IO], T.expected_root_public_inputs(ctx, table, wit, root, nodes, spans))
    load = mp2.load()
    assert load.mp2g_forest_free_slots(F.h) == 6 - 2  # the kept root and its kept cells root
    F.release(root)
    assert load.mp2g_forest_free_slots(F.h) == 6 - 1
    with pytest.raises(mp2.Mp2gError, match="not proved"):
        F.proof_words(root)
    # a pool that cannot hold one batch
    small = mp2.Forest([nb.provers[0].ctx], nb.desc, nb.chains, max(nb.pw_cells, nb.pw_rows), 3)
    nb.forest = small
    nb.register(table, wit, root, nodes)
    with pytest.raises(mp2.Mp2gError, match="pool is exhausted"):
        small.prove([[nb.cell_id(k, c) for k in range(n) for c in range(1, 5)] + list(range(n))])
    small.free()
    nb.forest = F
    nb.free()
    nb.provers[0].free()
