"""The BASELINE.json shapes under -m gpu: the bench's own step (configs[1]/[3]: B = 64 proofs per prover at 2^13 /
2^12 rows, four concurrent contexts, sampled proofs bit-exact against the oracle) and the 2^20-row table digest of
configs[3]. configs[2] (1024-leaf aggregation) is in test_gpu_map_reduce.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_step_is_self_verifying():
    """one bench step at the headline shape: 128 leaf proofs = 2 x 64 base (2^13 rows, 19 gates) + 2 x 64 wrap
    (2^12 rows, 13 gates) on 4 contexts; first and last proof of every prover equal the oracle's and verify"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "leaves", "--steps", "1", "--warmup", "1", "--batch", "128", "--streams", "4",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["verified"] >= 8
    assert line["config"]["batch_per_rank"] == 128 and line["config"]["streams"] == 4
    assert line["config"]["gates"] == {"base": 19, "wrap": 13}
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


def test_bench_refuses_a_wrong_proof(monkeypatch):
    """the self-check is live: a corrupted GPU proof makes check_against_oracle fail"""
    sys.path.insert(0, ROOT)
    import importlib
    import bench
    import circuits as C
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    FW = importlib.import_module("mapreduce-plonky2_amd.framework")
    ctx = mp2.Context(0)
    try:
        ckt = C.build(6, C.VERIFIER_KINDS, 11)
        cp = FW.CircuitProver(ctx, ckt, 2, bind_public_inputs=True)
        d_w = FW.tile_witness(ctx, ckt, 2, 5)
        ph = C.rand_field((2, 4), 6)
        cp.prove(d_w, ctx.to_device(ph))
        caps, openings, proofs = cp.results()
        ofp = O.standard_params(6, (int(ckt.pre.shape[0]), 135, 20, 16))
        def leaf(b, proof):
            return [(f"proof {b}", ckt, ofp, cp.circuit_digest, lambda: FW.witness_of(ckt, 5, b, ph[b]), ph[b], caps[b], openings[b], proof, True)]
        n, _ = bench.check_against_oracle([leaf(0, proofs[0]), leaf(1, proofs[1])], 0.0, False)
        assert n == 2
        bad = proofs[1].copy()
        bad[-1] ^= np.uint64(1)  # the PoW witness
        with pytest.raises(SystemExit):
            bench.check_against_oracle([leaf(1, bad)], 0.0, False)
    finally:
        ctx.close()


def test_table_digest_2p20_rows_baseline_config3(ctx, mp2):
    """compute_table_row_digest (mp2-v1/src/values_extraction/mod.rs:527-571) over the 2^20 rows x 4 value columns of
    configs[3]: the whole-table digest equals the curve sum of 16 shard digests (the N-rank split of sharding.py:
    block partition, one point per shard) and, on 4 random 256-row windows, the oracle's digest."""
    rows, n_cols, n_unique, shards = 1 << 20, 4, 1, 16
    rng = np.random.default_rng(0xC0FFEE04)
    ids = O.rand_field(n_cols, 0xC0FFEE04)
    values = rng.integers(0, 1 << 32, size=(rows, n_cols, 8), dtype=np.uint32)
    unique = rng.integers(0, 1 << 32, size=(rows, n_unique, 8), dtype=np.uint32)
    d_ids, d_v, d_u = ctx.to_device(ids), ctx.to_device(values), ctx.to_device(unique)
    whole = mp2.compute_table_row_digest_dev(ctx, d_ids, n_cols, d_v, d_u, n_unique, rows)
    per = rows // shards
    parts = [mp2.compute_table_row_digest(ctx, ids, values[s * per:(s + 1) * per], unique[s * per:(s + 1) * per])[0] for s in range(shards)]
    assert np.array_equal(mp2.curve_sum(ctx, np.stack(parts)), whole)
    L = O.lib()
    for start in rng.integers(0, rows - 256, size=4):
        v, u = O.arr(values[start:start + 256], np.uint32), O.arr(unique[start:start + 256], np.uint32)
        w, wei = np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
        L.orc_row_digest_batch(0, O.p(O.arr(ids)), O.sz(n_cols), O.p(v), O.p(u), O.sz(n_unique), O.sz(256), O.p(w), O.p(wei))
        got, got_wei = mp2.compute_table_row_digest(ctx, ids, v, u)
        assert np.array_equal(got, w) and np.array_equal(got_wei, wei)
