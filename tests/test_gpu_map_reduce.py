"""Map-reduce shape of recursion-framework/tests/integration.rs:138-261: 8 leaf proofs, a 2-to-1
reduction tree proved level by level in batches; every proof is accepted by the oracle's FRI
verifier and parents depend on their children's commitments."""
import ctypes

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def test_eight_leaves_two_to_one(ctx, mp2):
    log_n, ws, n_leaves = 6, (5, 9, 4, 3), 8
    ofp = O.standard_params(log_n, ws, pow_bits=6, num_queries=4)
    fp = mp2.FriParams()
    ctypes.memmove(ctypes.byref(fp), ctypes.byref(ofp), ctypes.sizeof(fp))
    n = 1 << log_n
    pre = O.rand_field((ws[0], n), 1)
    leaf_vals = [O.rand_field((n_leaves, w, n), 10 + i) for i, w in enumerate(ws[1:])]
    cd = O.rand_field(4, 3)
    levels = mp2.prove_aggregation_tree(ctx, fp, pre, leaf_vals, cd)
    assert [lv[0].shape[0] for lv in levels] == [8, 4, 2, 1]
    for li, (pi, caps, openings, proofs) in enumerate(levels):
        for b in range(pi.shape[0]):
            assert O.pcs_verify(ofp, cd, pi[b], caps[b], openings[b], proofs[b]) == 0
        if li:
            prev_caps = levels[li - 1][1]
            for b in range(pi.shape[0]):
                want = O.hash_n_to_m_no_pad(np.concatenate([prev_caps[2 * b, 1], prev_caps[2 * b + 1, 1]]), 4)
                assert np.array_equal(pi[b], want)
    # a parent proof does not verify under a sibling's public inputs
    pi1, caps1, op1, pr1 = levels[1]
    assert O.pcs_verify(ofp, cd, pi1[1], caps1[0], op1[0], pr1[0]) != 0


def test_two_ranks_prove_an_update_tree():
    """N > 1 end to end on the GPU box: two ranks share the GPU, split a batched work plan over a 14-node update
    tree (sharding.run_workplan), prove every node for real (gate-level circuit, witness check on) and exchange
    only root results; the root fingerprint equals the single-process one and every node is proved once."""
    import importlib.util
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(here, "_dist_gpu_worker.py")
    spec = importlib.util.spec_from_file_location("_dist_gpu_worker", worker)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want_root, n_proved, n_nodes = mod.run(None)
    assert n_proved == n_nodes == 14
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", worker]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"root={want_root} proved=14" in r.stdout, r.stdout[-2000:]
