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
