"""The oracle's FRI prover is accepted by the oracle's FRI verifier (restated from plonky2
fri/verifier.rs), and tampering is rejected. CPU only."""
import numpy as np
import pytest

import oracle as O


@pytest.mark.parametrize("log_n,variant", [(6, 0), (8, 0), (10, 1)])
def test_pcs_prove_verify(log_n, variant):
    ws = (5, 9, 4, 3)
    fp = O.standard_params(log_n, ws, variant=variant, pow_bits=8, num_queries=6)
    n = 1 << log_n
    vals = [O.rand_field((w, n), 100 + i) for i, w in enumerate(ws)]
    cd, ph = O.rand_field(4, 1), O.rand_field(4, 2)
    caps, openings, proof = O.pcs_prove(fp, vals, cd, ph)
    assert O.pcs_verify(fp, cd, ph, caps, openings, proof) == 0
    bad = proof.copy()
    bad[-3] ^= np.uint64(1)  # final polynomial coefficient
    assert O.pcs_verify(fp, cd, ph, caps, openings, bad) != 0
    bad_open = openings.copy()
    bad_open[0, 0] ^= np.uint64(1)
    assert O.pcs_verify(fp, cd, ph, caps, bad_open, proof) != 0


def test_reduction_strategy_matches_survey():
    # k=12:[4,4], 13:[4,4], 14:[4,4,4], 15:[4,4,4]
    for k, n_layers in ((12, 2), (13, 2), (14, 3), (15, 3)):
        fp = O.standard_params(k)
        assert fp.n_layers == n_layers and all(fp.arity_bits[i] == 4 for i in range(n_layers))


def test_value_fold_equals_coefficient_fold():
    import ctypes
    log_m, ab = 7, 4
    m = 1 << log_m
    coeffs = O.rand_field((m, 2), 5)
    coeffs[m // 8:] = 0
    shift = O.MULT_GEN
    vals = np.stack([O.fft(coeffs[:, c].copy(), coset_shift=shift) for c in range(2)], axis=1)
    br = O.bitrev_perm(m)
    vb = O.arr(vals[br])
    beta = O.rand_field(2, 9)
    out = np.zeros((m >> ab, 2), dtype=np.uint64)
    O.lib().orc_fri_fold_values(O.p(vb), log_m, ab, O.p(beta), ctypes.c_uint64(shift), O.p(out))
    assert out.any()
