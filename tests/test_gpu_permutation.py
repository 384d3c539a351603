"""Permutation argument (Z and partial products) on the GPU vs the oracle, stand-alone and inside
the batched prover (what prove() does between the wires and the Z commitment)."""
import ctypes

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
P = O.P


def identity_sigmas(log_n, num_routed):
    n = 1 << log_n
    w = pow(7277203076849721926, 1 << (32 - log_n), P)
    xs = [pow(w, i, P) for i in range(n)]
    return np.array([[pow(O.MULT_GEN, j, P) * x % P for x in xs] for j in range(num_routed)], dtype=np.uint64)


@pytest.mark.parametrize("log_n,num_routed,degree,nc", [(3, 8, 8, 1), (5, 16, 8, 2), (10, 80, 8, 2), (12, 80, 8, 2), (6, 12, 4, 2), (13, 16, 8, 1)])
def test_partial_products_match_oracle(ctx, mp2, log_n, num_routed, degree, nc):
    n = 1 << log_n
    wires = O.rand_field((num_routed + 3, n), 100 + log_n)
    sigmas = O.rand_field((num_routed, n), 200 + log_n)
    betas, gammas = O.rand_field(nc, 1), O.rand_field(nc, 2)
    got = mp2.partial_products_and_zs(ctx, wires, sigmas, betas, gammas, degree)
    want = O.partial_products_and_zs(wires, sigmas, betas, gammas, degree)
    assert np.array_equal(got, want)


def test_valid_permutation_wraps_to_one(ctx, mp2):
    """For a satisfied copy constraint Z(g^n) = Z(1) = 1: the last row's running product returns to 1."""
    log_n, R, deg = 6, 16, 8
    n = 1 << log_n
    sig = identity_sigmas(log_n, R)
    wires = O.rand_field((R, n), 5)
    # wire (3, 5) is copy-constrained to wire (7, 9): equal values, swapped sigmas
    wires[3, 5] = wires[7, 9]
    sig[3, 5], sig[7, 9] = sig[7, 9], sig[3, 5]
    betas, gammas = O.rand_field(2, 8), O.rand_field(2, 9)
    out = mp2.partial_products_and_zs(ctx, wires, sig, betas, gammas, deg)
    assert (out[:2, 0] == 1).all() and (out != 1).any()
    # Z(g^n): last Z times the product of the last row's chunk quotients
    for c in range(2):
        b, g = int(betas[c]), int(gammas[c])
        w = pow(7277203076849721926, 1 << (32 - log_n), P)
        x = pow(w, n - 1, P)
        q = 1
        for j in range(R):
            num = (int(wires[j, n - 1]) + b * pow(O.MULT_GEN, j, P) * x + g) % P
            den = (int(wires[j, n - 1]) + b * int(sig[j, n - 1]) + g) % P
            q = q * num * pow(den, P - 2, P) % P
        assert int(out[c, n - 1]) * q % P == 1
    # an unsatisfied constraint does not wrap
    wires[3, 5] ^= np.uint64(1)
    out = mp2.partial_products_and_zs(ctx, wires, sig, betas, gammas, deg)
    assert int(out[0, n - 1]) != 0


def test_prover_with_device_permutation(ctx, mp2):
    log_n, num_routed, B = 7, 16, 3
    ws = (2 + num_routed, num_routed + 5, 2 * (num_routed // 8), 4)
    ofp = O.standard_params(log_n, ws, pow_bits=6, num_queries=4)
    fp = mp2.FriParams()
    ctypes.memmove(ctypes.byref(fp), ctypes.byref(ofp), ctypes.sizeof(fp))
    n = 1 << log_n
    pre = O.rand_field((ws[0], n), 1)
    wires = [O.rand_field((ws[1], n), 10 + b) for b in range(B)]
    quot = [O.rand_field((ws[3], n), 20 + b) for b in range(B)]
    cd, ph = O.rand_field(4, 3), O.rand_field((B, 4), 4)
    pr = mp2.BatchedProver(ctx, fp, B)
    pr.set_preprocessed(ctx.to_device(pre))
    pr.enable_permutation(num_routed, 8)
    pr.prove([ctx.to_device(np.stack(wires)), None, ctx.to_device(np.stack(quot))], ctx.to_device(cd), ctx.to_device(ph))
    caps, openings, proofs = pr.results()
    dummy = np.zeros((ws[2], n), dtype=np.uint64)
    for b in range(B):
        oc, oo, op = O.pcs_prove(ofp, [pre, wires[b], dummy, quot[b]], cd, ph[b], num_routed=num_routed, degree=8)
        assert np.array_equal(caps[b], oc) and np.array_equal(openings[b], oo) and np.array_equal(proofs[b], op)
        assert O.pcs_verify(ofp, cd, ph[b], caps[b], openings[b], proofs[b]) == 0
    # shape errors
    with pytest.raises(mp2.Mp2gError):
        pr.enable_permutation(num_routed, 5)


def test_complete_proof_of_copy_constraint_circuit(ctx, mp2):
    """Wires in, proof out: Z / partial products, quotient chunks (gate-independent vanishing terms),
    commitments, openings and FRI all on the device. Bit-exact vs the oracle, FRI verifier accepts,
    and the PLONK identity vanishing(zeta) = Z_H(zeta) t(zeta) holds on the opened values."""
    log_n, R, B = 6, 16, 2
    ws = (2 + R, R + 3, 2 * (R // 8), 16)
    ofp = O.standard_params(log_n, ws, pow_bits=4, num_queries=3)
    fp = mp2.FriParams()
    ctypes.memmove(ctypes.byref(fp), ctypes.byref(ofp), ctypes.sizeof(fp))
    n = 1 << log_n
    sig, wires0 = O.copy_constraint_circuit(log_n, R, ws[1], 20, 7)
    # second witness for the same circuit: constrained cells must stay equal -> reuse and re-randomise free cells only
    pre = np.concatenate([O.rand_field((2, n), 1), sig])
    wires = [wires0, wires0.copy()]
    wires[1][R:] = O.rand_field((ws[1] - R, n), 77)  # unrouted advice columns are free
    cd, ph = O.rand_field(4, 2), O.rand_field((B, 4), 3)
    pr = mp2.BatchedProver(ctx, fp, B)
    pr.set_preprocessed(ctx.to_device(pre))
    pr.enable_permutation(R, 8)
    pr.enable_quotient()
    pr.prove([ctx.to_device(np.stack(wires)), None, None], ctx.to_device(cd), ctx.to_device(ph))
    caps, openings, proofs = pr.results()
    d2, d3 = np.zeros((ws[2], n), dtype=np.uint64), np.zeros((ws[3], n), dtype=np.uint64)
    for b in range(B):
        oc, oo, op, bgao = O.pcs_prove(ofp, [pre, wires[b], d2, d3], cd, ph[b], num_routed=R, degree=8, quotient=True, want_challenges=True)
        assert np.array_equal(caps[b], oc) and np.array_equal(openings[b], oo) and np.array_equal(proofs[b], op)
        assert O.pcs_verify(ofp, cd, ph[b], caps[b], openings[b], proofs[b]) == 0
        assert O.plonk_identity_check(ofp, R, 8, openings[b], bgao) == 0
    # an unsatisfied witness still yields a FRI-valid opening proof but fails the PLONK identity
    bad = O.rand_field(wires0.shape, 5)
    pr.prove([ctx.to_device(np.stack([bad, bad])), None, None], ctx.to_device(cd), ctx.to_device(ph))
    caps, openings, proofs = pr.results()
    _, _, _, bgao = O.pcs_prove(ofp, [pre, bad, d2, d3], cd, ph[0], num_routed=R, degree=8, quotient=True, want_challenges=True)
    assert O.plonk_identity_check(ofp, R, 8, openings[0], bgao) != 0


def test_complete_proof_standard_shape(ctx, mp2):
    """Same at standard_recursion_config shape: 80 routed of 135 wires, 2^12 rows."""
    log_n, R = 12, 80
    ws = (84, 135, 20, 16)
    ofp = O.standard_params(log_n, ws)
    fp = mp2.standard_recursion_params(log_n, ws)
    n = 1 << log_n
    sig, wires = O.copy_constraint_circuit(log_n, R, ws[1], 5000, 11)
    pre = np.concatenate([O.rand_field((4, n), 1), sig])
    cd, ph = O.rand_field(4, 2), O.rand_field((1, 4), 3)
    pr = mp2.BatchedProver(ctx, fp, 1)
    pr.set_preprocessed(ctx.to_device(pre))
    pr.enable_permutation(R, 8)
    pr.enable_quotient()
    pr.prove([ctx.to_device(wires[None]), None, None], ctx.to_device(cd), ctx.to_device(ph))
    caps, openings, proofs = pr.results()
    d2, d3 = np.zeros((ws[2], n), dtype=np.uint64), np.zeros((ws[3], n), dtype=np.uint64)
    oc, oo, op, bgao = O.pcs_prove(ofp, [pre, wires, d2, d3], cd, ph[0], num_routed=R, degree=8, quotient=True, want_challenges=True)
    assert np.array_equal(caps[0], oc) and np.array_equal(openings[0], oo) and np.array_equal(proofs[0], op)
    assert O.pcs_verify(ofp, cd, ph[0], caps[0], openings[0], proofs[0]) == 0
    assert O.plonk_identity_check(ofp, R, 8, openings[0], bgao) == 0
