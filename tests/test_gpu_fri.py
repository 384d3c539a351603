"""HIP challenger / FRI / batched PCS prover vs the CPU oracle, bit for bit, and accepted by the
oracle's restatement of plonky2's FRI verifier (FRI kernels of 2a)."""
import ctypes

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def to_mp2(mp2, ofp):
    fp = mp2.FriParams()
    ctypes.memmove(ctypes.byref(fp), ctypes.byref(ofp), ctypes.sizeof(fp))
    return fp


@pytest.mark.parametrize("variant", [0, 1])
def test_challenger_matches_oracle(ctx, mp2, variant):
    count = 3
    ch = mp2.Challenger(ctx, variant, count)
    och = [O.lib() and None] * count
    states = []
    for t in range(count):
        c = (ctypes.c_uint64 * 32)()  # opaque orc_challenger (12+8+8 u64 + 3 u32)
        O.lib().orc_ch_init(c, variant)
        states.append(c)
    O.lib().orc_ch_get.restype = ctypes.c_uint64
    rng = np.random.default_rng(5)
    for n_obs, n_get in [(4, 0), (4, 2), (64, 4), (3, 1), (0, 9), (17, 2), (8, 8), (1, 1)]:
        elems = O.rand_field((count, n_obs), int(rng.integers(1 << 30))) if n_obs else np.zeros((count, 0), dtype=np.uint64)
        if n_obs:
            ch.observe_elements(elems)
        got = ch.get_n_challenges(n_get) if n_get else np.zeros((count, 0), dtype=np.uint64)
        for t in range(count):
            if n_obs:
                e = O.arr(elems[t])
                O.lib().orc_ch_observe(states[t], O.p(e), O.sz(n_obs))
            want = [O.lib().orc_ch_get(states[t]) for _ in range(n_get)]
            assert [int(x) for x in got[t]] == want


@pytest.mark.parametrize("log_m,ab", [(4, 4), (7, 4), (11, 4), (15, 4), (6, 1), (6, 2), (9, 3)])
def test_fri_fold_matches_coefficient_fold(ctx, mp2, log_m, ab):
    m = 1 << log_m
    coeffs = O.rand_field((m, 2), 5 + log_m)
    coeffs[m // 8:] = 0
    shift = O.MULT_GEN
    vals = np.stack([O.fft(coeffs[:, c].copy(), coset_shift=shift) for c in range(2)], axis=1)
    vb = O.arr(vals[O.bitrev_perm(m)])
    beta = O.rand_field(2, 9)
    want = np.zeros((m >> ab, 2), dtype=np.uint64)
    O.lib().orc_fri_fold_values(O.p(vb), log_m, ab, O.p(beta), ctypes.c_uint64(shift), O.p(want))
    got = mp2.fri_fold(ctx, vb, ab, beta, shift)
    # oracle returns natural order; the HIP path keeps leaf (bit-reversed) order
    assert np.array_equal(got, want[O.bitrev_perm(m >> ab)])


@pytest.mark.parametrize("variant", [0, 1])
def test_fri_pow_smallest_witness(ctx, mp2, variant):
    state = O.rand_field(12, 77)
    for pos, bits in [(0, 8), (3, 10), (7, 12), (5, 0)]:
        w = mp2.fri_pow(ctx, state, pos, bits, variant)
        def ok(cand):
            s = state.copy()
            s[pos] = cand
            return bits == 0 or int(O.perm(s, variant)[7]) >> (64 - bits) == 0
        assert ok(w)
        assert not any(ok(c) for c in range(max(0, w - 300), w))


@pytest.mark.parametrize("log_n,variant,ws", [(3, 0, (3, 4, 2, 2)), (6, 0, (5, 9, 4, 3)), (8, 1, (5, 9, 4, 3)),
                                              (10, 0, (7, 13, 4, 4)), (12, 0, (84, 135, 20, 16))])
def test_pcs_prove_bit_exact_and_verifies(ctx, mp2, log_n, variant, ws):
    full = log_n == 12
    ofp = O.standard_params(log_n, ws, variant=variant, pow_bits=16 if full else 8, num_queries=28 if full else 6)
    fp = to_mp2(mp2, ofp)
    n = 1 << log_n
    vals = [O.rand_field((w, n), 0xC0FFEE01 + i) for i, w in enumerate(ws)]
    cd, ph = O.rand_field(4, 1), O.rand_field(4, 2)
    caps, openings, proof = mp2.pcs_prove(ctx, fp, vals, cd, ph)
    assert O.pcs_verify(ofp, cd, ph, caps, openings, proof) == 0
    ocaps, oopen, oproof = O.pcs_prove(ofp, vals, cd, ph)
    assert np.array_equal(caps, ocaps)
    assert np.array_equal(openings, oopen)
    assert np.array_equal(proof, oproof)


def test_batched_prover_matches_single(ctx, mp2):
    log_n, ws, B = 7, (5, 9, 4, 3), 5
    ofp = O.standard_params(log_n, ws, pow_bits=6, num_queries=4)
    fp = to_mp2(mp2, ofp)
    n = 1 << log_n
    pre = O.rand_field((ws[0], n), 1)
    per = [[O.rand_field((w, n), 100 * b + i) for i, w in enumerate(ws[1:])] for b in range(B)]
    cd = O.rand_field(4, 3)
    ph = O.rand_field((B, 4), 4)
    pr = mp2.BatchedProver(ctx, fp, B)
    pr.set_preprocessed(ctx.to_device(pre))
    d_vals = [ctx.to_device(np.stack([per[b][i] for b in range(B)])) for i in range(len(ws) - 1)]
    pr.prove(d_vals, ctx.to_device(cd), ctx.to_device(ph))
    caps, openings, proofs = pr.results()
    for b in range(B):
        oc, oo, op = O.pcs_prove(ofp, [pre] + per[b], cd, ph[b])
        assert np.array_equal(caps[b], oc) and np.array_equal(openings[b], oo) and np.array_equal(proofs[b], op)
        assert O.pcs_verify(ofp, cd, ph[b], caps[b], openings[b], proofs[b]) == 0
    # the prover object is reusable: a second call with the same inputs gives the same proofs
    pr.prove(d_vals, ctx.to_device(cd), ctx.to_device(ph))
    assert np.array_equal(pr.results()[2], proofs)


def test_prover_param_errors(ctx, mp2):
    fp = mp2.standard_recursion_params(6, (3, 4, 2, 2))
    fp.zs_count = 9
    with pytest.raises(mp2.Mp2gError):
        mp2.BatchedProver(ctx, fp, 1)


def test_granular_prove_matches_fused(ctx, mp2):
    """The step-by-step path a Rust host would drive (commit, challenger, Z, commit, ..., openings,
    fri_prove) produces the same proof as the fused pipeline and the oracle."""
    log_n, num_routed = 7, 16
    ws = (2 + num_routed, num_routed + 5, 2 * (num_routed // 8), 4)
    ofp = O.standard_params(log_n, ws, pow_bits=6, num_queries=4)
    fp = to_mp2(mp2, ofp)
    n = 1 << log_n
    pre, wires, quot = O.rand_field((ws[0], n), 1), O.rand_field((ws[1], n), 2), O.rand_field((ws[3], n), 3)
    cd, ph = O.rand_field(4, 4), O.rand_field(4, 5)
    ch = mp2.Challenger(ctx)
    ch.observe_elements(cd); ch.observe_elements(ph)
    b0 = mp2.PolynomialBatch.from_values(ctx, pre)
    b1 = mp2.PolynomialBatch.from_values(ctx, wires)
    ch.observe_elements(b1.cap.reshape(-1))
    bg = ch.get_n_challenges(4)[0]
    zs = mp2.partial_products_and_zs(ctx, wires, pre[ws[0] - num_routed:], bg[:2], bg[2:], 8)
    b2 = mp2.PolynomialBatch.from_values(ctx, zs)
    ch.observe_elements(b2.cap.reshape(-1))
    ch.get_n_challenges(2)  # alphas
    b3 = mp2.PolynomialBatch.from_values(ctx, quot)  # a host would compute the quotient here
    ch.observe_elements(b3.cap.reshape(-1))
    zeta = ch.get_n_challenges(2)[0]
    g = pow(7277203076849721926, 1 << (32 - log_n), O.P)
    gz = [int(zeta[0]) * g % O.P, int(zeta[1]) * g % O.P]
    openings = np.concatenate([b.eval_ext(zeta) for b in (b0, b1, b2, b3)] + [b2.eval_ext(gz)[:2]])
    ch.observe_elements(openings.reshape(-1))
    proof = mp2.fri_prove(ctx, fp, [b0, b1, b2, b3], zeta, ch)
    caps = np.stack([b.cap.reshape(-1) for b in (b0, b1, b2, b3)])
    oc, oo, op = O.pcs_prove(ofp, [pre, wires, zs, quot], cd, ph, num_routed=num_routed, degree=8)
    assert np.array_equal(caps, oc) and np.array_equal(openings, oo) and np.array_equal(proof, op)
    assert O.pcs_verify(ofp, cd, ph, caps, openings, proof) == 0
