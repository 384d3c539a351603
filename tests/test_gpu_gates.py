"""Gate constraint evaluators on the GPU vs the oracle: per-constraint values at
arbitrary points, the witness check on H, and the complete prove() of gate-level circuits, bit-exact
against the oracle's proof, accepted by the oracle's verifier (FRI + PLONK identity with gate terms)."""
import ctypes

import numpy as np
import pytest

import circuits as C
import oracle as O

pytestmark = pytest.mark.gpu
P = O.P


def gpu_gates(mp2, ckt):
    return [mp2.Gate(g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end) for g in ckt.gates]


def params(mp2, ckt, log_n, **kw):
    ws = (ckt.num_constants + C.NUM_ROUTED, C.NUM_WIRES, 20, 16)
    ofp = O.standard_params(log_n, ws, **kw)
    fp = mp2.FriParams()
    ctypes.memmove(ctypes.byref(fp), ctypes.byref(ofp), ctypes.sizeof(fp))
    return ofp, fp


@pytest.mark.parametrize("kinds", [[k] for k in C.ALL_KINDS if k[0] != C.NOOP] + [C.ALL_KINDS])
def test_constraints_match_oracle_at_random_points(ctx, mp2, kinds):
    ckt = C.build(5, [(C.NOOP, 0, 0, 0)] + [k for k in kinds if k[0] != C.NOOP], 21)
    npts = 300
    consts = O.rand_field((ckt.num_constants, npts), 1)
    # selectors hit real gate indices on some points so that filters are exercised at zero and non-zero values
    consts[:ckt.num_selectors, :100] = np.arange(100, dtype=np.uint64)[None, :] % np.uint64(len(ckt.gates) + 1)
    wires = O.rand_field((C.NUM_WIRES, npts), 2)
    got = mp2.eval_gate_constraints(ctx, gpu_gates(mp2, ckt), ckt.num_selectors, consts, wires, ckt.pi_hash)
    want = C.eval_on_points(ckt, consts, wires)
    assert np.array_equal(got, want)
    assert got.any()


def test_witness_check_on_h(ctx, mp2):
    ckt = C.build(6, C.ALL_KINDS, 4)
    gates = gpu_gates(mp2, ckt)
    out = mp2.eval_gate_constraints(ctx, gates, ckt.num_selectors, ckt.pre[:ckt.num_constants], ckt.wires, ckt.pi_hash)
    assert not out.any()
    bad = ckt.wires.copy()
    row = ckt.instances.index(next(i for i, g in enumerate(ckt.gates) if g.kind == C.POSEIDON2))
    bad[70, row] ^= np.uint64(1)  # one partial-round S-box input
    out = mp2.eval_gate_constraints(ctx, gates, ckt.num_selectors, ckt.pre[:ckt.num_constants], bad, ckt.pi_hash)
    assert out[:, row].any() and not np.delete(out, row, axis=1).any()


def test_gate_descriptor_queries(mp2):
    for k in C.ALL_KINDS:
        og = C.Gate(*k, 0, 0, 0)
        g = mp2.Gate(*k, 0, 0, 0)
        assert g.num_constraints == C.gate_num_constraints(og) and g.degree == C.gate_degree(og)


@pytest.mark.parametrize("kinds,log_n,B", [([(C.NOOP, 0, 0, 0), (C.CONSTANT, 2, 0, 0), (C.PUBLIC_INPUT, 0, 0, 0), (C.ARITHMETIC, 20, 0, 0)], 5, 2),
                                            (C.ALL_KINDS, 6, 2), ([(C.NOOP, 0, 0, 0), (C.POSEIDON2, 0, 0, 0), (C.ARITHMETIC, 20, 0, 0)], 8, 1)])
def test_complete_proof_with_gates(ctx, mp2, kinds, log_n, B):
    ckt = C.build(log_n, kinds, 5)
    ofp, fp = params(mp2, ckt, log_n, pow_bits=4, num_queries=3)
    cd = O.rand_field(4, 9)
    # the batch proves the same witness under different public-input hashes only when no PublicInput gate binds
    # them; here every proof of the batch uses the circuit's hash
    ph = np.stack([ckt.pi_hash] * B)
    pr = mp2.BatchedProver(ctx, fp, B)
    pr.set_preprocessed(ctx.to_device(ckt.pre))
    pr.enable_permutation(C.NUM_ROUTED, 8)
    pr.enable_quotient()
    pr.set_gates(gpu_gates(mp2, ckt), ckt.num_selectors)
    pr.prove([ctx.to_device(np.stack([ckt.wires] * B)), None, None], ctx.to_device(cd), ctx.to_device(ph))
    caps, openings, proofs = pr.results()
    oc, oo, op, bgao = C.prove(ckt, ofp, cd)
    for b in range(B):
        assert np.array_equal(caps[b], oc) and np.array_equal(openings[b], oo) and np.array_equal(proofs[b], op)
    assert O.pcs_verify(ofp, cd, ckt.pi_hash, caps[0], openings[0], proofs[0]) == 0
    assert C.identity_check(ckt, ofp, openings[0], bgao) == 0
    # a witness violating one gate: still bit-exact with the oracle, FRI opens, the identity fails
    gi = next(i for i, g in enumerate(ckt.gates) if g.kind == C.ARITHMETIC)
    row = ckt.instances.index(gi)
    ckt.wires[3, row] = (int(ckt.wires[3, row]) + 1) % P
    pr.prove([ctx.to_device(np.stack([ckt.wires] * B)), None, None], ctx.to_device(cd), ctx.to_device(ph))
    caps, openings, proofs = pr.results()
    oc, oo, op, bgao = C.prove(ckt, ofp, cd)
    assert np.array_equal(caps[0], oc) and np.array_equal(openings[0], oo) and np.array_equal(proofs[0], op)
    assert C.identity_check(ckt, ofp, openings[0], bgao) != 0
    # removing the table returns to the copy-constraint-only quotient
    pr.set_gates([], 1)
    pr.free()


def test_gate_table_validation(ctx, mp2):
    ckt = C.build(5, C.ALL_KINDS, 3)
    ofp, fp = params(mp2, ckt, 5, pow_bits=2, num_queries=2)
    pr = mp2.BatchedProver(ctx, fp, 1)
    pr.set_preprocessed(ctx.to_device(ckt.pre))
    with pytest.raises(mp2.Mp2gError):
        pr.set_gates(gpu_gates(mp2, ckt), ckt.num_selectors)  # needs enable_quotient first
    pr.enable_permutation(C.NUM_ROUTED, 8)
    pr.enable_quotient()
    good = gpu_gates(mp2, ckt)
    pr.set_gates(good, ckt.num_selectors)
    for mutate in (lambda g: setattr(g[0], "kind", 99), lambda g: setattr(g[1], "selector_index", 9),
                   lambda g: setattr(g[2], "group_end", g[2].group_start), lambda g: setattr(next(x for x in g if x.kind == mp2.GATE_ARITHMETIC), "p0", 40)):
        bad = gpu_gates(mp2, ckt)
        mutate(bad)
        with pytest.raises(mp2.Mp2gError):
            pr.set_gates(bad, ckt.num_selectors)
    with pytest.raises(mp2.Mp2gError):
        pr.set_gates(good, 0)
    big = [mp2.Gate(mp2.GATE_ARITHMETIC, 40, 0, 0, 0, 0, 1)]  # 160 routed wires > 135
    with pytest.raises(mp2.Mp2gError):
        pr.set_gates(big, 1)


def test_gate_parameter_fuzz(ctx, mp2):
    """random gate parameters (op counts, limb counts, bases, bit widths) at random points vs the oracle"""
    rng = np.random.default_rng(2024)
    for it in range(12):
        kinds = [(C.NOOP, 0, 0, 0), (C.CONSTANT, int(rng.integers(1, 3)), 0, 0), (C.ARITHMETIC, int(rng.integers(1, 21)), 0, 0),
                 (C.BASE_SUM, int(rng.integers(1, 64)), int(rng.integers(2, 6)), 0), (C.ARITHMETIC_EXT, int(rng.integers(1, 11)), 0, 0),
                 (C.MUL_EXT, int(rng.integers(1, 14)), 0, 0), (C.EXPONENTIATION, int(rng.integers(1, 67)), 0, 0),
                 (C.REDUCING, int(rng.integers(1, 44)), 0, 0), (C.REDUCING_EXT, int(rng.integers(1, 33)), 0, 0)]
        bits = int(rng.integers(1, 6))
        copies = int(rng.integers(1, max(2, min(78 // (2 + (1 << bits)), (135 - 80) // bits) + 1)))
        kinds.append((C.RANDOM_ACCESS, bits, copies, int(rng.integers(0, 3))))
        sb = int(rng.integers(2, 6))
        deg = int(rng.integers(2, min(1 << sb, 7) + 1))
        if 1 + 2 * (1 << sb) + 6 + 4 * (((1 << sb) - 2) // (deg - 1)) <= 135:
            kinds.append((C.COSET_INTERPOLATION, sb, deg, 0))
        kinds += [(C.U32_ARITHMETIC, int(rng.integers(1, 4)), 0, 0), (C.U32_RANGE_CHECK, int(rng.integers(1, 8)), 0, 0),
                  (C.U32_SUBTRACTION, int(rng.integers(1, 7)), 0, 0)]
        na = int(rng.integers(1, 9))
        kinds.append((C.U32_ADD_MANY, na, int(rng.integers(1, min(135 // (na + 21), 80 // (na + 3)) + 1)), 0))
        nch = int(rng.choice([4, 8, 16]))
        kinds.append((C.COMPARISON, nch * int(rng.integers(1, 3)), nch, 0))
        pick = [kinds[0]] + [kinds[i] for i in sorted(rng.choice(np.arange(1, len(kinds)), size=int(rng.integers(1, 6)), replace=False))]
        ckt = C.build(4, pick, 100 + it)
        out = C.eval_on_points(ckt, ckt.pre[:ckt.num_constants], ckt.wires)
        assert not out.any(), pick  # the witness generator and the oracle agree on every parameterisation
        npts = 200
        consts = O.rand_field((ckt.num_constants, npts), it)
        consts[:ckt.num_selectors, :64] = np.arange(64, dtype=np.uint64)[None, :] % np.uint64(len(ckt.gates) + 1)
        wires = O.rand_field((C.NUM_WIRES, npts), 50 + it)
        got = mp2.eval_gate_constraints(ctx, gpu_gates(mp2, ckt), ckt.num_selectors, consts, wires, ckt.pi_hash)
        assert np.array_equal(got, C.eval_on_points(ckt, consts, wires)), pick


def test_complete_proof_standard_shape_with_gates(ctx, mp2):
    """standard_recursion_config shape (2^12 rows, 135 wires, 80 routed, cap 16, 16-bit PoW, 28 queries) with
    every supported gate: bit-exact vs the oracle, FRI verifier accepts, PLONK identity holds; and the
    device-side witness check is clean on H at full size."""
    log_n = 12
    ckt = C.build(log_n, C.ALL_KINDS, 0xC0FFEE03)
    ofp, fp = params(mp2, ckt, log_n)
    gates = gpu_gates(mp2, ckt)
    assert not mp2.eval_gate_constraints(ctx, gates, ckt.num_selectors, ckt.pre[:ckt.num_constants], ckt.wires, ckt.pi_hash).any()
    cd = O.rand_field(4, 9)
    pr = mp2.BatchedProver(ctx, fp, 1)
    pr.set_preprocessed(ctx.to_device(ckt.pre))
    pr.enable_permutation(C.NUM_ROUTED, 8)
    pr.enable_quotient()
    pr.set_gates(gates, ckt.num_selectors)
    pr.prove([ctx.to_device(ckt.wires[None]), None, None], ctx.to_device(cd), ctx.to_device(ckt.pi_hash[None]))
    caps, openings, proofs = pr.results()
    oc, oo, op, bgao = C.prove(ckt, ofp, cd)
    assert np.array_equal(caps[0], oc) and np.array_equal(openings[0], oo) and np.array_equal(proofs[0], op)
    assert O.pcs_verify(ofp, cd, ckt.pi_hash, caps[0], openings[0], proofs[0]) == 0
    assert C.identity_check(ckt, ofp, openings[0], bgao) == 0
    # the proof survives the wire format
    blob = mp2.serialize_proof(fp, ckt.num_constants, caps[0], openings[0], proofs[0], ckt.pi_hash)
    c2, o2, p2, pi2 = mp2.deserialize_proof(fp, ckt.num_constants, blob, 4)
    assert np.array_equal(c2[1:], caps[0][1:]) and np.array_equal(o2, openings[0]) and np.array_equal(p2, proofs[0])
    pr.free()


def test_graph_replay_is_identical(ctx, mp2):
    """mp2g_prover_enable_graph: the captured launch sequence reproduces the plain one bit for bit, follows
    new inputs placed in the same buffers, and re-captures when the buffers change."""
    log_n, B = 6, 2
    ckt = C.build(log_n, C.ALL_KINDS, 8)
    ofp, fp = params(mp2, ckt, log_n, pow_bits=5, num_queries=3)
    pr = mp2.BatchedProver(ctx, fp, B)
    pr.set_preprocessed(ctx.to_device(ckt.pre))
    pr.enable_permutation(C.NUM_ROUTED, 8)
    pr.enable_quotient()
    pr.set_gates(gpu_gates(mp2, ckt), ckt.num_selectors)
    d_w, d_ph = ctx.to_device(np.stack([ckt.wires] * B)), ctx.to_device(np.stack([ckt.pi_hash] * B))
    cd1, cd2 = O.rand_field(4, 1), O.rand_field(4, 2)
    d_cd = ctx.to_device(cd1)
    pr.prove([d_w, None, None], d_cd, d_ph)
    ref1 = pr.results()
    pr.enable_graph(True)
    for _ in range(3):  # plain, capture, replay
        pr.prove([d_w, None, None], d_cd, d_ph)
        assert all(np.array_equal(a, b) for a, b in zip(ref1, pr.results()))
    d_cd.upload(cd2)  # same buffer, new transcript seed: the replay must follow the data
    pr.prove([d_w, None, None], d_cd, d_ph)
    got2 = pr.results()
    oc, oo, op, _ = C.prove(ckt, ofp, cd2)
    assert np.array_equal(got2[0][0], oc) and np.array_equal(got2[1][0], oo) and np.array_equal(got2[2][0], op)
    d_cd_other = ctx.to_device(cd1)  # different buffer: re-capture
    pr.prove([d_w, None, None], d_cd_other, d_ph)
    assert all(np.array_equal(a, b) for a, b in zip(ref1, pr.results()))
    pr.free()


def test_graph_survives_a_scratch_reallocation(ctx, mp2):
    """A captured graph references the context's shared NTT scratch (the quotient iNTT at log_n >= 10 runs two natural-order
    passes through it). A larger transform on the same context reallocates that buffer: the next prove must notice
    (generation counter) and re-capture instead of replaying into freed memory."""
    log_n, B = 10, 1
    kinds = [(C.NOOP, 0, 0, 0), (C.CONSTANT, 2, 0, 0), (C.PUBLIC_INPUT, 0, 0, 0), (C.ARITHMETIC, 20, 0, 0), (C.POSEIDON2, 0, 0, 0)]
    ckt = C.build(log_n, kinds, 12)
    ofp, fp = params(mp2, ckt, log_n, pow_bits=4, num_queries=3)
    pr = mp2.BatchedProver(ctx, fp, B)
    pr.set_preprocessed(ctx.to_device(ckt.pre))
    pr.enable_permutation(C.NUM_ROUTED, 8)
    pr.enable_quotient()
    pr.set_gates(gpu_gates(mp2, ckt), ckt.num_selectors)
    d_w, d_ph, d_cd = ctx.to_device(ckt.wires[None]), ctx.to_device(ckt.pi_hash[None]), ctx.to_device(O.rand_field(4, 1))
    pr.enable_graph(True)
    for _ in range(3):  # plain, capture, replay
        pr.prove([d_w, None, None], d_cd, d_ph)
    ref = pr.results()
    oc, oo, op, _ = C.prove(ckt, ofp, O.rand_field(4, 1))
    assert np.array_equal(ref[0][0], oc) and np.array_equal(ref[1][0], oo) and np.array_equal(ref[2][0], op)
    big = O.rand_field((64, 1 << 14), 3)  # a natural-order two-pass transform 64x larger than anything the prover ran: scratch grows
    want = O.fft(big)
    assert np.array_equal(ctx.ntt(big), want)
    for _ in range(2):
        pr.prove([d_w, None, None], d_cd, d_ph)
        got = pr.results()
        assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    pr.free()


def test_one_challenge_round(ctx, mp2):
    """num_challenges = 1 (zs_count 1): one beta, one gamma, one alpha are drawn (plonk/prover.rs get_n_challenges(num_challenges)),
    Z / partial products and the quotient use the same gamma, and the proof verifies"""
    log_n = 6
    ckt = C.build(log_n, C.ALL_KINDS, 17)
    ofp = O.standard_params(log_n, (int(ckt.pre.shape[0]), C.NUM_WIRES, 10, 8), zs_count=1, pow_bits=4, num_queries=3)
    fp = mp2.FriParams()
    ctypes.memmove(ctypes.byref(fp), ctypes.byref(ofp), ctypes.sizeof(fp))
    pr = mp2.BatchedProver(ctx, fp, 2)
    pr.set_preprocessed(ctx.to_device(ckt.pre))
    pr.enable_permutation(C.NUM_ROUTED, 8)
    pr.enable_quotient()
    pr.set_gates(gpu_gates(mp2, ckt), ckt.num_selectors)
    pr.enable_witness_check()
    cd = O.rand_field(4, 2)
    pr.prove([ctx.to_device(np.stack([ckt.wires] * 2)), None, None], ctx.to_device(cd), ctx.to_device(np.stack([ckt.pi_hash] * 2)))
    assert pr.witness_status().tolist() == [0, 0]
    caps, openings, proofs = pr.results()
    oc, oo, op, chal = C.prove(ckt, ofp, cd)
    for b in range(2):
        assert np.array_equal(caps[b], oc) and np.array_equal(openings[b], oo) and np.array_equal(proofs[b], op)
    assert C.verify(ckt, ofp, cd, ckt.pi_hash, caps[0], openings[0], proofs[0]) == 0
    assert C.identity_check(ckt, ofp, openings[0], chal) == 0
    pr.free()


def test_noop_only_table_and_distinct_witnesses(ctx, mp2):
    """(i) a gate table holding only NoopGate gives the copy-constraint-only quotient; (ii) the proofs of one
    batch are independent: different witnesses (free cells re-drawn) and public-input hashes per proof."""
    log_n, B = 6, 3
    kinds = [(C.NOOP, 0, 0, 0), (C.ARITHMETIC, 20, 0, 0), (C.POSEIDON2, 0, 0, 0), (C.BASE_SUM, 63, 2, 0)]  # no PublicInput gate
    ckt = C.build(log_n, kinds, 31)
    ofp, fp = params(mp2, ckt, log_n, pow_bits=4, num_queries=3)
    cd = O.rand_field(4, 9)
    noop_rows = [r for r, g in enumerate(ckt.instances) if ckt.gates[g].kind == C.NOOP]
    wires = []
    for b in range(B):
        w = ckt.wires.copy()
        w[C.NUM_ROUTED:, noop_rows] = O.rand_field((C.NUM_WIRES - C.NUM_ROUTED, len(noop_rows)), 500 + b)
        wires.append(w)
    ph = O.rand_field((B, 4), 77)
    pr = mp2.BatchedProver(ctx, fp, B)
    pr.set_preprocessed(ctx.to_device(ckt.pre))
    pr.enable_permutation(C.NUM_ROUTED, 8)
    pr.enable_quotient()
    pr.set_gates(gpu_gates(mp2, ckt), ckt.num_selectors)
    pr.prove([ctx.to_device(np.stack(wires)), None, None], ctx.to_device(cd), ctx.to_device(ph))
    caps, openings, proofs = pr.results()
    good = ckt.wires
    for b in range(B):
        ckt.wires, ckt.pi_hash = wires[b], ph[b]
        oc, oo, op, bgao = C.prove(ckt, ofp, cd)
        assert np.array_equal(caps[b], oc) and np.array_equal(openings[b], oo) and np.array_equal(proofs[b], op)
        assert O.pcs_verify(ofp, cd, ph[b], caps[b], openings[b], proofs[b]) == 0
        assert C.identity_check(ckt, ofp, openings[b], bgao) == 0
    assert not np.array_equal(proofs[0], proofs[1])
    ckt.wires = good
    # Noop-only table == no table
    sig, w0 = O.copy_constraint_circuit(log_n, C.NUM_ROUTED, C.NUM_WIRES, 40, 7)
    pre = np.concatenate([O.rand_field((ckt.num_constants, 1 << log_n), 1), sig])
    pr.set_preprocessed(ctx.to_device(pre))
    res = []
    for table in ([mp2.Gate(mp2.GATE_NOOP, 0, 0, 0, 0, 0, 1)], []):
        pr.set_gates(table, 1)
        pr.prove([ctx.to_device(np.stack([w0] * B)), None, None], ctx.to_device(cd), ctx.to_device(ph))
        res.append(pr.results())
    assert all(np.array_equal(a, b) for a, b in zip(*res))
    oc, oo, op = O.pcs_prove(ofp, [pre, w0, np.zeros((20, 1 << log_n), np.uint64), np.zeros((16, 1 << log_n), np.uint64)], cd, ph[0],
                             num_routed=C.NUM_ROUTED, degree=8, quotient=True)
    assert np.array_equal(res[0][2][0], op)
    pr.free()


def test_device_witness_check(ctx, mp2):
    """mp2g_prover_enable_witness_check: a satisfied batch is clean; a violated gate, a violated copy constraint
    and both are reported per proof, with an error naming the first offender (prove() panics in the reference)."""
    log_n, B = 6, 4
    ckt = C.build(log_n, C.ALL_KINDS, 12)
    ofp, fp = params(mp2, ckt, log_n, pow_bits=3, num_queries=2)
    n = 1 << log_n
    wN = pow(7277203076849721926, 1 << (32 - log_n), P)
    ident = lambda col, row: pow(O.MULT_GEN, col, P) * pow(wN, row, P) % P
    sig = ckt.pre[ckt.num_constants:]
    noop = next(i for i, g in enumerate(ckt.gates) if g.kind == C.NOOP)
    copy_cell = next((col, row) for row in range(n) if ckt.instances[row] == noop for col in range(C.NUM_ROUTED)
                     if int(sig[col, row]) != ident(col, row))
    arith_row = ckt.instances.index(next(i for i, g in enumerate(ckt.gates) if g.kind == C.ARITHMETIC))
    wires = [ckt.wires.copy() for _ in range(B)]
    wires[1][3, arith_row] ^= np.uint64(1)                      # gate violation (an output that is not routed anywhere)
    wires[2][copy_cell] = (int(wires[2][copy_cell]) + 1) % P    # copy violation in a Noop row
    wires[3][3, arith_row] ^= np.uint64(1)
    wires[3][copy_cell] = (int(wires[3][copy_cell]) + 1) % P
    pr = mp2.BatchedProver(ctx, fp, B)
    pr.set_preprocessed(ctx.to_device(ckt.pre))
    with pytest.raises(mp2.Mp2gError):
        pr.enable_witness_check()  # needs the permutation argument configured first
    pr.enable_permutation(C.NUM_ROUTED, 8)
    pr.enable_quotient()
    pr.set_gates(gpu_gates(mp2, ckt), ckt.num_selectors)
    with pytest.raises(mp2.Mp2gError):
        pr.witness_status()  # not enabled yet
    pr.enable_witness_check()
    d_cd, d_ph = ctx.to_device(O.rand_field(4, 1)), ctx.to_device(np.stack([ckt.pi_hash] * B))
    pr.prove([ctx.to_device(np.stack([ckt.wires] * B)), None, None], d_cd, d_ph)
    assert pr.witness_status().tolist() == [0] * B
    ref = pr.results()
    pr.prove([ctx.to_device(np.stack(wires)), None, None], d_cd, d_ph)
    with pytest.raises(mp2.Mp2gError, match="proof 1 of the batch violates a gate constraint") as e:
        pr.witness_status()
    assert e.value.flags.tolist() == [0, 2, 1, 3]
    got = pr.results()
    assert np.array_equal(got[2][0], ref[2][0]) and not np.array_equal(got[2][1], ref[2][1])  # the good proof of the batch is unaffected
    pr.enable_witness_check(False)
    pr.free()
