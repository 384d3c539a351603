"""Oracle self-consistency for the gate constraint evaluators (PARITY UNPINNED vs plonky2: there is no
golden vector for any gate in the reference). What is checked: a satisfied witness makes every filtered
constraint vanish on H; each gate kind really constrains (perturbing a wire of its row breaks it); the
complete proof verifies (FRI + the PLONK identity at zeta over the extension field); a bad witness fails."""
import numpy as np
import pytest

import circuits as C
import oracle as O

P = O.P


def params(ckt, log_n):
    return O.standard_params(log_n, (ckt.num_constants + C.NUM_ROUTED, C.NUM_WIRES, 20, 16), pow_bits=4, num_queries=3)


@pytest.mark.parametrize("kinds", [[k] for k in C.ALL_KINDS if k[0] != C.NOOP] + [C.ALL_KINDS])
def test_satisfied_witness_vanishes_on_h(kinds):
    ckt = C.build(5, [(C.NOOP, 0, 0, 0)] + [k for k in kinds if k[0] != C.NOOP], 11)
    out = C.eval_on_points(ckt, ckt.pre[:ckt.num_constants], ckt.wires)
    assert not out.any()
    # every gate kind constrains its row: flip one used wire of one of its rows
    for gi, g in enumerate(ckt.gates):
        if g.kind == C.NOOP:
            continue
        row = ckt.instances.index(gi)
        col = {C.BASE_SUM: 1, C.EXPONENTIATION: g.p0 + 1, C.RANDOM_ACCESS: 1}.get(g.kind, 0)
        bad = ckt.wires.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % P
        out = C.eval_on_points(ckt, ckt.pre[:ckt.num_constants], bad)
        assert out[:, row].any() and not np.delete(out, row, axis=1).any(), g.kind


def test_descriptor_tables_agree():
    import ctypes
    for k in C.ALL_KINDS:
        g = C.Gate(*k, 0, 0, 0)
        assert O.lib().orc_gate_degree(ctypes.byref(g)) == C.gate_degree(g)
        assert O.lib().orc_gate_num_constraints(ctypes.byref(g)) == C.gate_num_constraints(g)


def test_selector_groups():
    ckt = C.build(5, C.ALL_KINDS, 3)
    degs = [C.gate_degree(g) for g in ckt.gates]
    assert degs == sorted(degs)
    assert ckt.num_selectors > 1
    for g in ckt.gates:  # gates/selectors.rs: filter degree + gate degree fits the quotient degree factor
        assert (g.group_end - g.group_start) + C.gate_degree(g) <= C.MAX_DEGREE + 1
    few = C.build(5, [(C.NOOP, 0, 0, 0), (C.CONSTANT, 2, 0, 0), (C.ARITHMETIC, 20, 0, 0)], 3)
    assert few.num_selectors == 1  # max degree 3 + 3 gates - 1 <= 8: the single-selector special case


@pytest.mark.parametrize("kinds,log_n", [([(C.NOOP, 0, 0, 0), (C.CONSTANT, 2, 0, 0), (C.PUBLIC_INPUT, 0, 0, 0), (C.ARITHMETIC, 20, 0, 0)], 5),
                                          (C.ALL_KINDS, 6)])
def test_proof_with_gates_verifies(kinds, log_n):
    ckt = C.build(log_n, kinds, 5)
    fp = params(ckt, log_n)
    cd = O.rand_field(4, 9)
    caps, openings, proof, bgao = C.prove(ckt, fp, cd)
    assert O.pcs_verify(fp, cd, ckt.pi_hash, caps, openings, proof) == 0
    assert C.identity_check(ckt, fp, openings, bgao) == 0
    # the gate terms matter: the gate-less identity does not hold for this proof
    assert O.plonk_identity_check(fp, C.NUM_ROUTED, 8, openings, bgao) != 0
    # break one gate's witness (not a routed copy): FRI still opens correctly, the identity fails
    gi = next(i for i, g in enumerate(ckt.gates) if g.kind == C.ARITHMETIC)
    row = ckt.instances.index(gi)
    good = ckt.wires.copy()
    ckt.wires[3, row] = (int(ckt.wires[3, row]) + 1) % P
    caps, openings, proof, bgao = C.prove(ckt, fp, cd)
    assert O.pcs_verify(fp, cd, ckt.pi_hash, caps, openings, proof) == 0
    assert C.identity_check(ckt, fp, openings, bgao) != 0
    ckt.wires = good


def test_lookup_argument_oracle_self_consistency():
    """all 26 registered gate kinds + two lookup tables: the oracle's proof passes the oracle's verifier (lookup
    challenges, RE / Sum / LDC constraints, lookup polynomials in both FRI batches); a looked-up pair that is not in
    its table, a wrong multiplicity and a corrupted table row all break the PLONK identity"""
    ckt = C.build(7, C.ALL_KINDS + C.LOOKUP_KINDS, 3, luts=[(t, 100) for t in C.bits_lookup_tables()])
    assert len(ckt.gates) == 26
    fp = C.oracle_params(ckt, pow_bits=4, num_queries=3)
    assert fp.num_lookup_polys == 7 and fp.oracle_w[2] == 34
    cd = O.rand_field(4, 1)
    caps, op, pr, chal = C.prove(ckt, fp, cd)
    assert C.identity_check(ckt, fp, op, chal) == 0
    assert C.verify(ckt, fp, cd, ckt.pi_hash, caps, op, pr) == 0
    for (col, row) in ((2, ckt.luts[0]["first_lut_row"]), (1, ckt.luts[1]["last_lu_row"]), (1, ckt.luts[1]["first_lut_row"])):
        w = ckt.wires.copy()
        w[col, row] = (int(w[col, row]) + 1) % O.P
        c2, o2, p2, _ = C.prove_witness(ckt, fp, cd, w, ckt.pi_hash)
        assert C.verify(ckt, fp, cd, ckt.pi_hash, c2, o2, p2) >= 10
    # the lookup openings are part of the transcript: moving one breaks the proof
    bad = op.copy()
    bad[-1, 0] ^= np.uint64(1)
    assert C.verify(ckt, fp, cd, ckt.pi_hash, caps, bad, pr) != 0
