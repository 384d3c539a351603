"""Real recursion on the CPU (no GPU needed): recursion.py's eager builder + recursive verifier, proved and
verified by the oracle. base (the map circuit of recursion-framework/tests/integration.rs:65-93) -> wrap (plonky2's
verify_proof gadget with the base circuit's verifier data as constants, wrap_circuit.rs:58-99) -> verify; the wrap
proof's public inputs are the base proof's. The GPU counterpart is tests/test_gpu_recursion.py."""
import importlib

import numpy as np
import pytest

import circuits as C
import oracle as O

R = importlib.import_module("mapreduce-plonky2_amd.recursion")


def verifier_data(ckt):
    """VerifierOnlyCircuitData by the oracle: constants_sigmas cap and H(cap || H_pad([]) || degree_bits)"""
    cap = O.merkle_cap(O.merkle_build(O.lde_leaves(O.fft(ckt.pre, inverse=True), 3), 4), 4)
    dom = O.hash_n_to_m_no_pad([1, 0, 0, 0, 0, 0, 0, 1], 4)
    return cap, O.hash_n_to_m_no_pad(list(cap.reshape(-1)) + list(dom) + [ckt.log_n], 4)


@pytest.fixture(scope="module")
def base_and_proof():
    data = O.rand_field(4, 77)
    base = R.map_circuit(data)
    fp = C.oracle_params(base)  # standard_recursion_config: 28 queries, 16-bit proof of work
    cap, cd = verifier_data(base)
    caps, openings, proof, _ = C.prove(base, fp, cd)
    assert C.verify(base, fp, cd, base.pi_hash, caps, openings, proof) == 0
    return data, base, fp, cap, cd, caps, openings, proof


def test_builder_map_circuit(base_and_proof):
    data, base, fp, cap, cd, caps, openings, proof = base_and_proof
    assert int(base.public_inputs[0]) == sum(int(x) for x in data if int(x) % 2 == 0) % O.P
    assert np.array_equal(base.public_inputs[1:], O.hash_n_to_m_no_pad(data, 4))
    assert np.array_equal(base.pi_hash, O.hash_n_to_m_no_pad(base.public_inputs, 4))
    assert not C.eval_on_points(base, base.pre[:base.num_constants], base.wires).any()  # every gate constraint vanishes on H
    # the structure does not depend on the witness: another dataset gives the same preprocessed polynomials
    other = R.map_circuit(O.rand_field(4, 78))
    assert np.array_equal(other.pre, base.pre) and not np.array_equal(other.wires, base.wires)


def test_base_wrap_verify(base_and_proof):
    data, base, fp, cap, cd, caps, openings, proof = base_and_proof
    inner = R.InnerCircuit(base, fp, cap, cd, len(base.public_inputs))
    wrap = R.wrap_circuit(inner, caps, openings, proof, base.public_inputs)
    assert wrap.log_n == 12  # RECURSION_THRESHOLD (universal_verifier_gadget/mod.rs:34)
    assert C.CONSTANT not in {g.kind for g in wrap.gates}  # the constants ride in the RandomAccess rows' spare slots
    kinds = {g.kind for g in wrap.gates}
    assert {C.POSEIDON2, C.ARITHMETIC_EXT, C.BASE_SUM, C.RANDOM_ACCESS, C.REDUCING, C.COSET_INTERPOLATION, C.PUBLIC_INPUT} <= kinds
    assert not C.eval_on_points(wrap, wrap.pre[:wrap.num_constants], wrap.wires).any()
    assert np.array_equal(wrap.public_inputs, base.public_inputs)
    wfp = C.oracle_params(wrap)
    wcap, wcd = verifier_data(wrap)
    wc, wo, wp, _ = C.prove(wrap, wfp, wcd)
    assert C.verify(wrap, wfp, wcd, wrap.pi_hash, wc, wo, wp) == 0
    # the same wrap circuit around another base proof: same circuit (digest), other witness
    data2 = O.rand_field(4, 5)
    base2 = R.map_circuit(data2)
    c2, o2, p2, _ = C.prove(base2, fp, cd)
    wrap2 = R.wrap_circuit(inner, c2, o2, p2, base2.public_inputs)
    assert np.array_equal(wrap2.pre, wrap.pre)
    assert not C.eval_on_points(wrap2, wrap2.pre[:wrap2.num_constants], wrap2.wires).any()


def test_wrap_rejects_a_bad_inner_proof(base_and_proof):
    data, base, fp, cap, cd, caps, openings, proof = base_and_proof
    inner = R.InnerCircuit(base, fp, cap, cd, len(base.public_inputs))
    bad = openings.copy()
    bad[100, 0] = (int(bad[100, 0]) + 1) % O.P
    with pytest.raises(AssertionError):
        R.wrap_circuit(inner, caps, bad, proof, base.public_inputs)  # the eager builder stops at the failing connect()
    # built anyway, the witness violates the wrap circuit (copy constraints carry the failed equalities)
    wrap = R.wrap_circuit(inner, caps, bad, proof, base.public_inputs, strict=False)
    wfp = C.oracle_params(wrap, pow_bits=4, num_queries=2)
    wcap, wcd = verifier_data(wrap)
    wc, wo, wp, _ = C.prove(wrap, wfp, wcd)
    assert C.verify(wrap, wfp, wcd, wrap.pi_hash, wc, wo, wp) != 0
    # wrong public inputs for a good proof fail too (the transcript starts from their hash)
    pis = base.public_inputs.copy()
    pis[0] = (int(pis[0]) + 1) % O.P
    with pytest.raises(AssertionError):
        R.wrap_circuit(inner, caps, openings, proof, pis)


class OracleProver:
    """proving back end for recursion.RecursiveCircuits on the CPU: the oracle proves, and verifies what it proved"""

    def verifier_data(self, ckt):
        return verifier_data(ckt)

    def prove(self, ckt):
        cap, cd = verifier_data(ckt)
        fp = C.oracle_params(ckt)
        caps, openings, proof, _ = C.prove(ckt, fp, cd)
        assert C.verify(ckt, fp, cd, ckt.pi_hash, caps, openings, proof) == 0
        return caps, openings, proof

    def prove_batch(self, ckt, wires, pi_hash):
        cap, cd = verifier_data(ckt)
        fp = C.oracle_params(ckt)
        out = []
        for w, ph in zip(wires, pi_hash):
            caps, openings, proof, _ = C.prove_witness(ckt, fp, cd, w, ph)
            assert C.verify(ckt, fp, cd, ph, caps, openings, proof) == 0
            out.append((caps, openings, proof))
        return out

    def two_to_one(self, left, right):
        return [int(x) for x in O.perm(np.array(list(left) + list(right) + [0] * 4, dtype=np.uint64))[:4]]


def test_map_reduce_with_the_universal_verifier():
    """recursion-framework/tests/integration.rs:138-228 with real circuits: two map proofs and the reduce proof over
    them, every one wrapped to the shared shape; the reduce circuit holds two universal verifiers (verifier data as
    witnesses, their digest recomputed and shown to be in the circuit set, circuit_set.rs:136-237) and exposes (sum of
    the even elements, digest of the dataset, circuit-set digest)."""
    FWm = importlib.import_module("mapreduce-plonky2_amd.framework")
    circs = [R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)]
    fw = R.RecursiveCircuits(circs, OracleProver(), FWm.circuit_fri_params)
    assert {k: [c[0].log_n for c in v] for k, v in fw.chains.items()} == {"map": [6, 12], "reduce": [13, 12]}
    data = O.rand_field(8, 5)
    p0 = fw.generate_proof("map", [], [], data[:4])
    p1 = fw.generate_proof("map", [], [], data[4:])
    root = fw.generate_proof("reduce", [p0, p1], ["map", "map"], None)
    pis = root[3]
    assert int(pis[0]) == sum(int(x) for x in data if int(x) % 2 == 0) % O.P
    want = O.hash_n_to_m_no_pad(np.concatenate([O.hash_n_to_m_no_pad(data[:4], 4), O.hash_n_to_m_no_pad(data[4:], 4)]), 4)
    assert np.array_equal(pis[1:5], want)
    # the circuit-set digest: the oracle's Merkle root (cap height 0) over the two final wrap circuits' digests
    want_set = O.merkle_cap(O.merkle_build(np.stack([np.asarray(d, dtype=np.uint64) for d in fw.digests]), 0), 0)[0]
    assert np.array_equal(pis[5:], want_set) and np.array_equal(p0[3][5:], want_set)
    # the final proof verifies under the reduce circuit's final verifier data
    wckt, wcap, wdig = fw.chains["reduce"][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(pis, 4), *root[:3]) == 0
    # the same nodes through the recorded witness programs (csrc/witness.hip) instead of the Python builder: identical proofs
    b0, b1 = fw.generate_proofs_batch("map", [([], [], data[:4]), ([], [], data[4:])])
    assert all(np.array_equal(x, y) for x, y in zip(b0, p0)) and all(np.array_equal(x, y) for x, y in zip(b1, p1))
    (broot,) = fw.generate_proofs_batch("reduce", [([b0, b1], ["map", "map"], None)])
    assert all(np.array_equal(x, y) for x, y in zip(broot, root))
    # framework.rs:588-595 test_recursive_circuit_framework_serialization: the framework read back from its parameter file
    # (no circuit is rebuilt) produces the same proofs, through the builder and through the witness programs
    blob = fw.to_bytes()
    fw2 = R.RecursiveCircuits.from_bytes(blob, circs, OracleProver(), FWm.circuit_fri_params)
    assert [int(x) for x in fw2.set_digest] == [int(x) for x in fw.set_digest] and fw2.rec_common == fw.rec_common
    q0 = fw2.generate_proof("map", [], [], data[:4])
    assert all(np.array_equal(x, y) for x, y in zip(q0, p0))
    (qroot,) = fw2.generate_proofs_batch("reduce", [([b0, b1], ["map", "map"], None)])
    assert all(np.array_equal(x, y) for x, y in zip(qroot, root))
    with pytest.raises(ValueError, match="another set of circuits"):
        R.RecursiveCircuits.from_bytes(blob, circs[::-1], OracleProver(), FWm.circuit_fri_params)
    with pytest.raises(ValueError, match="another set of circuits"):
        R.RecursiveCircuits.from_bytes(blob, [circs[0], R.FrameworkCircuit("reduce", 3, R.reduce_logic, 5)], OracleProver(), FWm.circuit_fri_params)
    # a proof of a circuit outside the set cannot be used: the membership proof does not exist
    with pytest.raises(KeyError, match="circuit digest not found"):
        fw.membership([1, 2, 3, 4])
    # a child proof with foreign verifier data fails inside the universal verifier (the digest check / the transcript)
    with pytest.raises(AssertionError):
        bad = (p1[0], p1[1], p1[2], p0[3])  # p1's proof under p0's public inputs
        fw.generate_proof("reduce", [p0, bad], ["map", "map"], None)


def test_witness_program_replays_the_builder_and_validates_its_tape():
    """mp2g_witness_program_*: the recorded tape reproduces the builder's wire matrix for other inputs (no GPU involved),
    and a malformed tape is refused at create (the tape indexes host memory)."""
    import ctypes
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    a, b = O.rand_field(4, 1), O.rand_field(4, 2)
    ca, cb = R.map_circuit(a), R.map_circuit(b)
    assert np.array_equal(ca.tape, cb.tape) and np.array_equal(ca.input_sids, cb.input_sids)  # structure only
    prog = mp2.WitnessProgram(ca)
    wires, pi_hash, pis = prog.run(np.stack([a, b]))
    assert np.array_equal(wires[0], ca.wires) and np.array_equal(wires[1], cb.wires)
    assert np.array_equal(pi_hash[1], cb.pi_hash) and np.array_equal(pis[1], cb.public_inputs)
    with pytest.raises(mp2.Mp2gError):  # non-canonical input
        prog.run(np.array([[O.P, 0, 0, 0]], dtype=np.uint64))

    def create(tape, n_slots=None):
        h = ctypes.c_void_p()
        t = np.ascontiguousarray(tape, dtype=np.uint64)
        ins, cs = np.ascontiguousarray(ca.input_sids, dtype=np.uint32), np.ascontiguousarray(ca.const_slots, dtype=np.uint64)
        rc = mp2.load().mp2g_witness_program_create(t.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(t.size), int(n_slots or ca.n_slots), int(ca.log_n),
                                                    ins.ctypes.data_as(ctypes.c_void_p), int(ins.size), cs.ctypes.data_as(ctypes.c_void_p), int(cs.shape[0]),
                                                    ctypes.byref(h))
        if rc == 0:
            mp2.load().mp2g_witness_program_free(h)
        return rc
    assert create(ca.tape) == 0
    assert create(ca.tape[:-1]) != 0                      # truncated instruction
    bad = ca.tape.copy(); bad[0] = 99
    assert create(bad) != 0                               # unknown opcode
    assert create(ca.tape, n_slots=ca.n_slots - 5) != 0   # slot out of range
    bad = ca.tape.copy(); bad[1] = 1 << 20                # row beyond the circuit (first instruction is an ARITH or a P2 row)
    assert create(bad) != 0


def test_circuit_set_gadgets():
    """recursion-framework/src/universal_verifier_gadget/circuit_set.rs:297-372 (`test_circuit_set_gadgets`): ONE circuit proves
    that each of 42 digests belongs to a circuit set of 42 (padded to 64 leaves, cap height 0) given as a public input; the
    same witness against another set is refused. The oracle proves and verifies."""
    n_elements = 42
    prover = OracleProver()

    def tree(digests):
        size = 1 << (len(digests) - 1).bit_length()
        levels = [[[int(x) for x in d] for d in digests] + [[0, 0, 0, 0]] * (size - len(digests))]
        while len(levels[-1]) > 1:
            prev = levels[-1]
            levels.append([prover.two_to_one(prev[2 * i], prev[2 * i + 1]) for i in range(len(prev) // 2)])
        return levels

    def circuit(elements, levels, strict=True):
        b = R.Builder(strict)
        set_t = [b.add_virtual(int(x)) for x in levels[-1][0]]  # CircuitSetTarget::build_target
        for idx, el in enumerate(elements):
            el_t = [b.add_virtual(int(x)) for x in el]
            bits, sib, i = [], [], idx
            for lv in levels[:-1]:
                t = b.add_virtual(i & 1)
                b.assert_bool(t)
                bits.append(t)
                sib.append([b.add_virtual(int(x)) for x in lv[i ^ 1]])
                i >>= 1
            R.verify_merkle_proof_to_cap(b, el_t, bits, None, [set_t], sib)  # check_circuit_digest_membership
        b.register_public_inputs(set_t)
        return b.build()

    elements = [O.hash_n_to_m_no_pad(O.rand_field(4, 300 + i), 4) for i in range(n_elements)]
    levels = tree(elements)
    assert len(levels) == 7
    ckt = circuit(elements, levels)
    caps, openings, proof = prover.prove(ckt)  # proves and verifies
    assert [int(x) for x in ckt.public_inputs] == levels[-1][0]
    # the membership witnesses of another set do not open to this set's digest
    other = tree([O.hash_n_to_m_no_pad(O.rand_field(4, 900 + i), 4) for i in range(n_elements)])
    with pytest.raises(AssertionError):
        circuit(elements, other)
    # and the same circuit filled with that wrong witness does not give a proof the verifier accepts
    bad = circuit(elements, other, strict=False)
    assert np.array_equal(bad.pre, ckt.pre)
    with pytest.raises(AssertionError):
        prover.prove(bad)


@pytest.mark.parametrize("do_swap", [False, True])
def test_hash_maybe_swap_is_equivalent_to_hash_n(do_swap):
    """mp2-common/src/poseidon.rs:243-277: hash_maybe_swap(a, b, false) = hash_no_pad(a || b) and hash_maybe_swap(b, a, true) is the
    same hash; the circuit exposes it as its public inputs and the oracle proves and verifies."""
    a, bb = [0] * 4, [1] * 4
    want = O.hash_n_to_m_no_pad(np.array(a + bb, dtype=np.uint64), 4)
    first, second = (bb, a) if do_swap else (a, bb)
    b = R.Builder()
    ta, tb = [b.add_virtual(x) for x in first], [b.add_virtual(x) for x in second]
    swap = b.add_virtual(int(do_swap))
    b.assert_bool(swap)
    b.register_public_inputs(R.hash_maybe_swap(b, [ta, tb], swap))
    ckt = b.build()
    assert np.array_equal(np.asarray(ckt.public_inputs, dtype=np.uint64), want)
    OracleProver().prove(ckt)  # proves and verifies


def test_poseidon_hash_flattening_and_hash_to_int():
    """mp2-common/src/poseidon.rs:193-241 (`test_poseidon_hash_flattening`, `test_hash_to_int`): the in-circuit split of hash limbs
    into u32 limbs (big-endian flattening; the little-endian 128-bit scalar of hash_to_int) equals the value-side functions; the
    circuit connects them to the expected constants and the oracle proves it. Edge limbs included: 0, 2^32 - 1, p - 1 = (2^32 - 1,
    0) whose split is only unique through the canonicity check, and the witness program replays the same wires."""
    import importlib
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    for seed, h in ((1, [int(x) for x in O.rand_field(4, 71)]), (2, [0, 0xFFFFFFFF, O.P - 1, (0xFFFFFFFE << 32) | 0xFFFFFFFF])):
        flat_want = [v for x in h for v in (x >> 32, x & 0xFFFFFFFF)]          # flatten_poseidon_hash_value
        int_want = [v for x in h[:2] for v in (x & 0xFFFFFFFF, x >> 32)]       # hash_to_int_value's u32 limbs
        b = R.Builder()
        ht = [b.add_virtual(x) for x in h]
        for got, want in zip(R.flatten_poseidon_hash_target(b, ht), flat_want):
            b.connect(got, b.constant(want))
        for got, want in zip(R.hash_to_int_target(b, ht), int_want):
            b.connect(got, b.constant(want))
        b.register_public_inputs(ht)
        ckt = b.build()
        OracleProver().prove(ckt)
        wires, pi_hash, pis = mp2.WitnessProgram(ckt).run(np.array([h], dtype=np.uint64))
        assert np.array_equal(wires[0], ckt.wires) and np.array_equal(pis[0], np.array(h, dtype=np.uint64))
    # a non-canonical split (high = 2^32 - 1 with low != 0 would be p + something) cannot be witnessed: the check refuses it
    b = R.Builder(strict=False)
    x = b.add_virtual(5)
    lo, hi = b.split_low_high(x, 32, 64)
    bad_lo, bad_hi = b.add_virtual(6), b.add_virtual(0xFFFFFFFF)   # 6 + 2^32 (2^32 - 1) = 5 + p
    assert (6 + (0xFFFFFFFF << 32)) % O.P == 5
    low_zero, high_high = b.is_equal(bad_lo, b.zero()), b.is_equal(bad_hi, b.constant(0xFFFFFFFF))
    assert b.or_(low_zero, b.not_(high_high)).v == 0


def test_parallel_regions_of_a_witness_program():
    """the query rounds of a verifier are recorded as a parallel region (OP_PAR): replayed on one thread or on eight, the wire
    matrix is the same word for word, for one proof and for a batch; malformed regions are refused at create"""
    import importlib
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    base = R.map_circuit(O.rand_field(4, 91))
    caps, openings, proof = OracleProver().prove(base)
    cap, cd = verifier_data(base)
    inner = R.InnerCircuit(base, C.oracle_params(base), cap, cd, len(base.public_inputs))
    wrap = R.wrap_circuit(inner, caps, openings, proof, base.public_inputs)
    tape = [int(x) for x in wrap.tape]
    at = next(pos for pos, op in R.tape_instructions(tape) if op == R.OP_PAR)
    assert sum(1 for _, op in R.tape_instructions(tape) if op == R.OP_PAR) == 1
    n_sections = tape[at + 1]
    assert n_sections == C.oracle_params(base).num_queries == 28
    prog = mp2.WitnessProgram(wrap)
    x = np.stack([R.proof_inputs((caps, openings, proof, base.public_inputs))] * 3)
    one = prog.run(x[:1], 1)[0]
    assert np.array_equal(one[0], wrap.wires)
    for batch, threads in ((1, 8), (3, 8), (3, 2), (3, 1)):
        w = prog.run(x[:batch], threads)[0]
        assert all(np.array_equal(w[i], wrap.wires) for i in range(batch))
        rows = prog.run(x[:batch], threads, rows=True)[0]  # one contiguous row of 135 wires per gate row
        assert rows.shape == (batch, 1 << wrap.log_n, 135) and all(np.array_equal(rows[i].T, wrap.wires) for i in range(batch))

    def refused(bad_tape):
        c = copy_of(wrap, bad_tape)
        with pytest.raises(mp2.Mp2gError, match="parallel"):
            mp2.WitnessProgram(c)

    def copy_of(ckt, new_tape):
        import copy
        c = copy.copy(ckt)
        c.tape = np.array(new_tape, dtype=np.uint64)
        return c

    t = list(tape); t[at + 2] += 1; refused(t)                      # a section boundary inside an instruction
    t = list(tape); t[at + 2 + n_sections - 1] += 10 ** 9; refused(t)  # a section longer than the tape
    t = list(tape)
    body = at + 2 + n_sections
    t[body:body] = [R.OP_PAR, 1, 0]; t[at + 2] += 3; refused(t)      # a region inside a region


def test_parallel_sections_must_be_independent():
    """the builder refuses a parallel region whose sections are not independent (one reading what another computed)"""
    b = R.Builder()
    x, y = b.add_virtual(3), b.add_virtual(5)
    with b.parallel_sections() as region:
        with region.section():
            p = b.mul(x, y)
        with region.section():
            q = b.mul(x, x)
    assert (p.v, q.v) == (15, 9) and R.OP_PAR in [op for _, op in R.tape_instructions(b.tape)]
    b2 = R.Builder()
    x, y = b2.add_virtual(3), b2.add_virtual(5)
    with pytest.raises(AssertionError, match="reads slot"):
        with b2.parallel_sections() as region:
            with region.section():
                p = b2.mul(x, y)
            with region.section():
                b2.mul(p, x)  # depends on the first section


class PoseidonOracleProver(OracleProver):
    """the oracle as a PoseidonGoldilocksConfig back end (variant 1): caps, transcript and circuit digest by the original Poseidon"""

    def verifier_data(self, ckt):
        cap = O.merkle_cap(O.merkle_build(O.lde_leaves(O.fft(ckt.pre, inverse=True), 3), 4, 1), 4)
        dom = O.hash_n_to_m_no_pad([1, 0, 0, 0, 0, 0, 0, 1], 4, 1)
        return cap, O.hash_n_to_m_no_pad(list(cap.reshape(-1)) + list(dom) + [ckt.log_n], 4, 1)

    def prove(self, ckt):
        cap, cd = self.verifier_data(ckt)
        fp = C.oracle_params(ckt, variant=1)
        caps, openings, proof, _ = C.prove(ckt, fp, cd)
        assert C.verify(ckt, fp, cd, ckt.pi_hash, caps, openings, proof) == 0
        return caps, openings, proof

    def prove_batch(self, ckt, wires, pi_hash):
        cap, cd = self.verifier_data(ckt)
        fp = C.oracle_params(ckt, variant=1)
        out = []
        for w, ph in zip(wires, pi_hash):
            caps, openings, proof, _ = C.prove_witness(ckt, fp, cd, w, ph)
            assert C.verify(ckt, fp, cd, ph, caps, openings, proof) == 0
            out.append((caps, openings, proof))
        return out


def test_final_poseidon_wrap_of_a_framework_proof():
    """verifiable-db/src/api.rs:148-214 (WrapCircuitParams, `type WrapC = PoseidonGoldilocksConfig`): a circuit over the ORIGINAL
    Poseidon config that verifies a Poseidon2 proof of the framework with the universal verifier gadget and re-exposes its public
    inputs. The public inputs are hashed by PoseidonGate rows (their hash = the Poseidon sponge of the oracle); the proof is made and
    accepted under variant 1; the builder path and the witness-program path agree; a Poseidon2 verifier does not accept it."""
    FWm = importlib.import_module("mapreduce-plonky2_amd.framework")
    fw = R.RecursiveCircuits([R.FrameworkCircuit("map", 0, R.map_logic, 5)], OracleProver(), FWm.circuit_fri_params)
    data = O.rand_field(4, 31)
    inner = fw.generate_proof("map", [], [], data)
    fin = R.FinalWrapCircuit(fw, PoseidonOracleProver(), lambda ckt: FWm.circuit_fri_params(ckt, 1))
    assert fin.ckt.log_n == R.RECURSION_THRESHOLD and any(g.kind == C.POSEIDON for g in fin.ckt.gates)
    out = fin.generate_proof(inner, "map")
    assert np.array_equal(out[3], inner[3][:5])
    (out2,) = fin.generate_proofs_batch([inner], ["map"])
    assert all(np.array_equal(a, b) for a, b in zip(out, out2))
    ph = O.hash_n_to_m_no_pad(out[3], 4, 1)  # C::InnerHasher = PoseidonHash
    assert C.verify(fin.ckt, C.oracle_params(fin.ckt, variant=1), fin.digest, ph, *out[:3]) == 0
    assert C.verify(fin.ckt, C.oracle_params(fin.ckt, variant=0), fin.digest, ph, *out[:3]) != 0
    assert C.verify(fin.ckt, C.oracle_params(fin.ckt, variant=1), fin.digest, O.hash_n_to_m_no_pad(out[3], 4, 0), *out[:3]) != 0


def test_level_schedule_of_a_witness_program(base_and_proof):
    """the device executor's schedule (csrc/witness.hip, built at mp2g_witness_program_create, no GPU needed): the number of dependency
    levels of the wrap circuit's program equals the depth computed here from the tape (level = 1 + max level of the slots read), and a
    program that writes a slot twice is still accepted by the host executor but marked unschedulable (its device run refuses it)."""
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    data, base, fp, cap, cd, caps, openings, proof = base_and_proof
    inner = R.InnerCircuit(base, fp, cap, cd, len(base.public_inputs))
    w = R.wrap_circuit(inner, caps, openings, proof, base.public_inputs)
    prog = mp2.WitnessProgram(w)
    tape = [int(x) for x in w.tape]
    level, depth, pos = {}, 0, 0
    while pos < len(tape):
        if tape[pos] == R.OP_PAR:
            pos += 2 + tape[pos + 1]
            continue
        r, wr, _, nxt = R.instruction_slots(tape, pos)
        lv = 1 + max([level.get(int(s), 0) for s in r], default=0)
        for s in wr:
            assert int(s) not in level, "the builder's programs are in SSA form"
            level[int(s)] = lv
        depth = max(depth, lv)
        pos = nxt
    assert prog.n_levels == depth and 100 < depth < 1000
    # the host replay of the scheduled program still fills the builder's wires
    wires, _, _ = prog.run(w.input_values[None])
    assert np.array_equal(wires[0], w.wires)
