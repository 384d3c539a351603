"""The recursion framework with real circuits on the HIP prover: four map proofs, two reduce levels above them, every
proof = base prove() + wrap prove() by libmp2gpu with the device-side witness check on (prove()'s panic on a bad
witness), universal verifiers inside the reduce circuit; the root's public inputs are the dataset's (integration.rs:224-228)
and the root proof passes the oracle's verifier."""
import importlib

import numpy as np
import pytest

import circuits as C
import oracle as O

pytestmark = pytest.mark.gpu
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
FW = importlib.import_module("mapreduce-plonky2_amd.framework")


def test_map_reduce_tree_of_real_proofs(ctx, mp2):
    prover = FW.GpuProver(ctx)
    circs = [R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)]
    fw = R.RecursiveCircuits(circs, prover, FW.circuit_fri_params)
    data = O.rand_field(16, 0xC0FFEE03)
    level = [fw.generate_proof("map", [], [], data[4 * i:4 * i + 4]) for i in range(4)]
    names = ["map"] * 4
    while len(level) > 1:
        level = [fw.generate_proof("reduce", [level[2 * i], level[2 * i + 1]], [names[2 * i], names[2 * i + 1]], None) for i in range(len(level) // 2)]
        names = ["reduce"] * len(level)
    pis = level[0][3]
    assert int(pis[0]) == sum(int(x) for x in data if int(x) % 2 == 0) % O.P
    hs = [O.hash_n_to_m_no_pad(data[4 * i:4 * i + 4], 4) for i in range(4)]
    while len(hs) > 1:
        hs = [O.hash_n_to_m_no_pad(np.concatenate([hs[2 * i], hs[2 * i + 1]]), 4) for i in range(len(hs) // 2)]
    assert np.array_equal(pis[1:5], hs[0])
    assert np.array_equal(pis[5:], np.asarray(fw.set_digest, dtype=np.uint64))
    wckt, wcap, wdig = fw.chains["reduce"][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(pis, 4), *level[0][:3]) == 0
    prover.free()


def test_sixteen_leaf_tree_through_the_witness_programs(ctx, mp2):
    """the same framework at 16 leaves, level by level in batches: witnesses by the recorded witness programs
    (mp2g_witness_program_run on host threads), proofs by batched HIP provers with the witness check on. 31 framework
    proofs = 62 prove() calls' worth of real circuits (map 2^6 + wrap 2^12, reduce 2^13 + wrap 2^12)."""
    prover = FW.GpuProver(ctx)
    circs = [R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)]
    fw = R.RecursiveCircuits(circs, prover, FW.circuit_fri_params)
    n_leaves = 16
    data = O.rand_field(4 * n_leaves, 0xC0FFEE03)
    level = fw.generate_proofs_batch("map", [([], [], data[4 * i:4 * i + 4]) for i in range(n_leaves)])
    # the batch path and the Python-builder path agree on a leaf
    one = fw.generate_proof("map", [], [], data[0:4])
    assert all(np.array_equal(x, y) for x, y in zip(level[0], one))
    names = ["map"] * n_leaves
    while len(level) > 1:
        jobs = [([level[2 * i], level[2 * i + 1]], [names[2 * i], names[2 * i + 1]], None) for i in range(len(level) // 2)]
        level = fw.generate_proofs_batch("reduce", jobs)
        names = ["reduce"] * len(level)
    pis = level[0][3]
    assert int(pis[0]) == sum(int(x) for x in data if int(x) % 2 == 0) % O.P
    hs = [O.hash_n_to_m_no_pad(data[4 * i:4 * i + 4], 4) for i in range(n_leaves)]
    while len(hs) > 1:
        hs = [O.hash_n_to_m_no_pad(np.concatenate([hs[2 * i], hs[2 * i + 1]]), 4) for i in range(len(hs) // 2)]
    assert np.array_equal(pis[1:5], hs[0]) and np.array_equal(pis[5:], np.asarray(fw.set_digest, dtype=np.uint64))
    wckt, wcap, wdig = fw.chains["reduce"][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(pis, 4), *level[0][:3]) == 0
    prover.free()


def _hash_chain_leaf_logic(n_hashes, reverse=False):
    """LeafCircuitWires::circuit_logic (recursion-framework/src/circuit_builder.rs:398-430): a chain of n_hashes hashes over
    (state ++ generated) -- reversed when the builder parameter's flag is set --, generated *= generator after every hash; the
    last state is the public input"""
    def logic(b, child_pis, inputs):
        vals = inputs if inputs is not None else [0] * 9
        state = [b.add_virtual(int(x)) for x in vals[:8]]
        generator = b.add_virtual(int(vals[8]))
        generated = generator
        for _ in range(n_hashes):
            hash_input = state + [generated]
            state = b.hash_n_to_m_no_pad(hash_input[::-1] if reverse else hash_input, 4)
            generated = b.mul(generated, generator)
        return state
    return logic


def _recursive_logic(b, child_pis, inputs):
    """RecursiveCircuitWires::circuit_logic (circuit_builder.rs:463-481): hash of the children's public inputs and an 8-limb payload"""
    payload = [b.add_virtual(int(x)) for x in (inputs if inputs is not None else [0] * 8)]
    return b.hash_n_to_m_no_pad([t for pis in child_pis for t in pis[:4]] + payload, 4)


def test_reference_framework_test_circuits_with_one_to_four_verifiers(ctx, mp2):
    """recursion-framework/src/framework.rs:482-563 (`TestRecursiveCircuits::run_test`) on the HIP prover: a circuit set of
    five -- a hash-chain leaf and recursive circuits with 1, 2, 3 and 4 universal verifiers --, seven leaf proofs, four of them
    under the 4-verifier circuit, three under the 3-verifier one, both results under the 2-verifier one, that under the
    1-verifier one; every proof ends with the circuit-set digest and the last one passes the oracle's verifier. The 3- and
    4-verifier circuits have 2^14 rows: their wrap chains take two steps (2^14 -> 2^13 -> 2^12, WrapCircuit::wrap_proof with W = 2).
    The leaf is the reference's: a chain of 2^12 hashes (2^13 rows before wrapping)."""
    prover = FW.GpuProver(ctx)
    circs = [R.FrameworkCircuit("leaf", 0, _hash_chain_leaf_logic(1 << 12), 4)] + \
            [R.FrameworkCircuit(f"rec{k}", k, _recursive_logic, 4) for k in (1, 2, 3, 4)]
    fw = R.RecursiveCircuits(circs, prover, FW.circuit_fri_params)
    shapes = {k: [c[0].log_n for c in v] for k, v in fw.chains.items()}
    assert all(v[-1] == R.RECURSION_THRESHOLD for v in shapes.values())
    assert len(shapes["rec4"]) == 3 and shapes["rec4"][0] == 14, shapes  # a two-step wrap chain
    set_digest = np.asarray(fw.set_digest, dtype=np.uint64)
    rng = np.random.default_rng(0xC0FFEE03)
    rand = lambda n: O.rand_field(n, int(rng.integers(1 << 30)))
    leaves = fw.generate_proofs_batch("leaf", [([], [], rand(9)) for _ in range(7)])
    (p4,) = fw.generate_proofs_batch("rec4", [(leaves[:4], ["leaf"] * 4, rand(8))])
    (p3,) = fw.generate_proofs_batch("rec3", [(leaves[4:], ["leaf"] * 3, rand(8))])
    (p2,) = fw.generate_proofs_batch("rec2", [([p4, p3], ["rec4", "rec3"], rand(8))])
    (p1,) = fw.generate_proofs_batch("rec1", [([p2], ["rec2"], rand(8))])
    for pr in leaves + [p4, p3, p2, p1]:
        assert np.array_equal(pr[3][4:], set_digest)
    wckt, wcap, wdig = fw.chains["rec1"][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(p1[3], 4), *p1[:3]) == 0
    prover.free()


@pytest.mark.parametrize("num_verifiers", [1, 2, 3, 4, 5])
def test_circuit_with_universal_verifiers(ctx, mp2, num_verifiers):
    """recursion-framework/src/circuit_builder.rs:497-583 (`test_circuit_with_universal_verifier::<NUM_VERIFIERS>`; the reference
    runs 1..5, the map-reduce tests above cover 2): a circuit set of two -- a leaf chaining 2^7 hashes and a recursive circuit with
    NUM_VERIFIERS universal verifiers --, 2 N - 1 leaf proofs, a recursive proof over N of them, then a recursive proof over the
    N - 1 others AND that recursive proof (the circuit verifies a proof of itself through the universal verifier). Both end with
    the circuit-set digest and pass the oracle's verifier. With five verifiers the base circuit has 2^15 rows."""
    prover = FW.GpuProver(ctx)
    N = num_verifiers
    fw = R.RecursiveCircuits([R.FrameworkCircuit("leaf", 0, _hash_chain_leaf_logic(1 << 7), 4), R.FrameworkCircuit("rec", N, _recursive_logic, 4)],
                             prover, FW.circuit_fri_params)
    if N == 5:
        assert fw.chains["rec"][0][0].log_n == 15
    set_digest = np.asarray(fw.set_digest, dtype=np.uint64)
    rng = np.random.default_rng(0xC0FFEE03 + N)
    rand = lambda n: O.rand_field(n, int(rng.integers(1 << 30)))
    leaves = fw.generate_proofs_batch("leaf", [([], [], rand(9)) for _ in range(2 * N - 1)])
    wckt, wcap, wdig = fw.chains["rec"][-1]
    (first,) = fw.generate_proofs_batch("rec", [(leaves[:N], ["leaf"] * N, rand(8))])
    assert np.array_equal(first[3][4:], set_digest)
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(first[3], 4), *first[:3]) == 0
    (second,) = fw.generate_proofs_batch("rec", [(leaves[N:] + [first], ["leaf"] * (N - 1) + ["rec"], rand(8))])
    assert np.array_equal(second[3][4:], set_digest)
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(second[3], 4), *second[:3]) == 0
    prover.free()


def test_wrap_circuit_keys(ctx, mp2):
    """recursion-framework/src/universal_verifier_gadget/wrap_circuit.rs:268-324: two base circuits of the same size that differ
    only in the order of their hash inputs, each with its own wrap chain. Both wrapped proofs verify; the verifier data of the
    base circuits and of the final wrap circuits differ; a proof of the variant does not go through the other circuit's chain."""
    prover = FW.GpuProver(ctx)
    rng = np.random.default_rng(0xC0FFEE21)
    rand = lambda n: O.rand_field(n, int(rng.integers(1 << 30)))

    def instance(reverse, inputs):
        b = R.Builder()
        b.register_public_inputs(_hash_chain_leaf_logic(1 << 12, reverse)(b, [], inputs))
        return b.build()

    def base_proof(base):
        caps, openings, proof = prover.prove(base)
        return (caps, openings, proof, base.public_inputs)

    bases = [instance(False, rand(9)), instance(True, rand(9))]
    wraps = [R.WrapCircuit(b, prover, FW.circuit_fri_params) for b in bases]
    assert bases[0].log_n == bases[1].log_n == 13
    for base, wc in zip(bases, wraps):
        final = wc.wrap_proof(base, base_proof(base))
        wckt, wcap, wdig = wc.final_proof_circuit_data()
        assert wckt.log_n == R.RECURSION_THRESHOLD
        assert np.array_equal(final[3], base.public_inputs)
        assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(final[3], 4), *final[:3]) == 0
    for level in (0, -1):
        (_, cap_a, dig_a), (_, cap_b, dig_b) = wraps[0].chain[level], wraps[1].chain[level]
        assert not np.array_equal(np.asarray(cap_a), np.asarray(cap_b)) and not np.array_equal(np.asarray(dig_a), np.asarray(dig_b))
    # wrapping a proof with the wrong wrap circuit does not work: refused by shape, and -- past that check -- the verifier
    # circuit built around the other circuit's verifier data is not satisfied by the proof
    other = instance(True, rand(9))
    other_proof = base_proof(other)
    with pytest.raises(AssertionError, match="not a proof of this wrap circuit"):
        wraps[0].wrap_proof(other, other_proof)
    with pytest.raises(AssertionError):
        R.wrap_proof_chain(prover, FW.circuit_fri_params, wraps[0].chain, other, other_proof)
    prover.free()


def test_common_data_for_recursion(ctx, mp2):
    """recursion-framework/src/universal_verifier_gadget/mod.rs:66-113 (`build_data_for_universal_verifier`): a no-op circuit of
    2^SHRINK_LIMIT = 2^15 rows with 3 + 4 public inputs needs exactly two wrap steps, the last one has RECURSION_THRESHOLD
    degree bits, and its common data is the one every circuit of a framework with that many public inputs ends in."""
    prover = FW.GpuProver(ctx)
    base = R.dummy_circuit(15, 3 + 4)
    assert base.log_n == 15
    wc = R.WrapCircuit(base, prover, FW.circuit_fri_params)
    assert [c[0].log_n for c in wc.chain] == [15, 13, 12]
    fw = R.RecursiveCircuits([R.FrameworkCircuit("leaf", 0, lambda b, c, i: _hash_chain_leaf_logic(4)(b, c, i)[:3], 3)], prover, FW.circuit_fri_params)
    assert R.common_data(wc.final_proof_circuit_data()[0]) == fw.rec_common
    # and a real proof of the 2^15-row circuit goes through the chain
    b = R.Builder()
    pis = [int(x) for x in O.rand_field(7, 99)]
    b.register_public_inputs([b.add_virtual(v) for v in pis])
    inst = b.build(min_log_n=15)
    caps, openings, proof = prover.prove(inst)
    final = wc.wrap_proof(inst, (caps, openings, proof, inst.public_inputs))
    wckt, wcap, wdig = wc.final_proof_circuit_data()
    assert [int(x) for x in final[3]] == pis
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(final[3], 4), *final[:3]) == 0
    prover.free()


def test_recursive_circuit_framework_serialization(ctx, mp2):
    """recursion-framework/src/framework.rs:588-595: the framework written to a parameter file and read back (no circuit is
    rebuilt: preprocessed polynomials, gate tables, witness programs and verifier data come from the file) runs the same test;
    proofs made by the original framework are accepted as children by the restored one, and the results are identical."""
    import time
    prover = FW.GpuProver(ctx)
    circs = [R.FrameworkCircuit("leaf", 0, _hash_chain_leaf_logic(1 << 7), 4), R.FrameworkCircuit("rec", 2, _recursive_logic, 4)]
    t0 = time.perf_counter()
    fw = R.RecursiveCircuits(circs, prover, FW.circuit_fri_params)
    t_build = time.perf_counter() - t0
    blob = fw.to_bytes()
    prover2 = FW.GpuProver(ctx)
    t0 = time.perf_counter()
    fw2 = R.RecursiveCircuits.from_bytes(blob, circs, prover2, FW.circuit_fri_params)
    t_load = time.perf_counter() - t0
    print(f"parameter file {len(blob) / 1e6:.1f} MB; build {t_build:.2f} s, load {t_load:.2f} s")
    assert [int(x) for x in fw2.set_digest] == [int(x) for x in fw.set_digest]
    rng = np.random.default_rng(0xC0FFEE31)
    rand = lambda n: O.rand_field(n, int(rng.integers(1 << 30)))
    leaf_inputs, payload = [rand(9) for _ in range(2)], rand(8)
    leaves = fw.generate_proofs_batch("leaf", [([], [], x) for x in leaf_inputs])
    leaves2 = fw2.generate_proofs_batch("leaf", [([], [], x) for x in leaf_inputs])
    assert all(np.array_equal(a, b) for p, q in zip(leaves, leaves2) for a, b in zip(p, q))
    (root,) = fw.generate_proofs_batch("rec", [(leaves, ["leaf"] * 2, payload)])
    (root2,) = fw2.generate_proofs_batch("rec", [(leaves, ["leaf"] * 2, payload)])  # children proved by the original framework
    assert all(np.array_equal(a, b) for a, b in zip(root, root2))
    one = fw2.generate_proof("rec", leaves2, ["leaf"] * 2, payload)  # and the builder path of the restored framework
    assert all(np.array_equal(a, b) for a, b in zip(one, root))
    wckt, wcap, wdig = fw2.chains["rec"][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(root2[3], 4), *root2[:3]) == 0
    with pytest.raises(ValueError, match="another set of circuits"):
        R.RecursiveCircuits.from_bytes(blob, circs[::-1], prover2, FW.circuit_fri_params)
    # a file made with the other hasher does not load: the circuit set does not hash to its digest
    prover3 = FW.GpuProver(ctx, variant=mp2.POSEIDON)
    with pytest.raises(ValueError, match="does not hash to its digest"):
        R.RecursiveCircuits.from_bytes(blob, circs, prover3, FW.circuit_fri_params)
    for p in (prover, prover2, prover3):
        p.free()


def test_verifier_circuit_of_recursive_circuits_set(ctx, mp2):
    """recursion-framework/src/framework.rs:598-701: circuits OUTSIDE a set verify proofs of the set with the
    RecursiveCircuitsVerifierGadget. Set one: the hash-chain leaf and a 1-verifier recursive circuit; a leaf proof and a
    recursive proof over it. Set two: a circuit that verifies a proof of ANY circuit of set one (verifier data as witnesses,
    membership in set one) and a circuit that verifies proofs of the recursive circuit only (its verifier data as constants).
    The first accepts both proofs, the second accepts the recursive proof and -- like the reference's check_panic -- refuses
    the leaf proof."""
    prover = FW.GpuProver(ctx)
    fw1 = R.RecursiveCircuits([R.FrameworkCircuit("leaf", 0, _hash_chain_leaf_logic(1 << 12), 4), R.FrameworkCircuit("rec", 1, _recursive_logic, 4)],
                              prover, FW.circuit_fri_params)
    base_proof = fw1.generate_proof("leaf", [], [], O.rand_field(9, 11))
    rec_proof = fw1.generate_proof("rec", [base_proof], ["leaf"], O.rand_field(8, 12))
    gadget = R.RecursiveCircuitsVerifierGadget(fw1)

    def any_logic(b, child_pis, inputs):  # VerifierCircuitWires: NUM_PUBLIC_INPUTS = 0
        proof, vd, mem = inputs if inputs is not None else gadget.dummy_inputs()
        gadget.verify_proof_in_circuit_set(b, proof, vd, mem)
        return []

    def fixed_logic(b, child_pis, inputs):  # VerifierCircuitFixedWires, fixed to the recursive circuit
        proof = inputs if inputs is not None else gadget.dummy_inputs()[0]
        gadget.verify_proof_fixed_circuit_in_circuit_set(b, proof, fw1.vds["rec"])
        return []

    fw2 = R.RecursiveCircuits([R.FrameworkCircuit("verifier", 0, any_logic, 0), R.FrameworkCircuit("verifier_fixed", 0, fixed_logic, 0)],
                              prover, FW.circuit_fri_params)
    set2 = np.asarray(fw2.set_digest, dtype=np.uint64)
    for proof, name in ((base_proof, "leaf"), (rec_proof, "rec")):
        vd = fw1.vds[name]
        out = fw2.generate_proof("verifier", [], [], (proof, vd, fw1.membership(vd[1])))
        wckt, wcap, wdig = fw2.chains["verifier"][-1]
        assert np.array_equal(out[3], set2)
        assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(out[3], 4), *out[:3]) == 0
    out = fw2.generate_proof("verifier_fixed", [], [], rec_proof)
    wckt, wcap, wdig = fw2.chains["verifier_fixed"][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(out[3], 4), *out[:3]) == 0
    with pytest.raises(Exception):
        fw2.generate_proof("verifier_fixed", [], [], base_proof)
    prover.free()


def test_universal_verifier_negative_tests(ctx, mp2):
    """recursion-framework/src/universal_verifier_gadget/verifier_gadget.rs:465-606 (`negative_tests`): what the universal verifier
    and the wrap circuit must refuse. `variant` is the leaf circuit with its hash inputs reversed (flag true): same shape, another
    circuit, not in the set."""
    prover = FW.GpuProver(ctx)

    def variant_logic(b, child_pis, inputs):  # LeafCircuitWires with the flag set: every hash takes its inputs reversed
        vals = inputs if inputs is not None else [0] * 9
        state = [b.add_virtual(int(x)) for x in vals[:8]]
        generator = b.add_virtual(int(vals[8]))
        generated = generator
        for _ in range(1 << 7):
            state = b.hash_n_to_m_no_pad((state + [generated])[::-1], 4)
            generated = b.mul(generated, generator)
        return state

    mk = lambda leaf_logic: R.RecursiveCircuits([R.FrameworkCircuit("leaf", 0, leaf_logic, 4), R.FrameworkCircuit("rec", 1, _recursive_logic, 4)],
                                                prover, FW.circuit_fri_params)
    fw, fw_var = mk(_hash_chain_leaf_logic(1 << 7)), mk(variant_logic)
    assert not np.array_equal(fw.vds["leaf"][1], fw_var.vds["leaf"][1])
    fw_var.set_digest = fw.set_digest  # the variant's proof claims the right circuit set, as in the reference's test
    inputs = O.rand_field(9, 21)
    rec = fw.circuits["rec"]
    # 1. the wrap circuit of the leaf refuses a base proof of the variant
    var_base = fw_var.circuits["leaf"].build_base(fw_var, [], [], [], inputs, fw.set_digest)
    caps, openings, proof = prover.prove(var_base)
    leaf_ckt, leaf_cap, leaf_dig = fw.chains["leaf"][0]
    inner = R.InnerCircuit(leaf_ckt, FW.circuit_fri_params(leaf_ckt), leaf_cap, leaf_dig, len(leaf_ckt.public_inputs))
    with pytest.raises(AssertionError):
        R.wrap_circuit(inner, caps, openings, proof, var_base.public_inputs)
    # the variant's own wrap chain accepts it: a valid final proof of a circuit outside the set
    wrapped = fw_var.generate_proof("leaf", [], [], inputs)
    var_vd, leaf_vd = fw_var.vds["leaf"], fw.vds["leaf"]
    # 2. its digest is not in the set
    with pytest.raises(KeyError, match="circuit digest not found"):
        fw.membership(var_vd[1])
    build = lambda vd, mem: rec.build_base(fw, [wrapped], [vd], [mem], O.rand_field(8, 22), fw.set_digest)
    # 3. verifier data of a circuit that IS in the set do not verify the proof
    with pytest.raises(AssertionError):
        build(leaf_vd, fw.membership(leaf_vd[1]))
    # 4. membership proved against another set (one that holds the variant): the set digest differs
    with pytest.raises(AssertionError):
        build(var_vd, fw_var.membership(var_vd[1]))
    # 5. a valid membership proof of the leaf's digest beside the variant's verifier data
    with pytest.raises(AssertionError):
        build(var_vd, fw.membership(leaf_vd[1]))
    # 6. the variant's cap under the leaf's circuit digest
    with pytest.raises(AssertionError):
        build((var_vd[0], leaf_vd[1]), fw.membership(leaf_vd[1]))
    # 3 again through the production path: the witness program fills the wires without judging them, and prove()'s device-side
    # witness check refuses them (plonky2's prove() panics on an unsatisfied witness)
    with pytest.raises(Exception, match="witness"):
        fw.generate_proofs_batch("rec", [([wrapped], ["leaf"], O.rand_field(8, 22))])
    # and the same builder accepts the honest input
    good = fw.generate_proof("leaf", [], [], inputs)
    rec.build_base(fw, [good], [leaf_vd], [fw.membership(leaf_vd[1])], O.rand_field(8, 22), fw.set_digest)
    prover.free()


def test_reduce_circuit_with_testing_framework(ctx, mp2):
    """recursion-framework/tests/integration.rs:262-310: the reduce circuit tested in isolation with TestingRecursiveCircuits --
    its two input proofs are dummy proofs with chosen public inputs (a sum and a hash each), its own public inputs the sum of the
    sums and the hash of the hashes"""
    prover = FW.GpuProver(ctx)
    tf = R.TestingRecursiveCircuits([R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)], prover, FW.circuit_fri_params)
    test_pis = [np.concatenate([O.rand_field(1, 40 + i), O.hash_n_to_m_no_pad(O.rand_field(8, 50 + i), 4)]) for i in range(2)]
    proof = tf.generate_proof_from_public_inputs("reduce", test_pis, None)
    pis = proof[3]
    assert int(pis[0]) == (int(test_pis[0][0]) + int(test_pis[1][0])) % O.P
    assert np.array_equal(pis[1:5], O.hash_n_to_m_no_pad(np.concatenate([test_pis[0][1:], test_pis[1][1:]]), 4))
    assert np.array_equal(pis[5:], np.asarray(tf.fw.set_digest, dtype=np.uint64))
    wckt, wcap, wdig = tf.fw.chains["reduce"][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(pis, 4), *proof[:3]) == 0
    prover.free()


def test_independent_trees_in_parallel_sessions(ctx, mp2):
    """two independent 4-leaf trees proved at the same time, one thread + GPU context + ProofSession each (the way
    bench.py --workload recursion --trees N fills the GPU while another tree's witnesses are generated): same root
    proofs, word for word, as the two trees proved one after the other on the framework's own session"""
    import threading
    prover = FW.GpuProver(ctx)
    circs = [R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)]
    fw = R.RecursiveCircuits(circs, prover, FW.circuit_fri_params)
    datas = [O.rand_field(16, 0xC0FFEE03 + t) for t in range(2)]

    def tree(data, session):
        level = fw.generate_proofs_batch("map", [([], [], data[4 * i:4 * i + 4]) for i in range(4)], session=session)
        names = ["map"] * 4
        while len(level) > 1:
            jobs = [([level[2 * i], level[2 * i + 1]], [names[2 * i], names[2 * i + 1]], None) for i in range(len(level) // 2)]
            level = fw.generate_proofs_batch("reduce", jobs, session=session)
            names = ["reduce"] * len(level)
        return level[0]

    sequential = [tree(d, None) for d in datas]
    ctxs = [mp2.Context(0), mp2.Context(0)]
    provers = [FW.GpuProver(c) for c in ctxs]
    sessions = [R.ProofSession(p) for p in provers]
    out = [None, None]
    ths = [threading.Thread(target=lambda t=t: out.__setitem__(t, tree(datas[t], sessions[t]))) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for t in range(2):
        assert out[t] is not None
        assert all(np.array_equal(x, y) for x, y in zip(out[t], sequential[t]))
        assert int(out[t][3][0]) == sum(int(x) for x in datas[t] if int(x) % 2 == 0) % O.P
    for p in provers + [prover]:
        p.free()
    for c in ctxs:
        c.close()


def test_two_ranks_real_recursion_with_proof_handoff():
    """bench.py --workload recursion on two ranks (gloo rendezvous, both on the test box's GPU): each rank proves an 8-leaf
    tree of real framework proofs, then rank 1's root proof travels to rank 0 (sharding.send_device_proof / recv_device_proof: device
    tensors over RCCL, host tensors in this gloo run), whose reduce node verifies both roots in-circuit; the run itself
    asserts that the final public input is the sum of the even elements of both ranks' data."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "recursion", "--batch", "8", "--trees", "2", "--steps", "1", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MP2G_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["shapes"] == {"map": [6, 12], "reduce": [13, 12]}
    assert line["framework_proofs_per_s"] > 0 and len(line["config"]["root_public_inputs"]) == 9


def test_1024_leaf_aggregation_of_real_framework_proofs(ctx, mp2):
    """BASELINE configs[2] on the REAL circuits (recursion-framework/tests/integration.rs:138-261 at 1024 leaves): 1024 map proofs and
    the 1023 reduce proofs above them (two universal verifiers each), level by level in batches of 128 through generate_proofs_batch
    -- device-side witness programs, base prove() + wrap, witness check on: 2047 framework proofs. The root's public inputs are (sum
    of the even elements, hash tree of the chunks, circuit-set digest) as integration.rs:224-228; the first and the last node of
    every level pass the oracle's verifier; one node per level is re-proved from its captured witness by the oracle, bit for bit."""
    prover = FW.GpuProver(ctx, capacity=128)
    fw = R.RecursiveCircuits([R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)], prover, FW.circuit_fri_params)
    n_leaves, chunk = 1024, 128
    data = O.rand_field(4 * n_leaves, 0xC0FFEE03)

    def batched(name, jobs):
        out = []
        for lo in range(0, len(jobs), chunk):
            out += fw.generate_proofs_batch(name, jobs[lo:lo + chunk])
        return out

    level = batched("map", [([], [], data[4 * i:4 * i + 4]) for i in range(n_leaves)])
    names, n_proofs, levels = ["map"] * n_leaves, n_leaves, [("map", level)]
    while len(level) > 1:
        level = batched("reduce", [([level[2 * i], level[2 * i + 1]], [names[2 * i], names[2 * i + 1]], None) for i in range(len(level) // 2)])
        names = ["reduce"] * len(level)
        n_proofs += len(level)
        levels.append(("reduce", level))
    assert n_proofs == 2047 and len(levels) == 11
    pis = level[0][3]
    assert int(pis[0]) == sum(int(x) for x in data if int(x) % 2 == 0) % O.P
    hs = O.hash_no_pad_batch(data.reshape(n_leaves, 4), 4)
    while len(hs) > 1:
        hs = O.hash_no_pad_batch(hs.reshape(len(hs) // 2, 8), 4)
    assert np.array_equal(pis[1:5], hs[0]) and np.array_equal(pis[5:], np.asarray(fw.set_digest, dtype=np.uint64))
    for name, lv in levels:
        wckt, wcap, wdig = fw.chains[name][-1]
        ofp = C.oracle_params(wckt)
        for pr in {0: lv[0], len(lv) - 1: lv[-1]}.values():
            assert C.verify(wckt, ofp, wdig, O.hash_n_to_m_no_pad(pr[3], 4), *pr[:3]) == 0
    # one node per level (cycling through the positions) again with capture: the same final proof, and every prove() of its chain
    # equals the oracle's proof of the captured witness. Three levels are enough for the CPU budget: leaves, the middle, the root.
    for li in (0, 5, 10):
        name, lv = levels[li]
        i = (7 * li) % len(lv)
        job = ([], [], data[4 * i:4 * i + 4]) if li == 0 else ([levels[li - 1][1][2 * i], levels[li - 1][1][2 * i + 1]], [levels[li - 1][0]] * 2, None)
        cap = []
        (again,) = fw.generate_proofs_batch(name, [job], capture=cap)
        assert all(np.array_equal(a, b) for a, b in zip(again, lv[i]))
        for (nm, step, ckt, digest, wires, ph, caps, openings, proof) in cap:
            oc, oo, op, _ = C.prove_witness(ckt, C.oracle_params(ckt), np.asarray(digest, dtype=np.uint64), wires, ph)
            assert np.array_equal(oc, caps) and np.array_equal(oo, openings) and np.array_equal(op, proof), f"level {li} {nm} step {step}"
    prover.free()


def test_device_resident_child_proofs(ctx, mp2):
    """recursion.DeviceProof: a parent proved over children that never left the device (the word ranges of the children's prover
    outputs, copied into the parent's witness inputs by device copies -- how proofs cross ranks over RCCL) is the parent proved over
    the downloaded children, word for word; and the host form of a DeviceProof is the proof."""
    prover = FW.GpuProver(ctx)
    fw = R.RecursiveCircuits([R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)], prover, FW.circuit_fri_params)
    data = O.rand_field(8, 0xC0FFEE05)
    leaves = fw.generate_proofs_batch("map", [([], [], data[:4]), ([], [], data[4:])])
    dps = [prover.last_device_proof(b) for b in range(2)]
    for dp, leaf in zip(dps, leaves):
        caps, openings, fri, pis = dp.to_host(ctx)
        assert np.array_equal(caps[1:4], leaf[0][1:4]) and np.array_equal(openings, leaf[1]) and np.array_equal(fri, leaf[2]) and np.array_equal(pis, leaf[3])
    (on_device,) = fw.generate_proofs_batch("reduce", [(dps, ["map", "map"], None)])
    (on_host,) = fw.generate_proofs_batch("reduce", [(leaves, ["map", "map"], None)])
    assert all(np.array_equal(a, b) for a, b in zip(on_device, on_host))
    # mixed: one child on the device, one on the host; and the host-witness back end downloads a DeviceProof by itself
    dps = None
    leaves2 = fw.generate_proofs_batch("map", [([], [], data[:4])])
    dp = prover.last_device_proof(0)  # the map proof, in the outputs of the map chain's last prover (the reduce chains below use others)
    (mixed,) = fw.generate_proofs_batch("reduce", [([dp, leaves[1]], ["map", "map"], None)])
    assert all(np.array_equal(a, b) for a, b in zip(mixed, on_host))
    host_prover = FW.GpuProver(ctx, device_witness=False)
    (via_host,) = fw.generate_proofs_batch("reduce", [([dp, leaves[1]], ["map", "map"], None)], session=R.ProofSession(host_prover))
    assert all(np.array_equal(a, b) for a, b in zip(via_host, on_host))
    host_prover.free()
    prover.free()


def test_final_poseidon_wrap_on_the_gpu(ctx, mp2):
    """verifiable-db/src/api.rs:148-214,198-209: the final wrap -- a PoseidonGoldilocksConfig circuit that verifies a Poseidon2
    proof of the framework (here a reduce proof over two map proofs) in the circuit set and re-exposes its public inputs -- on the
    HIP prover with variant = Poseidon (Merkle caps, challenger, circuit digest by the original Poseidon; the public inputs hashed
    by PoseidonGate rows of the device-side witness program). Bit-exact against the oracle's variant-1 prover on the captured
    witness, accepted by its variant-1 verifier, refused under Poseidon2; the builder path gives the same proof."""
    prover = FW.GpuProver(ctx)
    fw = R.RecursiveCircuits([R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)], prover, FW.circuit_fri_params)
    data = O.rand_field(8, 0xC0FFEE06)
    leaves = fw.generate_proofs_batch("map", [([], [], data[:4]), ([], [], data[4:])])
    (root,) = fw.generate_proofs_batch("reduce", [(leaves, ["map", "map"], None)])
    wrap_prover = FW.GpuProver(ctx, variant=mp2.POSEIDON)
    fin = R.FinalWrapCircuit(fw, wrap_prover, lambda ckt: FW.circuit_fri_params(ckt, mp2.POSEIDON))
    assert fin.ckt.log_n == R.RECURSION_THRESHOLD
    cap = []
    outs = fin.generate_proofs_batch([root, leaves[0]], ["reduce", "map"], capture=cap)  # any circuit of the set
    for out, inner in zip(outs, (root, leaves[0])):
        assert np.array_equal(out[3], inner[3][:5])
        ph = O.hash_n_to_m_no_pad(out[3], 4, 1)
        ofp = C.oracle_params(fin.ckt, variant=1)
        assert C.verify(fin.ckt, ofp, fin.digest, ph, *out[:3]) == 0
        assert C.verify(fin.ckt, C.oracle_params(fin.ckt, variant=0), fin.digest, ph, *out[:3]) != 0
    (nm, step, ckt, digest, wires, ph, caps, openings, proof) = cap[0]
    oc, oo, op, _ = C.prove_witness(ckt, C.oracle_params(ckt, variant=1), np.asarray(digest, dtype=np.uint64), wires, ph)
    assert np.array_equal(oc, caps) and np.array_equal(oo, openings) and np.array_equal(op, proof)
    one = fin.generate_proof(root, "reduce")
    assert all(np.array_equal(a, b) for a, b in zip(one, outs[0]))
    # the host replay of the same program (PoseidonGate rows by the host executor) fills the same wires
    vd = fw.vds["reduce"]
    hw, hph, _ = fin.program().run(R.universal_inputs(root, vd, fw.membership(vd[1]))[None])
    assert np.array_equal(hw[0], wires) and np.array_equal(hph[0], ph)
    wrap_prover.free()
    prover.free()


def test_device_proof_through_torch_tensors_and_rccl(ctx, mp2):
    """the RCCL leg of the cross-rank hand-off on the one GPU of the test box: a world-size-1 "nccl" process group (= RCCL) is
    initialised the way bench.py does it, a final proof's word ranges are wrapped as torch device tensors without a copy
    (sharding._RawView, what send_device_proof sends), pass through an RCCL collective, and the tensors a receiver would hold
    (recv_device_proof builds a DeviceProof over their addresses) feed a parent's device-side witness: the parent proof equals the
    one proved from the downloaded child. (Two ranks on two GPUs differ from this only in send / recv replacing the collective.)"""
    import os
    import torch
    import torch.distributed as dist
    sharding = importlib.import_module("mapreduce-plonky2_amd.sharding")
    prover = FW.GpuProver(ctx)
    fw = R.RecursiveCircuits([R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)], prover, FW.circuit_fri_params)
    data = O.rand_field(8, 0xC0FFEE07)
    leaves = fw.generate_proofs_batch("map", [([], [], data[:4]), ([], [], data[4:])])
    dp = prover.last_device_proof(1)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ctx.sync()
        views = [torch.as_tensor(sharding._RawView(ptr, n, dp.keep), device=dev) for ptr, n in dp.parts]
        assert [v.numel() for v in views] == [n for _, n in dp.parts] and all(v.data_ptr() == ptr for v, (ptr, _) in zip(views, dp.parts))
        received = [v.clone() for v in views]  # what the other rank's recv buffers would hold
        for t in received:
            dist.all_reduce(t)  # RCCL touches the buffers (sum over one rank = identity)
        torch.cuda.synchronize()
        child = R.DeviceProof([(t.data_ptr(), t.numel()) for t in received], keep=received)
        host = child.to_host(ctx)
        assert np.array_equal(host[3], leaves[1][3]) and np.array_equal(host[2], leaves[1][2]) and np.array_equal(host[1], leaves[1][1])
        (a,) = fw.generate_proofs_batch("reduce", [([leaves[0], child], ["map", "map"], None)])
        (b,) = fw.generate_proofs_batch("reduce", [(leaves, ["map", "map"], None)])
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    finally:
        dist.destroy_process_group()
    prover.free()
