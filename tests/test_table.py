"""table.py on the CPU: the tree shapes against ryhope's sbbst, the circuit sets' shapes, and a one-row table (4 cells-tree proofs +
1 row-tree proof, every one base + wrap) with the oracle as the proving back end; the root's public inputs are the off-circuit
tree hash / digest / min / max."""
import importlib

import numpy as np
import pytest

import circuits as C
import oracle as O
from table_oracle import OracleTableWitness
from test_recursion import OracleProver

T = importlib.import_module("mapreduce-plonky2_amd.table")
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
FW = importlib.import_module("mapreduce-plonky2_amd.framework")


def test_sbbst_shape_restates_ryhope():
    """ryhope/src/tree/sbbst.rs: root = highest power of two <= n; a node's missing right child is replaced by the first
    in-range node down its left spine; in-order spans are contiguous and partition under a node"""
    assert [T.sbbst_root(n) for n in (1, 2, 3, 4, 5, 7, 8, 9)] == [1, 2, 2, 4, 4, 4, 8, 8]
    assert T.sbbst_children(4, 4) == (2, None) and T.sbbst_children(4, 2) == (1, 3) and T.sbbst_children(4, 1) == (None, None)
    assert T.sbbst_children(5, 4) == (2, 5) and T.sbbst_children(6, 4) == (2, 6) and T.sbbst_children(6, 6) == (5, None)
    assert T.sbbst_children(9, 8) == (4, 9) and T.sbbst_children(11, 8) == (4, 10) and T.sbbst_children(11, 10) == (9, 11)
    for n in range(1, 40):
        seen = []

        def walk(k):
            l, r = T.sbbst_children(n, k)
            lo, hi = T.sbbst_span(n, k)
            if l is not None:
                walk(l)
            seen.append(k)
            if r is not None:
                walk(r)
            under = [x for x in range(lo, hi + 1)]
            assert k in under

        walk(T.sbbst_root(n))
        assert seen == list(range(1, n + 1))  # a BST over 1..n holding every position once


def test_cells_tree_live_peak():
    """the proof pool of the native build is sized from the table's width: the widest live set of a row's cells tree when it is proved
    level by level and children are dropped with their parents (4 columns: two leaves + the full node above them)"""
    assert [T.cells_tree_live_peak(c) for c in (1, 2, 3, 4)] == [1, 2, 3, 3]
    for c in (6, 7, 10, 20):
        assert c // 2 < T.cells_tree_live_peak(c) <= c


def test_balanced_bst_spans():
    for n in (1, 2, 3, 7, 8, 100):
        root, nodes, spans = T.balanced_bst(n)
        assert sorted(nodes) == list(range(n)) and spans[root] == (0, n)
        for k, (l, r) in nodes.items():
            lo, hi = spans[k]
            assert (spans[l] == (lo, k) if l is not None else lo == k) and (spans[r] == (k + 1, hi) if r is not None else hi == k + 1)


@pytest.fixture(scope="module")
def params():
    empty = O.hash_n_to_m_no_pad(np.zeros(0, dtype=np.uint64), 4)
    return T.TableParams(OracleProver(), FW.circuit_fri_params, empty)


def test_circuit_sets_have_the_reference_shapes(params):
    """cells set of 4 (api.rs:143), row set of 3 (row_tree/api.rs:47); 28 / 43 public inputs + the set digest; every chain ends at
    RECURSION_THRESHOLD; the row full node (two row proofs + the cells proof = three verifiers) needs 2^14 rows and two wrap steps"""
    assert params.cells.set_size == 4 and params.rows.set_size == 3
    sh = params.shapes()
    assert all(v[-1] == R.RECURSION_THRESHOLD for v in sh.values())
    assert sh["row_full"] == [14, 13, 12] and sh["cells_full"] == [13, 12] and sh["row_leaf"][0] == 12
    assert len(params.cells.chains["cells_full"][0][0].public_inputs) == T.CELLS_IO + 4
    assert len(params.rows.chains["row_full"][0][0].public_inputs) == T.ROWS_IO + 4
    assert R.common_data(params.cells.chains["cells_leaf"][-1][0]) == params.cells.rec_common


class _Ctx:
    """the two calls expected_root_public_inputs makes on a context, by the oracle"""

    def hash_no_pad_batch(self, inputs, out_len=4, variant=0):
        a = np.asarray(inputs, dtype=np.uint64)
        if a.shape[1] == 0:
            return np.stack([O.hash_n_to_m_no_pad(np.zeros(0, dtype=np.uint64), out_len, variant)] * a.shape[0])
        return O.hash_no_pad_batch(a, out_len, variant)


def test_one_row_table_on_the_oracle_prover(params):
    table = T.SyntheticTable(1, 4, seed=0xC0FFEE04)
    root, nodes, spans = T.balanced_bst(1)
    wit = OracleTableWitness(table, spans)
    build = T.TableBuild(params, [R.ProofSession(OracleProver())], batch=4, subtree_size=1, host_threads=4)
    proof, name = build.run(table, wit, root, nodes)
    assert name == "row_leaf" and build.n_proofs == 5
    pis = proof[3]
    want = T.expected_root_public_inputs(_Ctx(), table, wit, root, nodes, spans)
    assert np.array_equal(pis[:T.ROWS_IO], want)
    assert np.array_equal(pis[T.ROWS_IO:], np.asarray(params.rows.set_digest, dtype=np.uint64))
    # one row: its tree digest is the table's (compute_table_row_digest over the single row)
    ow, owei = np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
    O.lib().orc_row_digest_batch(0, O.p(O.arr(table.col_ids)), O.sz(5), O.p(O.arr(table.values, np.uint32)), O.p(O.arr(table.values[:, 0:1], np.uint32)),
                                 O.sz(1), O.sz(1), O.p(ow), O.p(owei))
    assert np.array_equal(pis[4:15], owei)
    wckt, wcap, wdig = params.rows.chains["row_leaf"][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(pis, 4), *proof[:3]) == 0


def test_base_degree_padding_and_reference_gate_set():
    """the base-degree sweep's knobs (FrameworkCircuit(min_log_n, extra_gates)): a cells-tree leaf padded from its
    natural 2^6 rows to 2^8 rows with one row of every gate of the reference's leaf set it lacks -- the recorded witness program
    reproduces the builder's wires (extra rows included), the oracle proves the circuit and its verifier accepts (every gate's
    constraints vanish on H, the extra gates' too), and the unpadded circuit is untouched by the knobs' defaults"""
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    PC = importlib.import_module("mapreduce-plonky2_amd.circuits")
    from test_recursion import verifier_data
    empty = [int(x) for x in O.hash_n_to_m_no_pad(np.zeros(0, dtype=np.uint64), 4)]
    logic = T.cells_logic("leaf", empty)
    inputs = [int(x) for x in O.rand_field(T.CELL_LEN + 22, 5)]
    inputs[1:9] = [x & 0xFFFFFFFF for x in inputs[1:9]]
    inputs[9] = 0
    plain = R.FrameworkCircuit("cells_leaf", 0, logic, T.CELLS_IO).build_base(None, [], [], [], inputs, [1, 2, 3, 4])
    padded = R.FrameworkCircuit("cells_leaf", 0, logic, T.CELLS_IO, min_log_n=8, extra_gates=PC.LEAF_KINDS).build_base(None, [], [], [], inputs, [1, 2, 3, 4])
    assert plain.log_n == 6 and padded.log_n == 8
    have = {(g.kind, g.p0, g.p1, g.p2) for g in padded.gates}
    assert all(k in have for k in PC.LEAF_KINDS if k[0] != PC.CONSTANT), "the padded circuit carries the whole leaf gate set (constants ride in the RandomAccess row)"
    assert len(padded.gates) > len(plain.gates)
    assert np.array_equal(padded.public_inputs, plain.public_inputs)
    prog = mp2.WitnessProgram(padded)
    other = list(inputs)
    other[0] = 12345
    wires, pi_hash, pis = prog.run(np.array([[1, 2, 3, 4] + inputs, [1, 2, 3, 4] + other], dtype=np.uint64))  # inputs: the set digest, then the circuit's own
    assert np.array_equal(wires[0], padded.wires), "witness program != builder (the extra gate rows are part of the witness)"
    # the leaf set's u32 / comparison / base-4 / exponentiation rows come from their GENERATORS' tape instructions (include/mp2g.h
    # MP2G_OP_U32_ARITH ..), every operation of each row used -- not from plain wire writes
    ops = [op for _, op in R.tape_instructions(padded.tape)]
    assert ops.count(R.OP_U32_ARITH) == 3 and ops.count(R.OP_U32_SUB) == 6 and ops.count(R.OP_U32_ADD_MANY) == 5 and ops.count(R.OP_U32_RANGE_CHECK) == 7
    assert ops.count(R.OP_COMPARISON) == 1 and ops.count(R.OP_BASE_SPLIT) == 1 and ops.count(R.OP_EXP) == 1 and ops.count(R.OP_MUL_EXT) == 13
    again = R.FrameworkCircuit("cells_leaf", 0, logic, T.CELLS_IO, min_log_n=8, extra_gates=PC.LEAF_KINDS).build_base(None, [], [], [], other, [1, 2, 3, 4])
    assert np.array_equal(again.pre, padded.pre) and np.array_equal(wires[1], again.wires)
    fp = C.oracle_params(padded, pow_bits=8, num_queries=4)
    cap, cd = verifier_data(padded)
    caps, openings, proof, _ = C.prove(padded, fp, cd)
    assert C.verify(padded, fp, cd, padded.pi_hash, caps, openings, proof) == 0
    # the wrap step over it: plonky2's recursive verifier evaluates EVERY gate of the inner circuit in-circuit, the u32 / comparison /
    # exponentiation / MulExtension gates of the leaf set included (recursion.eval_gate_circuit); the strict builder stops at the first
    # constraint that does not match the opened values, and the wrap circuit's own witness satisfies all of its gates
    sfp = C.oracle_params(padded)
    sc, so, sp, _ = C.prove(padded, sfp, cd)
    inner = R.InnerCircuit(padded, sfp, cap, cd, len(padded.public_inputs))
    wrap = R.wrap_circuit(inner, sc, so, sp, padded.public_inputs)
    assert wrap.log_n in (12, 13) and np.array_equal(wrap.public_inputs, padded.public_inputs)
    assert not C.eval_on_points(wrap, wrap.pre[:wrap.num_constants], wrap.wires).any()
    tampered = so.copy()
    tampered[padded.num_constants + PC.NUM_ROUTED + 3, 0] ^= np.uint64(1)  # one opened wire value
    with pytest.raises(AssertionError):
        R.wrap_circuit(inner, sc, tampered, sp, padded.public_inputs)
    bad = padded.wires.copy()
    row = next(i for i, g in enumerate(padded.instances) if padded.gates[g].kind == PC.COMPARISON)
    bad[2, row] ^= 1  # the comparison gate's result bit
    c2, o2, p2, _ = C.prove_witness(padded, fp, cd, bad, padded.pi_hash)
    assert C.verify(padded, fp, cd, padded.pi_hash, c2, o2, p2) != 0, "a violated extra gate row must not verify"


def test_progress_file_of_a_long_block(tmp_path, monkeypatch):
    """MP2G_PROGRESS_FILE: the native build's progress writer reads the forest's atomic counter from a side thread and always leaves a
    last line when the plan is done (what a run cut off by a time limit is read from: profiles/r04/table_2p20_rows_progress.txt)"""
    import time
    import types
    T = importlib.import_module("mapreduce-plonky2_amd.table")
    path = tmp_path / "progress.txt"
    fake = types.SimpleNamespace(forest=types.SimpleNamespace(proved=40))
    monkeypatch.delenv("MP2G_PROGRESS_FILE", raising=False)
    T.NativeTableBuild._progress_writer(fake, time.perf_counter(), 50)()  # off: nothing written, nothing started
    assert not path.exists()
    monkeypatch.setenv("MP2G_PROGRESS_FILE", str(path))
    stop = T.NativeTableBuild._progress_writer(fake, time.perf_counter(), 50)
    fake.forest.proved = 50
    stop()
    lines = path.read_text().splitlines()
    assert len(lines) == 1 and lines[0].endswith("50 / 50 proofs (plan done)")
