#!/usr/bin/env python3
"""Writes a file in the schema of tools/ref_vectors/src/main.rs (`reference_vectors.json`) from THIS
repository's CPU oracle instead of the reference: `source` says "self". It exists to exercise the consumer
(tests/test_reference_vectors.py) end to end without a Rust toolchain -- a file made here proves that the test reads the schema
and drives every section, NOT parity with the reference. Test infrastructure: imports oracle/ through tests/oracle.py.

    python tools/ref_vectors/self_vectors.py out.json [poseidon2|poseidon]
"""
import ctypes
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import circuits as OC  # noqa: E402
import oracle as O  # noqa: E402

PC = importlib.import_module("mapreduce-plonky2_amd.circuits")
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
MULT_GEN, TWO_GEN = 14293326489335486720, 7277203076849721926
VARIANT = {"poseidon2": 0, "poseidon": 1}


def ints(a):
    return [int(x) for x in np.asarray(a).ravel()]


def gate_id(g):
    """plonky2's Gate::id() strings (format!("{:?}", self) of the gate structs) for the kinds a small circuit uses"""
    k = g.kind
    if k == PC.NOOP:
        return "NoopGate"
    if k == PC.CONSTANT:
        return f"ConstantGate {{ num_consts: {g.p0} }}"
    if k == PC.PUBLIC_INPUT:
        return "PublicInputGate"
    if k == PC.ARITHMETIC:
        return f"ArithmeticGate {{ num_ops: {g.p0} }}"
    if k == PC.POSEIDON2:
        return "Poseidon2Gate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"
    if k == PC.POSEIDON:
        return "PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"
    if k == PC.BASE_SUM:
        return f"BaseSumGate {{ num_limbs: {g.p0} }} + Base: {g.p1}"
    if k == PC.RANDOM_ACCESS:
        return (f"RandomAccessGate {{ bits: {g.p0}, num_copies: {g.p1}, num_extra_constants: {g.p2}, _phantom: "
                "PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>")
    if k == PC.ARITHMETIC_EXT:
        return f"ArithmeticExtensionGate {{ num_ops: {g.p0} }}"
    if k == PC.MUL_EXT:
        return f"MulExtensionGate {{ num_ops: {g.p0} }}"
    if k == PC.REDUCING:
        return f"ReducingGate {{ num_coeffs: {g.p0} }}"
    if k == PC.REDUCING_EXT:
        return f"ReducingExtensionGate {{ num_coeffs: {g.p0} }}"
    if k == PC.EXPONENTIATION:
        return f"ExponentiationGate {{ num_power_bits: {g.p0}, _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }}<D=2>"
    if k == PC.COSET_INTERPOLATION:  # (the reference's string also lists the barycentric weights; a reader takes the two parameters)
        return (f"CosetInterpolationGate {{ subgroup_bits: {g.p0}, degree: {g.p1}, _phantom: "
                "PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>")
    if k == PC.POSEIDON_MDS:
        return "PoseidonMdsGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"
    raise ValueError(f"no id string for gate kind {k}")


def hasher_section(v):
    lib = O.lib()
    out4 = lambda: np.zeros(4, dtype=np.uint64)

    def call(fn, inp):
        a, o = O.arr(inp), out4()
        fn(v, O.p(a), O.sz(a.size), O.p(o))
        return ints(o)

    l, r, o = O.arr([1, 2, 3, 4]), O.arr([5, 6, 7, 8]), out4()
    lib.orc_two_to_one(v, O.p(l), O.p(r), O.p(o))
    return {"permute_0_to_11": ints(O.perm(np.arange(12, dtype=np.uint64), v)),
            "hash_no_pad": {str(n): ints(O.hash_n_to_m_no_pad(np.arange(n, dtype=np.uint64), 4, v)) for n in (0, 1, 4, 7, 8, 9, 17, 135)},
            "hash_pad": {str(n): call(lib.orc_hash_pad, np.arange(n, dtype=np.uint64)) for n in (0, 3, 8)},
            "hash_or_noop": {str(n): call(lib.orc_hash_or_noop, np.arange(1, n + 1, dtype=np.uint64)) for n in (3, 4, 5)},
            "two_to_one_1234_5678": ints(o)}


def fft_section(log_n):
    n = 1 << log_n
    x = O.rand_field((1, n), 0xC0FFEE02)
    padded = np.concatenate([x, np.zeros((1, n), dtype=np.uint64)], axis=1)
    return {"input": ints(x), "fft": ints(O.fft(x)), "ifft": ints(O.fft(x, inverse=True)), "coset_fft": ints(O.fft(x, coset_shift=MULT_GEN)),
            "lde1_coset_fft": ints(O.fft(padded, coset_shift=MULT_GEN))}


def batch_section(v):
    log_n, w, rate_bits, cap_h, idx = 4, 3, 3, 4, 77
    n = 1 << log_n
    values = O.rand_field(w * n, 0xC0FFEE02).reshape(w, n)
    coeffs = O.fft(values, inverse=True)
    leaves = O.lde_leaves(coeffs, rate_bits)
    levels = O.merkle_build(leaves, cap_h, v)
    return {"log_n": log_n, "polys": w, "rate_bits": rate_bits, "cap_height": cap_h, "values": [ints(r) for r in values], "coeffs": [ints(r) for r in coeffs],
            "leaves": [ints(leaves[0]), ints(leaves[1]), ints(leaves[idx])], "leaf_indices": [0, 1, idx],
            "cap": [ints(h) for h in O.merkle_cap(levels, cap_h).reshape(-1, 4)], "proof_index": idx,
            "proof_siblings": [ints(h) for h in O.merkle_prove(levels, log_n + rate_bits, cap_h, idx).reshape(-1, 4)]}


class _Ch(ctypes.Structure):
    _fields_ = [("state", ctypes.c_uint64 * 12), ("inb", ctypes.c_uint64 * 8), ("out", ctypes.c_uint64 * 8), ("n_in", ctypes.c_uint32), ("n_out", ctypes.c_uint32),
                ("variant", ctypes.c_uint32)]


def challenger_script(v):
    """the observe / squeeze script of main.rs's challenger_section on the oracle's challenger"""
    lib = O.lib()
    lib.orc_ch_get.restype = ctypes.c_uint64
    ch = _Ch()
    lib.orc_ch_init(ctypes.byref(ch), v)

    def observe(xs):
        a = O.arr(xs)
        lib.orc_ch_observe(ctypes.byref(ch), O.p(a), O.sz(a.size))

    get = lambda k: [int(lib.orc_ch_get(ctypes.byref(ch))) for _ in range(k)]
    observe([1, 2, 3])
    a = get(2)
    observe([7, 8, 9, 10])
    e = get(2)
    observe(list(range(11, 23)))
    b = get(9)
    return {"script": "observe [1,2,3]; get 2; observe_hash [7,8,9,10]; get_extension; observe [11..=22]; get 9", "first_two": a, "extension": e, "next_nine": b}


def ecgfp5_section(v):
    lib = O.lib()
    inputs = [[1, 2, 3], list(range(9)), ints(O.rand_field(17, 0xC0FFEE04))]

    def point(w, wei):
        return {"encode": ints(w), "fields": ints(wei)}

    pts = []
    for i in inputs:
        a, w, wei = O.arr([i]), np.zeros((1, 5), dtype=np.uint64), np.zeros((1, 11), dtype=np.uint64)
        lib.orc_map_to_curve_batch(v, O.p(a), O.sz(len(i)), O.sz(1), O.p(w), O.p(wei))
        pts.append((w[0].copy(), wei[0].copy()))

    def csum(ws):
        a, w, wei = O.arr(np.stack(ws)) if ws else np.zeros((0, 5), dtype=np.uint64), np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
        assert lib.orc_curve_sum(O.p(a), O.sz(len(ws)), O.p(w), O.p(wei))
        return point(w, wei)

    h = [0x0123456789ABCDEF, 0xFFFFFFFF00000000, 3, 4]
    val = h[0] | (h[1] << 64)
    k = O.arr([(val >> (32 * i)) & 0xFFFFFFFF for i in range(4)], np.uint32)
    w, wei = np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
    assert lib.orc_scalar_mul(O.p(O.arr(pts[0][0])), O.p(k), 4, O.p(w), O.p(wei))
    return {"map_to_curve": [{"input": i, "point": point(*p)} for i, p in zip(inputs, pts)], "add_0_1": csum([pts[0][0], pts[1][0]]),
            "sum_all": csum([p[0] for p in pts]), "double_0": csum([pts[0][0], pts[0][0]]), "neutral": csum([]),
            "hash_to_int": {"hash": h, "value": str(val), "flatten": [x for limb in h for x in (limb >> 32, limb & 0xFFFFFFFF)]},
            "scalar_mul_0": point(w, wei)}


def u256_words(x):
    """a U256 as the 8 big-endian u32 words of u256.rs:870-877 / left_pad32 + pack(Big)"""
    return [(int(x) >> (32 * (7 - j))) & 0xFFFFFFFF for j in range(8)]


def table_section(v, given=None):
    """the `table` section (main.rs table_section): with `given` = that section of a vector file, recomputed from ITS rows"""
    lib = O.lib()
    if given is None:
        stream = ints(O.rand_field(6 * 4 * 4, 0xC0FFEE04))
        primaries = [5, 7, 5, 9, 7, 7]
        rows = [{"primary": str(primaries[r]),
                 "values": [str(sum(stream[(r * 4 + c) * 4 + i] << (64 * i) for i in range(4))) for c in range(4)]} for r in range(6)]
        primary_id, ids, unique_ids = 1000, [1001, 1002, 1003, 1004], [1001]
    else:
        rows, primary_id, ids, unique_ids = given["rows"], int(given["primary_id"]), [int(x) for x in given["column_ids"]], [int(x) for x in given["row_unique_columns"]]
    n, C = len(rows), len(ids)
    col_ids = O.arr(ids)
    values = np.array([[u256_words(x) for x in r["values"]] for r in rows], dtype=np.uint32).reshape(n, C, 8)
    primary = np.array([u256_words(r["primary"]) for r in rows], dtype=np.uint32).reshape(n, 8)
    unique = np.ascontiguousarray(values[:, [ids.index(u) for u in unique_ids]])

    def commitment(lo, hi, old):
        out = (ctypes.c_uint8 * 32)()
        oldb = (ctypes.c_uint8 * 32)(*old) if old is not None else None
        lib.orc_update_off_chain_data_commitment(v, ctypes.c_uint64(primary_id), O.p(np.ascontiguousarray(primary[lo:hi])), O.p(col_ids), O.sz(C),
                                                 O.p(np.ascontiguousarray(values[lo:hi])), O.p(np.ascontiguousarray(unique[lo:hi])), O.sz(len(unique_ids)), O.sz(hi - lo), oldb, out)
        return bytes(out)

    w, wei = np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
    lib.orc_row_digest_batch(v, O.p(col_ids), O.sz(C), O.p(values), O.p(unique), O.sz(len(unique_ids)), O.sz(n), O.p(w), O.p(wei))
    first = commitment(0, 4, None)
    empty = O.hash_n_to_m_no_pad(np.zeros(0, dtype=np.uint64), 4, v)  # empty_poseidon_hash (mp2-common/src/poseidon.rs): H(no limbs)

    def cell(left, right, c):  # MerkleCell::aggregate: H(H(left) || H(right) || id || value), hash.to_bytes() = 4 little-endian u64
        return O.hash_n_to_m_no_pad(list(left) + list(right) + [ids[c]] + u256_words(rows[0]["values"][c]), 4, v)

    leaf1, leaf3 = cell(empty, empty, 1), cell(empty, empty, 3)
    hexof = lambda h: np.asarray(h, dtype="<u8").tobytes().hex()
    cells_root = cell(leaf1, leaf3, 2)
    sec = lambda r: int(rows[r]["values"][0])  # the secondary index is column 0 of the other columns

    def row(r, left, right):  # RowPayload::aggregate: (hash, min, max) nodes; H(hL || hR || min || max || id || value || cells root)
        lo = left[1] if left else sec(r)
        hi = right[2] if right else sec(r)
        h = O.hash_n_to_m_no_pad(list(left[0] if left else empty) + list(right[0] if right else empty) + u256_words(lo) + u256_words(hi) + [ids[0]]
                                 + u256_words(sec(r)) + list(cells_root), 4, v)
        return (h, lo, hi)

    row_a, row_b = row(0, None, None), row(2, None, None)
    row_json = lambda t: {"hash": hexof(t[0]), "min": str(t[1]), "max": str(t[2])}
    row_tree = {"leaf_row_0": row_json(row_a), "leaf_row_2": row_json(row_b), "row_1_over_left_child": row_json(row(1, row_a, None)),
                "row_1_over_right_child": row_json(row(1, None, row_b)), "row_1_over_both": row_json(row(1, row_a, row_b))}
    return {"row_tree": row_tree, "primary_id": primary_id, "column_ids": ids, "row_unique_columns": unique_ids, "rows": rows,
            "row_unique_data_row0": ints(O.hash_n_to_m_no_pad([int(x) for x in unique[0].reshape(-1)], 4, v)),
            "row_digest": {"encode": ints(w), "fields": ints(wei)},
            "commitment_rows_0_to_3": first.hex(), "commitment_updated_with_rows_4_5": commitment(4, n, first).hex(),
            "cells_tree": {"leaf_column_1": hexof(leaf1), "leaf_column_3": hexof(leaf3), "column_2_over_left_child": hexof(cell(leaf1, empty, 2)),
                           "column_2_over_both": hexof(cell(leaf1, leaf3, 2))}}


def dump_circuit(ckt, v):
    """prove (oracle, smallest PoW witness), verify, and write out what main.rs's dump_circuit writes; also returns what a wrap needs"""
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    FW = importlib.import_module("mapreduce-plonky2_amd.framework")
    cap = O.merkle_cap(O.merkle_build(O.lde_leaves(O.fft(ckt.pre, inverse=True), 3), 4, v), 4)
    dom = np.zeros(4, dtype=np.uint64)
    e = O.arr(np.zeros(0, dtype=np.uint64))
    O.lib().orc_hash_pad(v, O.p(e), O.sz(0), O.p(dom))
    digest = O.hash_n_to_m_no_pad(list(cap.reshape(-1)) + list(dom) + [ckt.log_n], 4, v)
    fp = OC.oracle_params(ckt, variant=v)
    caps, openings, proof, _ = OC.prove_witness(ckt, fp, digest, ckt.wires, ckt.pi_hash)
    assert OC.verify(ckt, fp, digest, ckt.pi_hash, caps, openings, proof) == 0
    caps[0] = cap.reshape(-1)
    wire = mp2.serialize_proof(FW.circuit_fri_params(ckt, v), ckt.num_constants, caps, openings, proof, ckt.public_inputs)
    groups = sorted({(g.selector_index, g.group_start, g.group_end) for g in ckt.gates})
    doc = {"degree_bits": ckt.log_n, "config": "CircuitConfig::standard_recursion_config()", "gates": [gate_id(g) for g in ckt.gates],
           "selector_indices": [g.selector_index for g in ckt.gates], "selector_groups": [[s_, e_] for _, s_, e_ in groups], "num_constants": ckt.num_constants,
           "k_is": [pow(MULT_GEN, j, O.P) for j in range(80)], "constants_sigmas": [ints(r) for r in ckt.pre], "wires": [ints(r) for r in ckt.wires],
           "public_inputs": ints(ckt.public_inputs), "circuit_digest": ints(digest), "constants_sigmas_cap": [ints(h) for h in cap.reshape(-1, 4)],
           "proof_bincode_hex": wire.hex(), "pow_witness": int(proof[-1])}
    return doc, (fp, cap, digest, caps, openings, proof)


def proof_sections(v):
    """main.rs's proof_section: the small circuit built by recursion.Builder and, for the Poseidon2 configuration, the first
    wrapping step over its proof (recursion.wrap_circuit = wrap_circuit.rs:64-99 at wrap_step 0). This repository's in-circuit
    verifier hashes with Poseidon2, so a file for the `original_poseidon` configuration carries no `proof_recursive` from here."""
    b = R.Builder(hasher=v)
    x, y = b.add_virtual(3), b.add_virtual(0x123456789ABCDEF0)
    s = b.add(b.mul(x, y), x)
    t = b.mul_add(s, b.constant(0xC0FFEE), y)
    b.register_public_inputs([x, t])
    ckt = b.build(min_log_n=5)
    first, (fp, cap, digest, caps, openings, proof) = dump_circuit(ckt, v)
    if v != 0:
        return first, None
    inner = R.InnerCircuit(ckt, fp, cap, digest, len(ckt.public_inputs))
    wrap = R.wrap_circuit(inner, caps, openings, proof, ckt.public_inputs)
    return first, dump_circuit(wrap, v)[0]


def make(hasher="poseidon2"):
    v = VARIANT[hasher]
    block = O.hash_n_to_m_no_pad(np.frombuffer(b"BLOCK_NUMBER", dtype=np.uint8).astype(np.uint64), 4, v)
    return {"schema": 1, "source": "self (this repository's oracle/: exercises the consumer, proves nothing about the reference)", "default_hasher": hasher,
            "field": {"order": O.P, "multiplicative_group_generator": MULT_GEN, "power_of_two_generator": TWO_GEN, "two_adicity": 32, "coset_shift": MULT_GEN,
                      "root_of_unity_log3": pow(TWO_GEN, 1 << 29, O.P), "root_of_unity_log6": pow(TWO_GEN, 1 << 26, O.P)},
            "hashers": {"poseidon2": hasher_section(0), "poseidon": hasher_section(1)}, "identifier_block_column": int(block[0]),
            "fft": {"3": fft_section(3), "10": fft_section(10)}, "polynomial_batch": batch_section(v), "challenger": challenger_script(v),
            "ecgfp5": ecgfp5_section(v), "table": table_section(v), **{k: p for k, p in zip(("proof", "proof_recursive"), proof_sections(v)) if p is not None}}


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "self_vectors.json"
    with open(out, "w") as f:
        json.dump(make(sys.argv[2] if len(sys.argv) > 2 else "poseidon2"), f)
    print("wrote", out)
