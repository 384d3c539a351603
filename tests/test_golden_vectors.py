"""Committed golden vectors (tests/golden/oracle_vectors.json, made by tools/gen_golden.py): the
oracle reproduces them on CPU, the HIP path reproduces them on the GPU."""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))


def fnv(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def digest_inputs():
    rng = np.random.default_rng(20)
    col_ids = O.rand_field(4, 0xC0FFEE04)
    values = rng.integers(0, 1 << 32, size=(10, 4, 8), dtype=np.uint32)
    return col_ids, values, values[:, :1, :].copy()


def test_oracle_reproduces_golden():
    x = O.rand_field((4, 9), 0xC0FFEE04)
    assert O.hash_no_pad_batch(x, 4, 0).tolist() == G["hash_no_pad_9_to_4"]["poseidon2"]
    assert O.hash_no_pad_batch(x, 4, 1).tolist() == G["hash_no_pad_9_to_4"]["poseidon"]
    a = O.rand_field((1, 16), 0xC0FFEE02)
    assert O.fft(a)[0].tolist() == G["ntt_16"]["forward"]
    ws = tuple(G["pcs_prove_2p6"]["oracle_w"])
    ofp = O.standard_params(6, ws, pow_bits=6, num_queries=4)
    pv = [O.rand_field((w, 64), 100 + i) for i, w in enumerate(ws)]
    _, openings, proof = O.pcs_prove(ofp, pv, O.rand_field(4, 1), O.rand_field(4, 2))
    assert fnv(proof) == G["pcs_prove_2p6"]["proof_fnv1a"] and int(proof[-1]) == G["pcs_prove_2p6"]["pow_witness"]


def golden_gate_circuit():
    import circuits as C
    ckt = C.build(5, C.ALL_KINDS, 77)
    g = G["gate_constraints_5pts"]
    assert [[x.kind, x.p0, x.p1, x.p2, x.selector_index, x.group_start, x.group_end] for x in ckt.gates] == g["gates"]
    assert ckt.pi_hash.tolist() == g["pi_hash"] and ckt.num_selectors == g["num_selectors"]
    consts, wires = O.rand_field((ckt.num_constants, 5), 11), O.rand_field((C.NUM_WIRES, 5), 12)
    consts[:ckt.num_selectors, :] = np.arange(5, dtype=np.uint64)[None, :] + np.arange(ckt.num_selectors, dtype=np.uint64)[:, None] * np.uint64(3)
    return C, ckt, consts, wires


def test_oracle_reproduces_golden_gates():
    C, ckt, consts, wires = golden_gate_circuit()
    ev = C.eval_on_points(ckt, consts, wires)
    g = G["gate_constraints_5pts"]
    assert ev[0].tolist() == g["c0"] and ev[1].tolist() == g["c1"] and fnv(ev) == g["all_fnv1a"]
    p = G["gate_level_proof_2p5"]
    assert fnv(ckt.wires) == p["wires_fnv1a"] and fnv(ckt.pre) == p["pre_fnv1a"]  # the witness generator is frozen too
    ofp = O.standard_params(5, (ckt.num_constants + C.NUM_ROUTED, C.NUM_WIRES, 20, 16), pow_bits=5, num_queries=3)
    gc, go, gp, _ = C.prove(ckt, ofp, O.rand_field(4, 3))
    assert fnv(gc) == p["caps_fnv1a"] and fnv(go) == p["openings_fnv1a"] and fnv(gp) == p["proof_fnv1a"]


@pytest.mark.gpu
def test_hip_reproduces_golden_gates(ctx, mp2):
    C, ckt, consts, wires = golden_gate_circuit()
    gates = [mp2.Gate(*row) for row in G["gate_constraints_5pts"]["gates"]]
    ev = mp2.eval_gate_constraints(ctx, gates, ckt.num_selectors, consts, wires, ckt.pi_hash)
    g = G["gate_constraints_5pts"]
    assert ev[0].tolist() == g["c0"] and ev[1].tolist() == g["c1"] and fnv(ev) == g["all_fnv1a"]
    p = G["gate_level_proof_2p5"]
    fp = mp2.standard_recursion_params(5, (ckt.num_constants + C.NUM_ROUTED, C.NUM_WIRES, 20, 16), pow_bits=5, num_queries=3)
    pr = mp2.BatchedProver(ctx, fp, 1)
    pr.set_preprocessed(ctx.to_device(ckt.pre))
    pr.enable_permutation(C.NUM_ROUTED, 8)
    pr.enable_quotient()
    pr.set_gates(gates, ckt.num_selectors)
    pr.prove([ctx.to_device(ckt.wires[None]), None, None], ctx.to_device(O.rand_field(4, 3)), ctx.to_device(ckt.pi_hash[None]))
    caps, openings, proofs = pr.results()
    assert fnv(caps[0]) == p["caps_fnv1a"] and fnv(openings[0]) == p["openings_fnv1a"] and fnv(proofs[0]) == p["proof_fnv1a"]
    pr.free()


@pytest.mark.gpu
def test_hip_reproduces_golden(ctx, mp2):
    x = O.rand_field((4, 9), 0xC0FFEE04)
    assert ctx.hash_no_pad_batch(x, 4, 0).tolist() == G["hash_no_pad_9_to_4"]["poseidon2"]
    assert ctx.hash_no_pad_batch(x, 4, 1).tolist() == G["hash_no_pad_9_to_4"]["poseidon"]
    a = O.rand_field((1, 16), 0xC0FFEE02)
    assert ctx.ntt(a)[0].tolist() == G["ntt_16"]["forward"]
    assert ctx.ntt(a, coset_shift=O.MULT_GEN)[0].tolist() == G["ntt_16"]["coset_g"]
    b = mp2.PolynomialBatch.from_values(ctx, O.rand_field((5, 64), 0xC0FFEE01))
    assert fnv(b.cap) == G["commit_5x64"]["cap_fnv1a"] and b.cap[0].tolist() == G["commit_5x64"]["cap0"]
    ws = tuple(G["pcs_prove_2p6"]["oracle_w"])
    fp = mp2.standard_recursion_params(6, ws, pow_bits=6, num_queries=4)
    pv = [O.rand_field((w, 64), 100 + i) for i, w in enumerate(ws)]
    _, openings, proof = mp2.pcs_prove(ctx, fp, pv, O.rand_field(4, 1), O.rand_field(4, 2))
    assert fnv(proof) == G["pcs_prove_2p6"]["proof_fnv1a"] and fnv(openings) == G["pcs_prove_2p6"]["openings_fnv1a"]
    w, wei = mp2.map_to_curve_batch(ctx, O.rand_field((3, 9), 7), weierstrass=True)
    assert w.tolist() == G["map_to_curve_9"]["encodings"] and wei.tolist() == G["map_to_curve_9"]["weierstrass"]
    col_ids, values, unique = digest_inputs()
    dw, _ = mp2.compute_table_row_digest(ctx, col_ids, values, unique)
    assert dw.tolist() == G["row_digest_10x4"]["encoding"]


def _commitment_inputs():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden_commitment", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "gen_golden_commitment.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.inputs()


GC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "commitment_vectors.json")))


def test_oracle_reproduces_golden_commitment():
    pid, primary, col_ids, values, uniq, old = _commitment_inputs()
    unique = np.ascontiguousarray(values[:, uniq, :])
    for name, v in (("poseidon2", 0), ("poseidon", 1)):
        for key, oc in (("fresh", None), ("update", old)):
            o = np.zeros(32, dtype=np.uint8)
            ob = np.frombuffer(oc, dtype=np.uint8).copy() if oc is not None else None
            O.lib().orc_update_off_chain_data_commitment(v, ctypes.c_uint64(pid), O.p(O.arr(primary, np.uint32)), O.p(col_ids), O.sz(3), O.p(O.arr(values, np.uint32)),
                                                         O.p(O.arr(unique, np.uint32)), O.sz(2), O.sz(9), O.p(ob) if ob is not None else None, O.p(o))
            assert o.tobytes().hex() == GC[name][key], (name, key)


@pytest.mark.gpu
def test_hip_reproduces_golden_commitment(ctx):
    import importlib
    dg = importlib.import_module("mapreduce-plonky2_amd.digest")
    pid, primary, col_ids, values, uniq, old = _commitment_inputs()
    ucols = [col_ids[i] for i in uniq]
    for name, v in (("poseidon2", 0), ("poseidon", 1)):
        assert dg.update_off_chain_data_commitment(ctx, pid, primary, col_ids, values, ucols, None, v).hex() == GC[name]["fresh"]
        assert dg.update_off_chain_data_commitment(ctx, pid, primary, col_ids, values, ucols, old, v).hex() == GC[name]["update"]


def _tape_program(mp2, spec):
    import ctypes
    t = np.ascontiguousarray(spec["tape"], dtype=np.uint64)
    ins = np.ascontiguousarray(spec["input_sids"], dtype=np.uint32)
    cs = np.ascontiguousarray(spec["const_slots"], dtype=np.uint64).reshape(-1, 2)
    h = ctypes.c_void_p()
    rc = mp2.load().mp2g_witness_program_create(t.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(t.size), spec["n_slots"], spec["log_n"],
                                                ins.ctypes.data_as(ctypes.c_void_p), int(ins.size), cs.ctypes.data_as(ctypes.c_void_p), int(cs.shape[0]), ctypes.byref(h))
    assert rc == 0, mp2.load().mp2g_last_error()
    return h


@pytest.mark.parametrize("name", ["c_witness_tape_demo", "leaf_gate_opcodes"])
def test_host_replay_reproduces_the_golden_tapes(name):
    """tests/golden/witness_tape_vectors.json (tools/gen_golden_tape.py): the public tape format as data -- the hand-written tape of
    examples/c_witness_tape.c and a tape using every round-6 opcode, opcodes by number; the library's HOST replay must give the wire
    matrices and probe values the file holds (they were computed by the Python builder, not by the library)"""
    import ctypes
    import importlib
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    spec = json.load(open(os.path.join(HERE, "golden", "witness_tape_vectors.json")))[name]
    h = _tape_program(mp2, spec)
    n = 1 << spec["log_n"]
    probe = np.ascontiguousarray(spec["probe"], dtype=np.uint32)
    for case in spec["cases"]:
        inputs = np.ascontiguousarray([case["inputs"]], dtype=np.uint64)
        wires = np.zeros((135, n), dtype=np.uint64)
        got = np.zeros(probe.size, dtype=np.uint64)
        rc = mp2.load().mp2g_witness_program_run(h, inputs.ctypes.data_as(ctypes.c_void_p), 1, 1, wires.ctypes.data_as(ctypes.c_void_p),
                                                 probe.ctypes.data_as(ctypes.c_void_p), int(probe.size), got.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0 and fnv(wires) == case["wires_fnv1a"] and [int(x) for x in got] == case["probe"]
    mp2.load().mp2g_witness_program_free(h)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c_witness_tape_demo", "leaf_gate_opcodes"])
def test_device_replay_reproduces_the_golden_tapes(ctx, mp2, name):
    """the same tapes through mp2g_witness_program_run_dev: all cases as one batch"""
    import ctypes
    spec = json.load(open(os.path.join(HERE, "golden", "witness_tape_vectors.json")))[name]
    h = _tape_program(mp2, spec)
    n, B = 1 << spec["log_n"], len(spec["cases"])
    probe = np.ascontiguousarray(spec["probe"], dtype=np.uint32)
    assert mp2.load().mp2g_witness_program_set_probe(h, probe.ctypes.data_as(ctypes.c_void_p), int(probe.size)) == 0
    d_in = ctx.to_device(np.ascontiguousarray([c["inputs"] for c in spec["cases"]], dtype=np.uint64))
    d_w, d_pr = ctx.alloc(B * 135 * n * 8), ctx.alloc(B * probe.size * 8)
    assert mp2.load().mp2g_witness_program_run_dev(h, ctx.h, d_in.ptr, B, d_w.ptr, d_pr.ptr) == 0, mp2.load().mp2g_last_error()
    wires, got = d_w.download((B, 135, n)), d_pr.download((B, probe.size))
    for k, case in enumerate(spec["cases"]):
        assert fnv(wires[k]) == case["wires_fnv1a"] and [int(x) for x in got[k]] == case["probe"]
    mp2.load().mp2g_witness_program_free(h)
