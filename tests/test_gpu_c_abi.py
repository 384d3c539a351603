"""include/mp2g.h is usable from plain C: build examples/c_abi_demo.c with gcc, run it on the GPU and
compare the proof it serializes with the one the Python harness gets for the same inputs."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_demo():
    exe = os.path.join(ROOT, "examples", "c_abi_demo")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + os.path.join(ROOT, "mapreduce-plonky2_amd"),
                           "-lmp2gpu", "-Wl,-rpath," + os.path.join(ROOT, "mapreduce-plonky2_amd"), "-o", exe])
    return exe


def build_prove_circuit():
    exe = os.path.join(ROOT, "examples", "c_prove_circuit")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_prove_circuit.c"), "-L" + os.path.join(ROOT, "mapreduce-plonky2_amd"),
                           "-lmp2gpu", "-Wl,-rpath," + os.path.join(ROOT, "mapreduce-plonky2_amd"), "-o", exe])
    return exe


def build_generate_proof():
    exe = os.path.join(ROOT, "examples", "c_generate_proof")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_generate_proof.c"), "-L" + os.path.join(ROOT, "mapreduce-plonky2_amd"),
                           "-lmp2gpu", "-Wl,-rpath," + os.path.join(ROOT, "mapreduce-plonky2_amd"), "-o", exe])
    return exe


def build_forest():
    exe = os.path.join(ROOT, "examples", "c_forest")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_forest.c"), "-L" + os.path.join(ROOT, "mapreduce-plonky2_amd"),
                           "-lmp2gpu", "-Wl,-rpath," + os.path.join(ROOT, "mapreduce-plonky2_amd"), "-o", exe])
    return exe


def test_header_compiles_as_c():
    """No GPU needed: the header is valid C11 and the demos link against the library."""
    build_demo()
    build_prove_circuit()
    build_generate_proof()
    build_forest()
    # the hand-assembled witness tape (tests/test_gpu_witness_tape.py runs it): uses enum mp2g_witness_op from the header alone
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_witness_tape.c"), "-L" + os.path.join(ROOT, "mapreduce-plonky2_amd"),
                           "-lmp2gpu", "-Wl,-rpath," + os.path.join(ROOT, "mapreduce-plonky2_amd"), "-o", os.path.join(ROOT, "examples", "c_witness_tape")])


@pytest.mark.gpu
def test_c_client_matches_python(ctx, mp2):
    exe = build_demo()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"proof_words=(\d+) bytes=(\d+) fnv1a=([0-9a-f]+) pow_witness=(\d+)", out.stdout)
    assert m, out.stdout
    ws = (5, 9, 4, 3)
    fp = mp2.standard_recursion_params(6, ws, pow_bits=6, num_queries=4)
    vals = [O.rand_field((w, 64), 100 + i) for i, w in enumerate(ws)]
    caps, openings, proof = mp2.pcs_prove(ctx, fp, vals, O.rand_field(4, 1), O.rand_field(4, 2))
    data = mp2.serialize_proof(fp, 2, caps, openings, proof, [7, 8, 9])
    h = 1469598103934665603
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert int(m.group(1)) == fp.proof_words and int(m.group(2)) == len(data)
    assert int(m.group(4)) == int(proof[-1])
    assert m.group(3) == f"{h:016x}"


def fnv(data):
    h = 1469598103934665603
    for b in bytes(data):
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


@pytest.mark.gpu
def test_c_client_proves_a_gate_level_circuit(ctx, mp2, tmp_path):
    """examples/c_prove_circuit.c: CircuitData + witness in, proof bytes out, through the C ABI alone; same proof
    as the Python harness; an unsatisfied witness is reported the way prove() fails in the reference."""
    import circuits as C
    exe = build_prove_circuit()
    log_n, pow_bits, queries = 6, 5, 3
    ckt = C.build(log_n, C.LEAF_KINDS, 41)
    cd = O.rand_field(4, 6)

    def write(path, wires):
        with open(path, "wb") as f:
            f.write(np.array([log_n, ckt.num_constants, C.NUM_ROUTED, C.NUM_WIRES, len(ckt.gates), ckt.num_selectors, pow_bits, queries],
                             dtype=np.uint32).tobytes())
            f.write(bytes(ckt.gate_array))
            f.write(O.arr(ckt.pi_hash).tobytes() + O.arr(cd).tobytes() + O.arr(ckt.pre).tobytes() + O.arr(wires).tobytes())

    good, bad = str(tmp_path / "good.bin"), str(tmp_path / "bad.bin")
    write(good, ckt.wires)
    out = subprocess.run([exe, good], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "witness_flags=0" in out.stdout
    m = re.search(r"proof_words=(\d+) bytes=(\d+) proof_fnv1a=([0-9a-f]+) wire_fnv1a=([0-9a-f]+)", out.stdout)
    assert m, out.stdout
    fp = mp2.standard_recursion_params(log_n, (ckt.num_constants + C.NUM_ROUTED, C.NUM_WIRES, 20, 16), pow_bits=pow_bits, num_queries=queries)
    pr = mp2.BatchedProver(ctx, fp, 1)
    pr.set_preprocessed(ctx.to_device(ckt.pre))
    pr.enable_permutation(C.NUM_ROUTED, 8)
    pr.enable_quotient()
    pr.set_gates([mp2.Gate(g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end) for g in ckt.gates], ckt.num_selectors)
    pr.prove([ctx.to_device(ckt.wires[None]), None, None], ctx.to_device(cd), ctx.to_device(ckt.pi_hash[None]))
    caps, openings, proofs = pr.results()
    data = mp2.serialize_proof(fp, ckt.num_constants, caps[0], openings[0], proofs[0], ckt.pi_hash)
    assert int(m.group(1)) == fp.proof_words and int(m.group(2)) == len(data)
    assert m.group(3) == fnv(proofs[0].tobytes()) and m.group(4) == fnv(data)
    pr.free()
    w = ckt.wires.copy()
    row = ckt.instances.index(next(i for i, g in enumerate(ckt.gates) if g.kind == C.U32_SUBTRACTION))
    w[3, row] ^= np.uint64(1)
    write(bad, w)
    out = subprocess.run([exe, bad], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3 and "violates a gate constraint" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_c_client_generates_framework_proofs(ctx, mp2, tmp_path):
    """examples/c_generate_proof.c: CircuitWithUniversalVerifier::generate_proof (witness generation on the device, base prove(), wrap
    chain) for a batch of nodes through the C ABI alone (mp2g_chain_*): a reduce circuit with two universal verifiers over two map
    proofs, three nodes at once. Same final proofs and public inputs as the Python host; a child proof with an altered public input
    makes generate_proof fail the way the reference's prove() panics."""
    import importlib
    R = importlib.import_module("mapreduce-plonky2_amd.recursion")
    FW = importlib.import_module("mapreduce-plonky2_amd.framework")
    exe = build_generate_proof()
    prover = FW.GpuProver(ctx)
    fw = R.RecursiveCircuits([R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)], prover, FW.circuit_fri_params)
    data = O.rand_field(16, 0xC0FFEE08)
    leaves = fw.generate_proofs_batch("map", [([], [], data[4 * i:4 * i + 4]) for i in range(4)])
    jobs = [([leaves[0], leaves[1]], ["map", "map"], None), ([leaves[2], leaves[3]], ["map", "map"], None), ([leaves[1], leaves[2]], ["map", "map"], None)]
    want = fw.generate_proofs_batch("reduce", jobs)
    vd = fw.vds["map"]
    rows = np.stack([np.concatenate([np.asarray(fw.set_digest, dtype=np.uint64)] + [R.universal_inputs(p, vd, fw.membership(vd[1])) for p in kids]) for kids, _, _ in jobs])

    def write(path, inputs):
        with open(path, "wb") as f:
            f.write(np.array([len(fw.chains["reduce"]), inputs.shape[0], inputs.shape[1]], dtype=np.uint32).tobytes())
            for step, (ckt, cap, digest) in enumerate(fw.chains["reduce"]):
                prog = fw.witness_programs("reduce")[step]
                fp = FW.circuit_fri_params(ckt)
                cs = np.ascontiguousarray(ckt.const_slots, dtype=np.uint64).reshape(-1, 2)
                tape = np.ascontiguousarray(ckt.tape, dtype=np.uint64)
                f.write(np.array([ckt.log_n, ckt.num_constants, len(ckt.gates), ckt.num_selectors, fp.pow_bits, fp.num_queries, ckt.n_slots, len(ckt.input_sids),
                                  cs.shape[0], prog.probe.size, tape.size & 0xFFFFFFFF, tape.size >> 32], dtype=np.uint32).tobytes())
                f.write(bytes(ckt.gate_array))
                f.write(O.arr(digest).tobytes() + O.arr(ckt.pre).tobytes() + tape.tobytes())
                f.write(np.ascontiguousarray(ckt.input_sids, dtype=np.uint32).tobytes() + cs.tobytes() + np.ascontiguousarray(prog.probe, dtype=np.uint32).tobytes())
            f.write(O.arr(inputs).tobytes())

    def fnv(a):
        h = 1469598103934665603
        for b in np.ascontiguousarray(a).tobytes():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return f"{h:016x}"

    good, bad = str(tmp_path / "reduce.bin"), str(tmp_path / "reduce_bad.bin")
    write(good, rows)
    out = subprocess.run([exe, good], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("node ")]
    assert len(lines) == 3
    for b, (line, pr) in enumerate(zip(lines, want)):
        assert f"proof_fnv1a={fnv(pr[2])}" in line and f"openings_fnv1a={fnv(pr[1])}" in line and f"caps_fnv1a={fnv(pr[0])}" in line and f"pis_fnv1a={fnv(pr[3])}" in line, (b, line)
    tampered = rows.copy()
    tampered[1, 4 + 64 + 4] ^= np.uint64(1)  # first public input of node 1's first child (after the set digest and the child's verifier data)
    write(bad, tampered)
    out = subprocess.run([exe, bad], capture_output=True, text=True, timeout=300)
    assert out.returncode == 3 and "invalid witness" in out.stdout, out.stdout + out.stderr
    prover.free()


@pytest.mark.gpu
def test_c_client_proves_a_forest(ctx, mp2, tmp_path):
    """examples/c_forest.c: the map / reduce tree of recursion-framework/tests/integration.rs over 8 leaves (8 map + 7 reduce framework
    proofs) through mp2g_forest_* from plain C -- circuits described once, nodes registered once, two waves of units, two worker threads
    inside the library, child proofs in the device pool. The root proof equals the Python host's word for word; a leaf whose recorded
    child is missing makes the call fail with the library's message."""
    import importlib
    R = importlib.import_module("mapreduce-plonky2_amd.recursion")
    FW = importlib.import_module("mapreduce-plonky2_amd.framework")
    exe = build_forest()
    prover = FW.GpuProver(ctx)
    fw = R.RecursiveCircuits([R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)], prover, FW.circuit_fri_params)
    n_leaves = 8
    data = O.rand_field(4 * n_leaves, 0xC0FFEE09)
    level = fw.generate_proofs_batch("map", [([], [], data[4 * i:4 * i + 4]) for i in range(n_leaves)])
    names = ["map"] * n_leaves
    while len(level) > 1:
        level = fw.generate_proofs_batch("reduce", [([level[2 * i], level[2 * i + 1]], [names[2 * i], names[2 * i + 1]], None) for i in range(len(level) // 2)])
        names = ["reduce"] * len(level)
    root = level[0]
    want_words = R.proof_inputs(root)
    set_digest = np.asarray(fw.set_digest, dtype=np.uint64)
    pw = want_words.size
    mem_len = 5 * max(0, (fw.set_size - 1).bit_length())

    def block(name):
        vd = fw.vds[name]
        bits, sib = fw.membership(vd[1])
        return np.concatenate([O.arr(vd[0]).ravel(), O.arr(vd[1]).ravel()]), np.concatenate([O.arr(bits).ravel(), O.arr(sib).ravel()]).astype(np.uint64)

    # node ids: leaves 100 + i; reduce nodes (level l, index i) = 1000 * l + i
    map_ids = np.arange(100, 100 + n_leaves, dtype=np.uint64)
    map_consts = np.stack([np.concatenate([set_digest, data[4 * i:4 * i + 4]]) for i in range(n_leaves)])
    red_ids, red_kids, red_consts = [], [], []
    prev_ids, prev_name, lvl = list(map_ids), "map", 1
    while len(prev_ids) > 1:
        cur = []
        head, tail = block(prev_name)
        for i in range(len(prev_ids) // 2):
            nid = 1000 * lvl + i
            cur.append(nid)
            red_ids.append(nid)
            red_kids.append([prev_ids[2 * i], prev_ids[2 * i + 1]])
            red_consts.append(np.concatenate([set_digest, head, tail, head, tail]))
        prev_ids, prev_name, lvl = cur, "reduce", lvl + 1
    root_id = prev_ids[0]
    n_in_map, n_in_red = fw.witness_programs("map")[0].n_inputs, fw.witness_programs("reduce")[0].n_inputs
    off0 = 4 + 68
    off1 = off0 + pw + mem_len + 68
    assert n_in_map == 8 and n_in_red == off1 + pw + mem_len

    def write(path, kids):
        with open(path, "wb") as f:
            f.write(np.array([2, 2, 4, pw, 64], dtype=np.uint32).tobytes())
            for name, desc in (("map", [n_in_map, 0, 0, 0, 0, 0, n_in_map]), ("reduce", [n_in_red, 2, off0, off1, 0, 0, n_in_red - 2 * pw])):
                f.write(np.array([len(fw.chains[name])], dtype=np.uint32).tobytes())
                for step, (ckt, cap, digest) in enumerate(fw.chains[name]):
                    prog = fw.witness_programs(name)[step]
                    fp = FW.circuit_fri_params(ckt)
                    cs = np.ascontiguousarray(ckt.const_slots, dtype=np.uint64).reshape(-1, 2)
                    tape = np.ascontiguousarray(ckt.tape, dtype=np.uint64)
                    f.write(np.array([ckt.log_n, ckt.num_constants, len(ckt.gates), ckt.num_selectors, fp.pow_bits, fp.num_queries, ckt.n_slots, len(ckt.input_sids),
                                      cs.shape[0], prog.probe.size, tape.size & 0xFFFFFFFF, tape.size >> 32], dtype=np.uint32).tobytes())
                    f.write(bytes(ckt.gate_array))
                    f.write(O.arr(digest).tobytes() + O.arr(ckt.pre).tobytes() + tape.tobytes())
                    f.write(np.ascontiguousarray(ckt.input_sids, dtype=np.uint32).tobytes() + cs.tobytes() + np.ascontiguousarray(prog.probe, dtype=np.uint32).tobytes())
                f.write(np.array(desc, dtype=np.uint32).tobytes())
            f.write(np.array([n_leaves], dtype=np.uint32).tobytes() + map_ids.tobytes() + O.arr(map_consts).tobytes())
            f.write(np.array([len(red_ids)], dtype=np.uint32).tobytes() + O.arr(red_ids).tobytes() + O.arr(kids).tobytes() + O.arr(np.stack(red_consts)).tobytes())
            # two waves: the two 4-leaf subtrees as independent units, then the root
            units = [[100, 101, 102, 103, 1000, 1001, 2000], [104, 105, 106, 107, 1002, 1003, 2001]]
            f.write(np.array([2, 2, 0, 7, 14], dtype=np.uint32).tobytes() + O.arr(units).tobytes())
            f.write(np.array([1, 0, 1], dtype=np.uint32).tobytes() + O.arr([root_id]).tobytes())
            f.write(O.arr([root_id]).tobytes())

    def fnv(a):
        h = 1469598103934665603
        for b in np.ascontiguousarray(a).tobytes():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return f"{h:016x}"

    good, bad = str(tmp_path / "forest.bin"), str(tmp_path / "forest_bad.bin")
    write(good, red_kids)
    out = subprocess.run([exe, good], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f"proved=15 root_words={pw} root_fnv1a={fnv(want_words)}" in out.stdout, out.stdout
    broken = [list(k) for k in red_kids]
    broken[0][1] = 999  # a child nobody registered
    write(bad, broken)
    out = subprocess.run([exe, bad], capture_output=True, text=True, timeout=600)
    assert out.returncode == 3 and "unknown child" in out.stdout, out.stdout + out.stderr
    prover.free()
