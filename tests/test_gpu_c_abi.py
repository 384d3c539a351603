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


def test_header_compiles_as_c():
    """No GPU needed: the header is valid C11 and the demos link against the library."""
    build_demo()
    build_prove_circuit()


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
