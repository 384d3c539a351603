"""The public witness-tape format on the GPU: the device replay (mp2g_witness_program_run_dev) of the leaf circuits' user-logic
opcodes against the host replay and the builder, proofs of those witnesses with the witness check on, verified by the oracle; and a
plain-C client that hand-assembles a tape from include/mp2g.h alone (examples/c_witness_tape.c)."""
import importlib
import os
import re
import subprocess

import numpy as np
import pytest

import circuits as C
import oracle as O
from test_witness_tape import leaf_logic_circuit, leaf_logic_inputs

pytestmark = pytest.mark.gpu
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
FW = importlib.import_module("mapreduce-plonky2_amd.framework")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fnv(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def test_leaf_gate_opcodes_on_the_device(ctx, mp2):
    """U32Arithmetic / U32Subtraction / U32AddMany / U32RangeCheck / Comparison / BaseSplit<4> / MulExtension / Exponentiation as tape
    instructions: the device replay of a batch equals the host replay and the builder word for word, prove() accepts the witnesses
    (device-side witness check: every gate and copy constraint on H) and the oracle verifies the proofs; a non-u32 input makes
    prove() refuse that proof of the batch and only that one"""
    ins = [leaf_logic_inputs(s) for s in (11, 12, 13, 14)]
    ins[3][1] = ins[3][0]
    ckts = [leaf_logic_circuit(v) for v in ins]
    ck = ckts[0]
    prog = mp2.WitnessProgram(ck)
    B, n = len(ins), 1 << ck.log_n
    a = np.array(ins, dtype=np.uint64)
    want_w, want_h, want_pi = prog.run(a)
    d_in, d_w, d_pr = ctx.to_device(a), ctx.alloc(B * 135 * n * 8), ctx.alloc(B * prog.probe.size * 8)
    prog.run_dev(ctx, d_in, B, d_w, d_pr)
    got_w, got_pr = d_w.download((B, 135, n)), d_pr.download((B, prog.probe.size))
    for k in range(B):
        assert np.array_equal(got_w[k], ckts[k].wires), f"device replay != builder (proof {k})"
    assert np.array_equal(got_w, want_w) and np.array_equal(got_pr[:, :4], want_h) and np.array_equal(got_pr[:, 4:], want_pi)
    cp = FW.CircuitProver(ctx, ck, B, witness_check=True, pow_bits=8, num_queries=6)
    d_hash = ctx.to_device(np.ascontiguousarray(got_pr[:, :4]))
    cp.prove(d_w, d_hash)
    assert cp.pr.witness_status().tolist() == [0] * B
    caps, openings, proofs = cp.results()
    fp = C.oracle_params(ck, pow_bits=8, num_queries=6)
    for k in range(B):
        assert C.verify(ck, fp, cp.circuit_digest, got_pr[k, :4], caps[k], openings[k], proofs[k]) == 0, f"the oracle rejects proof {k}"
    oc, oo, op, _ = C.prove_witness(ck, fp, cp.circuit_digest, got_w[0], got_pr[0, :4])
    assert np.array_equal(caps[0], oc) and np.array_equal(openings[0], oo) and np.array_equal(proofs[0], op), "GPU proof != the oracle's proof of the same witness"
    bad = a.copy()
    bad[2, 0] = 1 << 33  # not a u32: the limbs of the products / sums no longer recompose
    d_in.upload(bad)
    prog.run_dev(ctx, d_in, B, d_w, d_pr)
    d_hash.upload(np.ascontiguousarray(d_pr.download((B, prog.probe.size))[:, :4]))
    cp.prove(d_w, d_hash)
    with pytest.raises(mp2.Mp2gError) as ei:
        cp.pr.witness_status()
    assert ei.value.flags[2] != 0 and [int(ei.value.flags[k]) for k in (0, 1, 3)] == [0, 0, 0]
    cp.free()
    prog.free()


def build_c_witness_tape():
    exe = os.path.join(ROOT, "examples", "c_witness_tape")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_witness_tape.c"), "-L" + os.path.join(ROOT, "mapreduce-plonky2_amd"),
                           "-lmp2gpu", "-Wl,-rpath," + os.path.join(ROOT, "mapreduce-plonky2_amd"), "-o", exe])
    return exe


def tape_demo_circuit(a, b_, c):
    """the circuit examples/c_witness_tape.c states in its header: y = a b + c; h = permutation(a, b, c, y, 0 x 8); the bits of c
    (c < 2^20); public inputs y, h0..h3, bit0"""
    b = R.Builder()
    ta, tb, tc = b.add_virtual(a), b.add_virtual(b_), b.add_virtual(c)
    y = b.mul_add(ta, tb, tc)
    z = b.zero()
    h = b.permute([ta, tb, tc, y] + [z] * 8)
    bits = b.split_le_base2(tc, 20)
    b.register_public_inputs([y] + h[:4] + [bits[0]])
    return b.build()


def test_c_client_hand_assembles_a_witness_tape(ctx, mp2, tmp_path):
    """examples/c_witness_tape.c writes its tape word by word from enum mp2g_witness_op (one ArithmeticGate operation, two
    Poseidon2Gate rows, one BaseSumGate split, the PublicInputGate and ConstantGate wires; its own slot numbering), replays it on
    the device for three proofs, proves them with the witness check on. The wires equal the Python builder's for the same circuit,
    the proofs equal the Python host's proofs and pass the oracle's verifier; a tape that misstates the row's gate constants is
    refused by prove()."""
    exe = build_c_witness_tape()
    rng = np.random.default_rng(5)
    ins = [[int(x) for x in O.rand_field(2, 50 + k)] + [int(rng.integers(0, 1 << 20))] for k in range(3)]
    ckts = [tape_demo_circuit(*v) for v in ins]
    ck = ckts[0]
    # the layout the C file restates
    assert ck.log_n == 6 and [ck.gates[i].kind for i in ck.instances[:7]] == [C.ARITHMETIC, C.POSEIDON2, C.BASE_SUM, C.POSEIDON2, C.PUBLIC_INPUT, C.CONSTANT, C.NOOP]
    assert ck.pi_row == 4 and [int(ck.pre[ck.num_constants - 2 + k, 0]) for k in range(2)] == [1, 1]  # row 0's gate constants
    pow_bits, queries = 8, 6
    cp = FW.CircuitProver(ctx, ck, 1, witness_check=True, pow_bits=pow_bits, num_queries=queries)
    path, out_path = str(tmp_path / "tape_demo.bin"), str(tmp_path / "tape_demo_out.bin")
    with open(path, "wb") as f:
        f.write(np.array([ck.log_n, ck.num_constants, len(ck.gates), ck.num_selectors, pow_bits, queries, len(ins)], dtype=np.uint32).tobytes())
        f.write(bytes(ck.gate_array))
        f.write(O.arr(cp.circuit_digest).tobytes() + O.arr(ck.pre).tobytes() + np.array(ins, dtype=np.uint64).tobytes())
    out = subprocess.run([exe, path, out_path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("proof ")]
    assert len(lines) == 3 and re.search(r"tape_words=\d+ levels=(\d+)", out.stdout)
    fp = C.oracle_params(ck, pow_bits=pow_bits, num_queries=queries)
    capw, n_open, pw = 4 << fp.cap_height, int(O.lib().orc_n_openings(__import__("ctypes").byref(fp))), int(O.lib().orc_fri_proof_words(__import__("ctypes").byref(fp)))
    blob = np.fromfile(out_path, dtype=np.uint64)
    per = 10 + 4 * capw + 2 * n_open + pw
    assert blob.size == 3 * per
    for k, (v, c) in enumerate(zip(ins, ckts)):
        rec = blob[k * per:(k + 1) * per]
        probe, caps, openings, proof = rec[:10], rec[10:10 + 4 * capw].reshape(4, capw), rec[10 + 4 * capw:10 + 4 * capw + 2 * n_open].reshape(n_open, 2), rec[10 + 4 * capw + 2 * n_open:]
        assert f"wires_fnv1a={fnv(c.wires)}" in lines[k], "the C tape's wires differ from the builder's"
        assert np.array_equal(probe[:4], c.pi_hash) and np.array_equal(probe[4:], c.public_inputs)
        assert int(probe[4]) == (v[0] * v[1] + v[2]) % O.P and int(probe[9]) == v[2] & 1
        assert C.verify(ck, fp, cp.circuit_digest, c.pi_hash, caps, openings, proof) == 0, f"the oracle rejects the C client's proof {k}"
        cp.prove(ctx.to_device(c.wires[None]), ctx.to_device(c.pi_hash[None]))
        pc, po, pp = cp.results()
        assert np.array_equal(pc[0], caps) and np.array_equal(po[0], openings) and np.array_equal(pp[0], proof), "C client's proof != the Python host's"
    bad = subprocess.run([exe, path, out_path, "bad"], capture_output=True, text=True, timeout=300)
    assert bad.returncode == 3 and "refused the witness" in bad.stdout, bad.stdout + bad.stderr
    cp.free()
