#!/usr/bin/env python3
"""Generate tests/golden/witness_tape_vectors.json: the PUBLIC witness-tape format (include/mp2g.h enum mp2g_witness_op) frozen as data.

Two tapes with their inputs and what their replay must produce: (1) the hand-written tape of examples/c_witness_tape.c, word for
word as the C file assembles it (an ArithmeticGate operation, two Poseidon2Gate rows, a BaseSumGate split, PublicInputGate and
ConstantGate wires); (2) the tape recursion.Builder records for a circuit that uses every round-6 opcode (U32Arithmetic /
Subtraction / AddMany / RangeCheck, Comparison, BaseSplit, MulExtension, Exponentiation; tests/test_witness_tape.py
leaf_logic_circuit). Expected values come from the Python builder's eager evaluation (wire matrix FNV-1a, public-inputs hash,
public inputs) -- NOT from the library's replay -- and every gate constraint was checked to vanish on those wires by the oracle's
evaluators when the file was made. A change of an opcode's number, operand order, wire layout or arithmetic breaks the test that
replays these tapes (tests/test_golden_vectors.py), whatever the builder does by then."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import circuits as C  # noqa: E402
import oracle as O  # noqa: E402
from test_witness_tape import leaf_logic_circuit, leaf_logic_inputs  # noqa: E402

R = importlib.import_module("mapreduce-plonky2_amd.recursion")


def fnv(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def demo_circuit(a, b_, c):
    b = R.Builder()
    ta, tb, tc = b.add_virtual(a), b.add_virtual(b_), b.add_virtual(c)
    y = b.mul_add(ta, tb, tc)
    z = b.zero()
    h = b.permute([ta, tb, tc, y] + [z] * 8)
    bits = b.split_le_base2(tc, 20)
    b.register_public_inputs([y] + h[:4] + [bits[0]])
    return b.build()


def c_demo_tape():
    """examples/c_witness_tape.c, word for word (opcodes by NUMBER: this file must not follow a renumbering)"""
    S_ZERO, S_A, S_B, S_C, S_Y, S_H, S_BIT, S_PIH, N_SLOTS = 0, 1, 2, 3, 4, 5, 17, 80, 92
    t = [1, 0, 0, 1, 1, S_A, S_B, S_C, S_Y]
    t += [3, 1, S_A, S_B, S_C, S_Y] + [S_ZERO] * 8 + [S_ZERO] + [S_H + i for i in range(12)]
    t += [4, 2, S_C] + [S_BIT + i for i in range(63)]
    t += [3, 3, S_Y, S_H, S_H + 1, S_H + 2, S_H + 3, S_BIT] + [S_ZERO] * 6 + [S_ZERO] + [S_PIH + i for i in range(12)]
    for i in range(4):
        t += [9, 4, i, S_PIH + i]
    t += [9, 5, 0, S_ZERO]
    return {"tape": t, "n_slots": N_SLOTS, "log_n": 6, "input_sids": [S_A, S_B, S_C], "const_slots": [[S_ZERO, 0]],
            "probe": [S_PIH, S_PIH + 1, S_PIH + 2, S_PIH + 3, S_Y, S_H, S_H + 1, S_H + 2, S_H + 3, S_BIT]}


def main():
    out = {"_generator": "tools/gen_golden_tape.py (expected values: the Python builder's eager evaluation; constraints checked by the oracle's gate evaluators)"}
    demo = c_demo_tape()
    cases = []
    rng = np.random.default_rng(6)
    for k in range(3):
        v = [int(x) for x in O.rand_field(2, 600 + k)] + [int(rng.integers(0, 1 << 20))]
        ck = demo_circuit(*v)
        assert not C.eval_on_points(ck, ck.pre[:ck.num_constants], ck.wires).any()
        cases.append({"inputs": v, "wires_fnv1a": fnv(ck.wires), "probe": [int(x) for x in ck.pi_hash] + [int(x) for x in ck.public_inputs]})
    demo["cases"] = cases
    out["c_witness_tape_demo"] = demo
    ins = [leaf_logic_inputs(s) for s in (21, 22, 23)]
    ins[2][1] = ins[2][0]
    ckts = [leaf_logic_circuit(v) for v in ins]
    ck = ckts[0]
    for c in ckts:
        assert np.array_equal(c.tape, ck.tape) and not C.eval_on_points(c, c.pre[:c.num_constants], c.wires).any()
    out["leaf_gate_opcodes"] = {"tape": [int(x) for x in ck.tape], "n_slots": int(ck.n_slots), "log_n": int(ck.log_n), "input_sids": [int(x) for x in ck.input_sids],
                                "const_slots": [[int(a), int(b)] for a, b in ck.const_slots], "probe": [int(x) for x in ck.pi_hash_sids] + [int(x) for x in ck.public_input_sids],
                                "opcodes_used": sorted({int(op) for _, op in R.tape_instructions(ck.tape)}),
                                "cases": [{"inputs": v, "wires_fnv1a": fnv(c.wires), "probe": [int(x) for x in c.pi_hash] + [int(x) for x in c.public_inputs]} for v, c in zip(ins, ckts)]}
    path = os.path.join(ROOT, "tests", "golden", "witness_tape_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes; opcodes", out["leaf_gate_opcodes"]["opcodes_used"])


if __name__ == "__main__":
    main()
