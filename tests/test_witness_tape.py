"""The public witness-tape format (include/mp2g.h enum mp2g_witness_op) on the CPU: the opcodes of the leaf circuits' user-logic
gates -- the generators of mp2-common/src/serialization/circuit_data_serialization.rs:186-231 that the recursion circuits do not
need (U32Arithmetic / U32Subtraction / U32AddMany / U32RangeCheck / Comparison / BaseSplit<4> / MulExtension / Exponentiation) --
replayed by the library's host executor against (a) the builder's own eager values and (b) the ORACLE's gate evaluators: every
constraint of every gate vanishes on the replayed wires, so generator and evaluator agree on each gate's wire layout. The device
replay of the same tapes is tests/test_gpu_witness_tape.py."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest

import circuits as C
import oracle as O

R = importlib.import_module("mapreduce-plonky2_amd.recursion")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def leaf_logic_circuit(vals):
    """one operation (or a few) of every new opcode, chained so that later gates read earlier gates' outputs"""
    b = R.Builder()
    x = [b.add_virtual(int(v)) for v in vals]
    lo, hi = b.u32_arithmetic(x[0], x[1], x[2])
    lo2, hi2 = b.u32_arithmetic(x[3], x[3], x[3])          # u32::MAX^2 + u32::MAX: high = u32::MAX, low = 0 (the canonicity edge)
    lo3, hi3 = b.u32_arithmetic(lo, hi, lo2)               # third operation of the row, fed by the first two
    lo4, _ = b.u32_arithmetic(lo3, x[0], hi3)              # a second U32Arithmetic row
    r, bo = b.u32_sub(x[0], x[1], b.zero())
    r2, bo2 = b.u32_sub(x[1], x[0], bo)                    # one of the two borrows
    s, co = b.u32_add_many([x[0], x[1], x[2]], bo2)
    s2, co2 = b.u32_add_many([s, r, r2], co)
    for t in (s, r, r2, s2, lo4):
        b.u32_range_check(t)
    le = b.comparison_le(x[0], x[1])
    le_eq = b.comparison_le(x[1], x[1])
    le_rev = b.comparison_le(x[1], x[0])
    limbs = b.base_split(x[2], 2, 20)                      # BaseSumGate<4>, 20 limbs: the reference's leaf gate set
    m = b.mul_ext_gate(3, R.E(x[4], x[5]), R.E(x[5], x[0]))
    m2 = b.mul_ext_gate(3, m, m)
    bits = b.split_le_base2(x[6], 10)
    ex = b.exponentiation(x[4], bits)
    b.register_public_inputs([lo, hi, lo2, hi2, lo4, r, bo, r2, bo2, s2, co2, le, le_eq, le_rev, limbs[3], m2.a, m2.b, ex])
    return b.build()


def leaf_logic_inputs(seed):
    rng = np.random.default_rng(seed)
    return [int(rng.integers(0, 1 << 32)) for _ in range(3)] + [0xFFFFFFFF] + [int(v) for v in O.rand_field(2, seed)] + [int(rng.integers(0, 1 << 10))]


def test_leaf_gate_opcodes_replay_the_generators():
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    ins = [leaf_logic_inputs(s) for s in (1, 2, 3)]
    ins[2][0], ins[2][1] = ins[2][1], ins[2][1]            # equal operands: borrow 0, comparison true both ways
    ckts = [leaf_logic_circuit(v) for v in ins]
    assert all(np.array_equal(c.tape, ckts[0].tape) and np.array_equal(c.pre, ckts[0].pre) for c in ckts)  # structure only
    kinds = {g.kind for g in ckts[0].gates}
    assert {C.U32_ARITHMETIC, C.U32_SUBTRACTION, C.U32_ADD_MANY, C.U32_RANGE_CHECK, C.COMPARISON, C.BASE_SUM, C.MUL_EXT, C.EXPONENTIATION} <= kinds
    ops = {op for _, op in R.tape_instructions(ckts[0].tape)}
    assert {R.OP_U32_ARITH, R.OP_U32_SUB, R.OP_U32_ADD_MANY, R.OP_U32_RANGE_CHECK, R.OP_COMPARISON, R.OP_BASE_SPLIT, R.OP_MUL_EXT, R.OP_EXP} <= ops
    prog = mp2.WitnessProgram(ckts[0])
    wires, pi_hash, pis = prog.run(np.array(ins, dtype=np.uint64))
    for k, c in enumerate(ckts):
        assert np.array_equal(wires[k], c.wires), f"host replay != builder (inputs {k})"
        assert np.array_equal(pis[k], c.public_inputs) and np.array_equal(pi_hash[k], c.pi_hash)
        assert not C.eval_on_points(c, c.pre[:c.num_constants], wires[k]).any(), "a gate constraint does not vanish on the replayed wires"
    a = ins[0]
    want = (a[0] * a[1] + a[2])
    assert int(pis[0][0]) == want & 0xFFFFFFFF and int(pis[0][1]) == want >> 32 and int(pis[0][2]) == 0 and int(pis[0][3]) == 0xFFFFFFFF
    assert int(pis[0][11]) == (1 if a[0] <= a[1] else 0) and int(pis[0][12]) == 1 and int(pis[0][13]) == (1 if a[1] <= a[0] else 0)
    assert [int(pis[2][k]) for k in (11, 12, 13)] == [1, 1, 1] and int(pis[2][6]) == 0
    assert int(pis[0][17]) == pow(a[4], a[6], O.P)
    # the level schedule: the chained operations sit on increasing levels, the tape is in single-assignment form
    assert prog.n_levels >= 6
    # a corrupted witness (an input that is not a u32) violates the range decomposition: the oracle's evaluator sees it
    bad_in = list(a)
    bad_in[0] = 1 << 33
    w_bad, _, _ = prog.run(np.array([bad_in], dtype=np.uint64))
    assert C.eval_on_points(ckts[0], ckts[0].pre[:ckts[0].num_constants], w_bad[0]).any()
    prog.free()


def test_tape_validation_of_the_leaf_gate_opcodes():
    """mp2g_witness_program_create refuses what the header says it refuses: operation indices beyond the gate's count, gate
    parameters whose wires do not fit 135 columns, counts that run past the tape, slots beyond n_slots"""
    mp2 = importlib.import_module("mapreduce-plonky2_amd")

    def create(tape, n_slots=64, log_n=3):
        h = ctypes.c_void_p()
        t = np.ascontiguousarray(tape, dtype=np.uint64)
        ins = np.arange(4, dtype=np.uint32)
        rc = mp2.load().mp2g_witness_program_create(t.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(t.size), n_slots, log_n,
                                                    ins.ctypes.data_as(ctypes.c_void_p), 4, None, 0, ctypes.byref(h))
        if rc == 0:
            mp2.load().mp2g_witness_program_free(h)
            return None
        return mp2.load().mp2g_last_error().decode() if isinstance(mp2.load().mp2g_last_error(), bytes) else "refused"

    ok = [R.OP_U32_ARITH, 0, 2, 3, 0, 1, 2, 10, 11]
    assert create(ok) is None
    assert create([R.OP_U32_ARITH, 0, 3, 3, 0, 1, 2, 10, 11]) is not None          # operation 3 of 3
    assert create([R.OP_U32_ARITH, 0, 0, 4, 0, 1, 2, 10, 11]) is not None          # 4 operations do not fit a row
    assert create([R.OP_U32_ARITH, 8, 0, 3, 0, 1, 2, 10, 11]) is not None          # row 8 of 8
    assert create([R.OP_U32_ARITH, 0, 0, 3, 0, 1, 2, 10, 64]) is not None          # slot 64 of 64
    assert create(ok[:-1]) is not None                                               # truncated
    assert create([R.OP_U32_SUB, 1, 5, 6, 0, 1, 2, 10, 11]) is None and create([R.OP_U32_SUB, 1, 6, 6, 0, 1, 2, 10, 11]) is not None
    assert create([R.OP_U32_ADD_MANY, 2, 4, 5, 3, 0, 1, 2, 3, 10, 11]) is None
    assert create([R.OP_U32_ADD_MANY, 2, 4, 5, 17, 0, 1, 2, 3, 10, 11]) is not None  # 17 addends
    assert create([R.OP_U32_ADD_MANY, 2, 0, 6, 3, 0, 1, 2, 3, 10, 11]) is not None   # (3 + 3 + 18) x 6 wires > 135
    assert create([R.OP_U32_ADD_MANY, 2, 0, 5, 3, 0, 1]) is not None                 # the addend count runs past the tape
    assert create([R.OP_U32_RANGE_CHECK, 0, 6, 7, 1]) is None and create([R.OP_U32_RANGE_CHECK, 0, 7, 7, 1]) is not None
    assert create([R.OP_COMPARISON, 0, 32, 16, 0, 1, 10]) is None
    assert create([R.OP_COMPARISON, 0, 64, 16, 0, 1, 10]) is not None and create([R.OP_COMPARISON, 0, 32, 17, 0, 1, 10]) is not None
    assert create([R.OP_BASE_SPLIT, 0, 2, 20, 1] + list(range(10, 30))) is None
    assert create([R.OP_BASE_SPLIT, 0, 3, 20, 1] + list(range(10, 30))) is not None  # base 8
    assert create([R.OP_BASE_SPLIT, 0, 2, 32, 1] + list(range(10, 42))) is not None  # 64 bits of limbs
    assert create([R.OP_MUL_EXT, 0, 12, 5, 0, 1, 2, 3, 10, 11]) is None and create([R.OP_MUL_EXT, 0, 13, 5, 0, 1, 2, 3, 10, 11]) is not None
    assert create([R.OP_MUL_EXT, 0, 0, O.P, 0, 1, 2, 3, 10, 11]) is not None         # non-canonical constant
    assert create([R.OP_EXP, 0, 3, 0, 1, 2, 3, 10]) is None and create([R.OP_EXP, 0, 67, 0] + [1] * 67 + [10]) is not None
    assert create([24, 0, 0]) is not None and create([0]) is not None                # no such opcode (MP2G_OP_END = 24)


def test_public_header_and_python_agree_on_the_opcodes():
    """the numbers of enum mp2g_witness_op in include/mp2g.h are the OP_* of recursion.py (the Python host is one client of the
    public format, not its definition), and csrc/witness.h takes them from the header"""
    text = open(os.path.join(ROOT, "include", "mp2g.h")).read()
    body = text[text.index("enum mp2g_witness_op {"):]
    body = body[:body.index("};")]
    public = {m.group(1): int(m.group(2)) for m in re.finditer(r"MP2G_(OP_[A-Z0-9_]+) = (\d+)", body)}
    assert len(public) == 24 and public.pop("OP_END") == 24
    mine = {k: int(v) for k, v in vars(R).items() if k.startswith("OP_")}
    assert mine == public
    internal = open(os.path.join(ROOT, "mapreduce-plonky2_amd", "csrc", "witness.h")).read()
    for name in public:
        assert f"{name} = MP2G_{name}" in internal


def test_random_tapes_are_refused_or_replayed_without_harm():
    """the tape indexes host memory (slot table, wire matrix): whatever words a caller hands to mp2g_witness_program_create, the library
    either refuses the tape or replays it inside its buffers. 400 random tapes -- well-formed instructions with operands drawn around
    their limits (rows, columns, counts, slots one past the end), some truncated -- each created and, if accepted, replayed on the
    host for two random input vectors into a guarded wire buffer"""
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    lib = mp2.load()
    rng = np.random.default_rng(0xC0FFEE07)
    log_n, n_slots, n = 3, 40, 8
    P = O.P

    def slot():
        return int(rng.integers(0, n_slots + (1 if rng.random() < 0.05 else 0)))

    def row():
        return int(rng.integers(0, n + (1 if rng.random() < 0.05 else 0)))

    def instr():
        op = int(rng.integers(1, 25))
        sl = lambda k: [slot() for _ in range(k)]
        if op == R.OP_ARITH: return [op, row(), int(rng.integers(0, 21)), int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64))] + sl(4)
        if op == R.OP_ARITH_EXT: return [op, row(), int(rng.integers(0, 11)), 1, 2] + sl(8)
        if op in (R.OP_P2, R.OP_POSEIDON): return [op, row()] + sl(25)
        if op == R.OP_BASE_SUM: return [op, row()] + sl(64)
        if op == R.OP_RA: return [op, row(), int(rng.integers(0, 5))] + sl(18)
        if op == R.OP_REDUCING: return [op, row()] + sl(4 + 43 + 2)
        if op == R.OP_REDUCING_EXT: return [op, row()] + sl(4 + 64 + 2)
        if op == R.OP_COSET:
            bits = int(rng.integers(1, 7))
            return [op, row(), bits] + sl(1 + (2 << min(bits, 5)) + 4)
        if op == R.OP_WIRE: return [op, row(), int(rng.integers(0, 137))] + sl(1)
        if op == R.OP_HINT_DIV_EXT: return [op] + sl(6)
        if op in (R.OP_HINT_LO63, R.OP_HINT_HI): return [op] + sl(2)
        if op == R.OP_HINT_SPLIT: return [op, slot(), int(rng.integers(0, 66))] + sl(2)
        if op == R.OP_PAR: return [op, 1, 4, R.OP_HINT_HI, slot(), slot()][:int(rng.integers(3, 7))]
        if op == R.OP_U32_ARITH: return [op, row(), int(rng.integers(0, 4)), int(rng.integers(0, 5))] + sl(5)
        if op == R.OP_U32_SUB: return [op, row(), int(rng.integers(0, 7)), int(rng.integers(0, 8))] + sl(5)
        if op == R.OP_U32_ADD_MANY:
            na = int(rng.integers(0, 19))
            return [op, row(), int(rng.integers(0, 6)), int(rng.integers(0, 7)), na] + sl(min(na, 17) + 3)
        if op == R.OP_U32_RANGE_CHECK: return [op, row(), int(rng.integers(0, 8)), int(rng.integers(0, 9))] + sl(1)
        if op == R.OP_COMPARISON: return [op, row(), int(rng.integers(0, 66)), int(rng.integers(0, 19))] + sl(3)
        if op == R.OP_BASE_SPLIT:
            nl = int(rng.integers(0, 66))
            return [op, row(), int(rng.integers(0, 4)), nl] + sl(1 + min(nl, 64))
        if op == R.OP_MUL_EXT: return [op, row(), int(rng.integers(0, 15)), int(rng.integers(0, P, dtype=np.uint64))] + sl(6)
        if op == R.OP_EXP:
            nb = int(rng.integers(0, 69))
            return [op, row(), nb] + sl(1 + min(nb, 67) + 1)
        return [op, 0, 0]  # MP2G_OP_END and beyond: no such opcode

    accepted = 0
    ins = np.arange(4, dtype=np.uint32)
    for _ in range(400):
        tape = []
        for _ in range(int(rng.integers(1, 6))):
            tape += instr()
        if rng.random() < 0.1:
            tape = tape[:-1]
        t = np.ascontiguousarray(tape, dtype=np.uint64)
        h = ctypes.c_void_p()
        rc = lib.mp2g_witness_program_create(t.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(t.size), n_slots, log_n,
                                             ins.ctypes.data_as(ctypes.c_void_p), 4, None, 0, ctypes.byref(h))
        if rc:
            continue
        accepted += 1
        guard = 64
        wires = np.full(2 * 135 * n + 2 * guard, 0xDEADBEEFDEADBEEF, dtype=np.uint64)
        inputs = np.ascontiguousarray(O.rand_field(8, int(rng.integers(1, 1 << 30))).reshape(2, 4))
        rc = lib.mp2g_witness_program_run(h, inputs.ctypes.data_as(ctypes.c_void_p), 2, 2, ctypes.c_void_p(wires.ctypes.data + 8 * guard), None, 0, None)
        assert rc == 0
        assert (wires[:guard] == 0xDEADBEEFDEADBEEF).all() and (wires[-guard:] == 0xDEADBEEFDEADBEEF).all(), "a replay wrote outside the wire matrix"
        lib.mp2g_witness_program_free(h)
    assert 20 <= accepted <= 380, accepted  # both outcomes occur
