"""A small eager CircuitBuilder and plonky2's recursive verifier on top of it: the circuit logic of the
recursion framework's wrap step (recursion-framework/src/universal_verifier_gadget/wrap_circuit.rs:122-148:
`builder.verify_proof(&pt, &inner_data, cd)` with the inner circuit's verifier data as constants, the inner
proof's public inputs re-registered) and of its map / reduce circuits (tests/integration.rs:65-136), so that
base -> wrap -> verify runs end to end through the HIP prover with REAL witnesses: the wrap circuit's wire matrix
contains the inner proof and every constraint of [dep] plonky2 plonk/verifier.rs + fri/recursive_verifier.rs.

"Eager": every target carries its value, an operation computes its result while it places the gate (the slot
packing follows CircuitBuilder::find_slot: operations with equal gate constants share a row), and connect()
asserts equality on the spot -- a wrong witness or a wrong gadget fails at the line that produced it. The circuit
STRUCTURE (rows, gate constants, copy constraints) never depends on witness values, so building the same circuit
around another inner proof gives the same preprocessed polynomials (same circuit digest) and a new wire matrix.

Restated from memory of the published plonky2 sources like the rest of the [dep] behaviour:
parity unpinned; the external anchors are the oracle's verifier accepting the wrap proofs and the witness check
(every gate constraint of the wrap circuit vanishes on H).

Host-side Python (witness generation stays on the host in this back end); a 2^12..2^13-row verifier circuit
takes seconds to fill. Nothing here touches the GPU library or the CPU oracle.
"""
import numpy as np

from . import Gate, MULT_GEN
from . import circuits as C

P = C.P
W7 = 7  # quadratic extension X^2 = 7
TWO_GEN = 7277203076849721926
NUM_WIRES, NUM_ROUTED = C.NUM_WIRES, C.NUM_ROUTED
K2 = C.poseidon2_constants


def root_of_unity(bits):
    return pow(TWO_GEN, 1 << (32 - bits), P)


def inv(x):
    return pow(x % P, P - 2, P)


# ---- extension-field values (tuples) ---------------------------------------------------------------------------------
def xadd(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def xsub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def xmul(a, b):
    return ((a[0] * b[0] + W7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def xscale(a, k):
    return (a[0] * k % P, a[1] * k % P)


def xinv(a):
    n = inv((a[0] * a[0] - W7 * a[1] * a[1]) % P)
    return (a[0] * n % P, (P - a[1]) * n % P)


class T:
    """a base-field target: its value, (once it sits in a routed wire) its home cell, and its slot in the witness
    program's value table"""
    __slots__ = ("v", "cell", "sid")

    def __init__(self, v, cell=None, sid=-1):
        self.v, self.cell, self.sid = v % P, cell, sid


# Witness program ("tape"): the builder's operations in order, over value slots. The circuit structure does not depend
# on witness values, so the tape recorded while the structure is built IS the circuit's witness generator
# (plonky2's generate_partial_witness for this circuit): csrc/witness.hip replays it for a batch of input vectors.
# Every instruction: opcode, then its operands (row / slot indices, u64 constants), fixed length per opcode.
(OP_ARITH, OP_ARITH_EXT, OP_P2, OP_BASE_SUM, OP_RA, OP_REDUCING, OP_REDUCING_EXT, OP_COSET, OP_WIRE, OP_HINT_DIV_EXT,
 OP_HINT_LO63, OP_HINT_HI, OP_HINT_SPLIT, OP_PAR, OP_POSEIDON,
 # the leaf circuits' user-logic gates (round 6; include/mp2g.h enum mp2g_witness_op documents every operand layout)
 OP_U32_ARITH, OP_U32_SUB, OP_U32_ADD_MANY, OP_U32_RANGE_CHECK, OP_COMPARISON, OP_BASE_SPLIT, OP_MUL_EXT, OP_EXP) = range(1, 24)


class E:
    """an extension-field target"""
    __slots__ = ("a", "b")

    def __init__(self, a, b):
        self.a, self.b = a, b

    @property
    def v(self):
        return (self.a.v, self.b.v)


class Row:
    __slots__ = ("kind", "p0", "p1", "p2", "consts", "wires")

    def __init__(self, kind, p0=0, p1=0, p2=0, consts=(0, 0)):
        self.kind, self.p0, self.p1, self.p2 = kind, p0, p1, p2
        self.consts = list(consts)
        self.wires = [None] * NUM_WIRES


def tape_instructions(tape):
    """(position, opcode) of every instruction of a recorded witness program (the lengths csrc/witness.hip's op_len gives)"""
    fixed = {OP_ARITH: 8, OP_ARITH_EXT: 12, OP_P2: 26, OP_POSEIDON: 26, OP_BASE_SUM: 2 + 63, OP_RA: 20, OP_REDUCING: 5 + 43 + 2, OP_REDUCING_EXT: 5 + 64 + 2,
             OP_WIRE: 3, OP_HINT_DIV_EXT: 6, OP_HINT_LO63: 2, OP_HINT_HI: 2, OP_HINT_SPLIT: 4, OP_U32_ARITH: 8, OP_U32_SUB: 8, OP_U32_RANGE_CHECK: 4,
             OP_COMPARISON: 6, OP_MUL_EXT: 9}
    t, n = 0, len(tape)
    while t < n:
        op = int(tape[t])
        yield t, op
        if op == OP_COSET:
            t += 1 + 3 + (2 << int(tape[t + 2])) + 4
        elif op == OP_PAR:
            t += 1 + 1 + int(tape[t + 1])
        elif op == OP_U32_ADD_MANY:
            t += 1 + 4 + int(tape[t + 4]) + 3
        elif op == OP_BASE_SPLIT:
            t += 1 + 4 + int(tape[t + 3])
        elif op == OP_EXP:
            t += 1 + 3 + int(tape[t + 2]) + 1
        else:
            t += 1 + fixed[op]


def instruction_slots(tape, pos):
    """(slots read, slots written, (row, col) wire cells written, next position) of the instruction at tape[pos] -- the operand roles
    of csrc/witness.hip's executor"""
    op = int(tape[pos])
    need = {OP_ARITH: 8, OP_ARITH_EXT: 12, OP_P2: 26, OP_POSEIDON: 26, OP_BASE_SUM: 65, OP_RA: 20, OP_REDUCING: 50, OP_REDUCING_EXT: 71, OP_WIRE: 3,
            OP_HINT_DIV_EXT: 6, OP_HINT_LO63: 2, OP_HINT_HI: 2, OP_HINT_SPLIT: 4}.get(op)
    if need is None and op == OP_COSET:
        need = 3 + (2 << int(tape[pos + 2])) + 4
    t = tape[pos + 1:pos + 1 + (need or 0)]
    if op == OP_ARITH:
        return t[4:7], [t[7]], [(t[0], 4 * t[1] + k) for k in range(4)], pos + 9
    if op == OP_ARITH_EXT:
        return t[4:10], t[10:12], [(t[0], 8 * t[1] + k) for k in range(8)], pos + 13
    if op in (OP_P2, OP_POSEIDON):
        return t[1:14], t[14:26], [(t[0], c) for c in range(135)], pos + 27
    if op == OP_BASE_SUM:
        return [t[1]], t[2:2 + 63], [(t[0], c) for c in range(64)], pos + 1 + 2 + 63
    if op == OP_RA:
        base = (2 + 16) * t[1]
        return t[2:19], [t[19]], [(t[0], base + k) for k in range(18)] + [(t[0], (2 + 16) * 4 + 2 + 4 * t[1] + k) for k in range(4)], pos + 21
    if op == OP_REDUCING:
        return t[1:5 + 43], t[5 + 43:5 + 43 + 2], [(t[0], c) for c in range(135)], pos + 1 + 5 + 43 + 2
    if op == OP_REDUCING_EXT:
        return t[1:5 + 64], t[5 + 64:5 + 64 + 2], [(t[0], c) for c in range(135)], pos + 1 + 5 + 64 + 2
    if op == OP_COSET:
        npts = 1 << t[1]
        return t[2:3 + 2 * npts + 2], t[3 + 2 * npts + 2:3 + 2 * npts + 4], [(t[0], c) for c in range(135)], pos + 1 + 3 + 2 * npts + 4
    if op == OP_WIRE:
        return [t[2]], [], [(t[0], t[1])], pos + 4
    if op == OP_HINT_DIV_EXT:
        return t[0:4], t[4:6], [], pos + 7
    if op in (OP_HINT_LO63, OP_HINT_HI):
        return [t[0]], [t[1]], [], pos + 3
    if op == OP_HINT_SPLIT:
        return [t[0]], t[2:4], [], pos + 5
    t = tape[pos + 1:]
    if op in (OP_U32_ARITH, OP_U32_SUB):
        per, limbs = (6, 32) if op == OP_U32_ARITH else (5, 16)
        row, i, ops = t[0], t[1], t[2]
        return t[3:6], t[6:8], [(row, per * i + k) for k in range(per)] + [(row, per * ops + limbs * i + j) for j in range(limbs)], pos + 9
    if op == OP_U32_ADD_MANY:
        row, i, ops, na = t[0], t[1], t[2], t[3]
        per = na + 3
        return t[4:5 + na], t[5 + na:7 + na], [(row, per * i + k) for k in range(per)] + [(row, per * ops + 18 * i + j) for j in range(18)], pos + 1 + 4 + na + 3
    if op == OP_U32_RANGE_CHECK:
        row, i, k = t[0], t[1], t[2]
        return [t[3]], [], [(row, i)] + [(row, k + 16 * i + j) for j in range(16)], pos + 5
    if op == OP_COMPARISON:
        return t[3:5], [t[5]], [(t[0], c) for c in range(135)], pos + 7
    if op == OP_BASE_SPLIT:
        return [t[3]], t[4:4 + t[2]], [(t[0], c) for c in range(1 + t[2])], pos + 1 + 4 + t[2]
    if op == OP_MUL_EXT:
        return t[3:7], t[7:9], [(t[0], 6 * t[1] + k) for k in range(6)], pos + 10
    if op == OP_EXP:
        nb = t[1]
        return t[2:3 + nb], [t[3 + nb]], [(t[0], c) for c in range(2 * nb + 2)], pos + 1 + 3 + nb + 1
    raise ValueError(f"opcode {op}")


def check_sections_independent(tape, start, lengths):
    """the sections [start, start + lengths[0]), ... of a parallel region may run concurrently: no section reads a slot another
    section writes, no two sections write the same slot or the same wire cell"""
    slot_writer, cell_writer, reads = {}, {}, []
    pos = start
    for k, ln in enumerate(lengths):
        end = pos + ln
        rd = set()
        while pos < end:
            r, w, cells, pos = instruction_slots(tape, pos)
            rd.update(r)
            for sl in w:
                assert slot_writer.setdefault(sl, k) == k, f"sections {slot_writer[sl]} and {k} both write slot {sl}"
            for c in cells:
                assert cell_writer.setdefault(c, k) == k, f"sections {cell_writer[c]} and {k} both write wire {c}"
        assert pos == end, "a section ends inside an instruction"
        reads.append(rd)
    for k, rd in enumerate(reads):
        for sl in rd:
            assert slot_writer.get(sl, k) == k, f"section {k} reads slot {sl} written by section {slot_writer[sl]}"


class _ParallelRegion:
    def __init__(self, b):
        self.b, self.lengths, self.at = b, [], None

    def __enter__(self):
        assert not getattr(self.b, "_in_region", False), "parallel regions do not nest"
        self.b._in_region = True
        self.start = self.at = len(self.b.tape)
        return self

    def section(self):
        return _ParallelSection(self)

    def __exit__(self, exc_type, exc, tb):
        self.b._in_region = False
        if exc_type is None:
            assert len(self.b.tape) == self.at, "witness operations recorded between the sections of a parallel region"
            if len(self.lengths) > 1:
                check_sections_independent(self.b.tape, self.start, self.lengths)
                self.b.tape[self.start:self.start] = [OP_PAR, len(self.lengths)] + self.lengths
        return False


class _ParallelSection:
    def __init__(self, region):
        self.r = region

    def __enter__(self):
        assert len(self.r.b.tape) == self.r.at, "witness operations recorded between the sections of a parallel region"
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.r.lengths.append(len(self.r.b.tape) - self.r.at)
            self.r.at = len(self.r.b.tape)
        return False


class Builder:
    """standard_recursion_config: 135 wires, 80 routed, 2 gate constants per row"""
    ARITH_OPS, ARITH_EXT_OPS = 20, 10
    RA_BITS, RA_COPIES = 4, 4
    BASE_SUM_LIMBS = 63
    REDUCING_COEFFS = 43
    REDUCING_EXT_COEFFS = 32

    def __init__(self, strict=True, hasher=0):
        # hasher: the circuit's config hasher (0 = Poseidon2GoldilocksConfig, the reference's C; 1 = PoseidonGoldilocksConfig, its
        # WrapC): what build() hashes the public inputs with (C::InnerHasher) -- Poseidon2Gate or PoseidonGate rows. The prover of
        # the circuit must commit and draw challenges with the same hasher (CircuitProver(variant=...)).
        self.hasher = hasher
        # strict: connect() asserts equal values (a wrong witness fails where it is produced). Off, the builder records
        # the copy constraint anyway and the witness simply violates it -- what a dishonest prover would hand to prove()
        self.strict = strict
        self.rows = []
        self.parent = {}
        self.open = {}      # slot key -> [row index, used slots]
        self.consts = {}    # value -> T
        self.public_inputs = []
        self.tape = []          # flat list of ints (u64)
        self.n_slots = 0
        self.input_sids = []    # slots the caller fills, in add_virtual order
        self.input_vals = []
        self.const_slots = []   # (sid, value)

    # ---- cells and copy constraints ------------------------------------------------------------------------------
    def _find(self, c):
        p = self.parent
        root = c
        while p.get(root, root) != root:
            root = p[root]
        while p.get(c, c) != root:
            p[c], c = root, p[c]
        return root

    def _union(self, a, b):
        ra, rb = self._find(a), self._find(b)
        if ra != rb:
            self.parent[ra] = rb

    def _put(self, row, col, t):
        """write target t into routed wire (row, col): the cell becomes its home, or is copy-constrained to it"""
        assert col < NUM_ROUTED
        w = self.rows[row].wires
        assert w[col] is None
        w[col] = t.v
        if t.cell is None:
            t.cell = (row, col)
        else:
            self._union((row, col), t.cell)

    def _sid(self):
        self.n_slots += 1
        return self.n_slots - 1

    def _out(self, row, col, v):
        self.rows[row].wires[col] = v % P
        return T(v, (row, col), self._sid())

    def connect(self, a, b):
        assert not self.strict or a.v == b.v, f"connect: {a.v} != {b.v}"
        if a.cell is None and b.cell is None:
            # two free targets: give them a home in a constant-free arithmetic slot (0 * 0 * 1 + 0 * x): cheap and rare
            self.arithmetic(0, a, a, 0, a)
        if a.cell is None:
            a.cell = b.cell
        elif b.cell is None:
            b.cell = a.cell
        else:
            self._union(a.cell, b.cell)

    def connect_ext(self, a, b):
        self.connect(a.a, b.a)
        self.connect(a.b, b.b)

    def add_virtual(self, v):
        """add_virtual_target: a witness value (an input of the witness program); its home is the first wire it is used in"""
        t = T(v, None, self._sid())
        self.input_sids.append(t.sid)
        self.input_vals.append(t.v)
        return t

    def add_virtual_ext(self, v):
        return E(self.add_virtual(v[0]), self.add_virtual(v[1]))

    def _hint(self, v):
        """a value the witness generator computes from other values (not an input)"""
        return T(v, None, self._sid())

    def parallel_sections(self):
        """`with b.parallel_sections() as region:` then `with region.section():` around each of a run of INDEPENDENT pieces of the
        circuit (they may read what was built before the region, not each other's values): the recorded witness program marks
        them, and the executor replays them on separate threads when it has fewer proofs than threads. Nothing changes in the
        circuit or in the values. Regions do not nest."""
        return _ParallelRegion(self)

    # ---- rows and slots ---------------------------------------------------------------------------------------------
    def _new_row(self, kind, p0=0, p1=0, p2=0, consts=(0, 0)):
        self.rows.append(Row(kind, p0, p1, p2, consts))
        return len(self.rows) - 1

    def _slot(self, key, per_row, make):
        st = self.open.get(key)
        if st is None or st[1] == per_row:
            st = self.open[key] = [make(), 0]
        st[1] += 1
        return st[0], st[1] - 1

    # ---- ConstantGate ------------------------------------------------------------------------------------------------
    def constant(self, v):
        """CircuitBuilder::constant: one target per distinct value. Its source wire is chosen at build() (plonky2 defers
        it the same way): first the two extra-constant slots every RandomAccessGate row carries, ConstantGate rows only
        for what is left -- a verifier circuit has ~200 RandomAccess rows and ~300 constants, so it needs no ConstantGate."""
        v %= P
        t = self.consts.get(v)
        if t is None:
            t = self.consts[v] = T(v, None, self._sid())
            self.const_slots.append((t.sid, v))
        return t

    def _place_constants(self):
        vs = 1 << self.RA_BITS
        free = [(r, i) for r, row in enumerate(self.rows) if row.kind == C.RANDOM_ACCESS for i in range(2)]
        free.reverse()
        for v, t in self.consts.items():
            if free:
                r, i = free.pop()
                col = (2 + vs) * self.RA_COPIES + i
            else:
                r, i = self._slot(("const",), 2, lambda: self._new_row(C.CONSTANT, 2))
                col = i
            self.rows[r].consts[i] = v
            self.rows[r].wires[col] = v
            if t.cell is None:
                t.cell = (r, col)
            else:
                self._union((r, col), t.cell)
            self.tape += [OP_WIRE, r, col, t.sid]

    def zero(self):
        return self.constant(0)

    def one(self):
        return self.constant(1)

    def constant_ext(self, v):
        return E(self.constant(v[0]), self.constant(v[1]))

    def zero_ext(self):
        return E(self.zero(), self.zero())

    def one_ext(self):
        return E(self.one(), self.zero())

    def to_ext(self, t):
        return E(t, self.zero())

    # ---- ArithmeticGate: c0 m0 m1 + c1 addend -----------------------------------------------------------------------------
    def arithmetic(self, c0, m0, m1, c1, ad):
        c0 %= P
        c1 %= P
        row, i = self._slot(("arith", c0, c1), self.ARITH_OPS, lambda: self._new_row(C.ARITHMETIC, self.ARITH_OPS, consts=(c0, c1)))
        self._put(row, 4 * i, m0)
        self._put(row, 4 * i + 1, m1)
        self._put(row, 4 * i + 2, ad)
        out = self._out(row, 4 * i + 3, c0 * m0.v * m1.v + c1 * ad.v)
        self.tape += [OP_ARITH, row, i, c0, c1, m0.sid, m1.sid, ad.sid, out.sid]
        return out

    def mul(self, a, b):
        return self.arithmetic(1, a, b, 0, a)

    def add(self, a, b):
        return self.arithmetic(1, a, self.one(), 1, b)

    def sub(self, a, b):
        return self.arithmetic(1, a, self.one(), P - 1, b)

    def mul_add(self, a, b, c):
        return self.arithmetic(1, a, b, 1, c)

    def mul_const(self, k, a):
        return self.arithmetic(k, a, self.one(), 0, a)

    def mul_const_add(self, k, a, b):
        return self.arithmetic(k, a, self.one(), 1, b)

    def add_const(self, a, k):
        return self.arithmetic(1, a, self.one(), k, self.one())

    def assert_zero(self, a):
        self.connect(a, self.zero())

    def assert_bool(self, b):
        self.assert_zero(self.arithmetic(1, b, b, P - 1, b))  # b^2 - b

    def select(self, bit, x, y):
        """bit ? x : y = bit (x - y) + y"""
        return self.mul_add(bit, self.sub(x, y), y)

    def le_sum(self, bits):
        acc = self.zero()
        for b in reversed(bits):
            acc = self.arithmetic(2, acc, self.one(), 1, b)
        return acc

    def exp_from_bits_const_base(self, base, bits):
        """base^(sum bits_i 2^i) for a constant base: prod_i (bit_i (base^(2^i) - 1) + 1)"""
        acc = self.one()
        pw = base % P
        for b in bits:
            f = self.arithmetic(pw - 1, b, self.one(), 1, self.one())
            acc = self.mul(acc, f)
            pw = pw * pw % P
        return acc

    def exp_power_of_2(self, x, k):
        for _ in range(k):
            x = self.mul(x, x)
        return x

    # ---- ArithmeticExtensionGate: c0 m0 m1 + c1 addend over the extension ---------------------------------------------------
    def arithmetic_ext(self, c0, m0, m1, c1, ad):
        c0 %= P
        c1 %= P
        row, i = self._slot(("arith_ext", c0, c1), self.ARITH_EXT_OPS,
                            lambda: self._new_row(C.ARITHMETIC_EXT, self.ARITH_EXT_OPS, consts=(c0, c1)))
        for k, t in enumerate((m0.a, m0.b, m1.a, m1.b, ad.a, ad.b)):
            self._put(row, 8 * i + k, t)
        r = xadd(xscale(xmul(m0.v, m1.v), c0), xscale(ad.v, c1))
        out = E(self._out(row, 8 * i + 6, r[0]), self._out(row, 8 * i + 7, r[1]))
        self.tape += [OP_ARITH_EXT, row, i, c0, c1, m0.a.sid, m0.b.sid, m1.a.sid, m1.b.sid, ad.a.sid, ad.b.sid, out.a.sid, out.b.sid]
        return out

    def mul_ext(self, a, b):
        return self.arithmetic_ext(1, a, b, 0, a)

    def add_ext(self, a, b):
        return self.arithmetic_ext(1, a, self.one_ext(), 1, b)

    def sub_ext(self, a, b):
        return self.arithmetic_ext(1, a, self.one_ext(), P - 1, b)

    def mul_add_ext(self, a, b, c):
        return self.arithmetic_ext(1, a, b, 1, c)

    def mul_sub_ext(self, a, b, c):
        return self.arithmetic_ext(1, a, b, P - 1, c)

    # A row of ArithmeticExtensionGate shares its two gate constants: an operation with a constant nobody else uses
    # would own a whole row. Small constants that recur everywhere (matrix entries, -1, 7) stay gate constants; any
    # other constant comes from a ConstantGate wire (two per row) and the operation joins the shared (1, 0) / (1, 1) rows.
    _HOT = frozenset(list(range(0, 9)) + [P - 1, P - 2])

    def mul_const_ext(self, k, a):
        k %= P
        if k in self._HOT:
            return self.arithmetic_ext(k, a, self.one_ext(), 0, a)
        return self.mul_ext(self.constant_ext((k, 0)), a)

    def mul_const_add_ext(self, k, a, b):
        k %= P
        if k in self._HOT:
            return self.arithmetic_ext(k, a, self.one_ext(), 1, b)
        return self.mul_add_ext(self.constant_ext((k, 0)), a, b)

    def add_const_ext(self, a, k):
        """a + k for a base-field constant k"""
        k %= P
        if k in self._HOT:
            return self.arithmetic_ext(1, a, self.one_ext(), k, self.one_ext())
        return self.add_ext(a, self.constant_ext((k, 0)))

    def scalar_mul_ext(self, s, a):
        """base target s times extension a"""
        return self.mul_ext(self.to_ext(s), a)

    def div_ext(self, num, den):
        """q with q * den = num (the quotient is a witness, the product is constrained)"""
        qv = xmul(num.v, xinv(den.v))
        q = E(self._hint(qv[0]), self._hint(qv[1]))
        self.tape += [OP_HINT_DIV_EXT, num.a.sid, num.b.sid, den.a.sid, den.b.sid, q.a.sid, q.b.sid]
        self.connect_ext(self.mul_ext(q, den), num)
        return q

    def exp_power_of_2_ext(self, x, k):
        for _ in range(k):
            x = self.mul_ext(x, x)
        return x

    def reduce_with_powers_ext(self, terms, alpha):
        """sum_i terms[i] alpha^i, alpha an extension target (Horner from the back)"""
        acc = self.zero_ext()
        for t in reversed(terms):
            acc = self.mul_add_ext(acc, alpha, t)
        return acc

    # ---- Poseidon2Gate --------------------------------------------------------------------------------------------------
    def permute_swapped(self, inputs, swap):
        """one Poseidon2 permutation row; swap (a boolean target) exchanges inputs[0..4) and [4..8) first"""
        row = self._new_row(C.POSEIDON2)
        w = self.rows[row].wires
        for i, t in enumerate(inputs):
            self._put(row, i, t)
        self._put(row, 24, swap)
        Kc = K2()
        s = [0] * 12
        for i in range(4):
            delta = swap.v * (inputs[i + 4].v - inputs[i].v) % P
            w[25 + i] = delta
            s[i], s[i + 4] = (inputs[i].v + delta) % P, (inputs[i + 4].v - delta) % P
        for i in range(8, 12):
            s[i] = inputs[i].v
        s = C.p2_external(s)
        for r in range(4):
            s = [(s[i] + Kc["POSEIDON2_RC_EXT"][12 * r + i]) % P for i in range(12)]
            if r:
                w[29 + 12 * (r - 1):29 + 12 * r] = s
            s = C.p2_external([pow(x, 7, P) for x in s])
        for r in range(22):
            s[0] = (s[0] + Kc["POSEIDON2_RC_INT"][r]) % P
            w[65 + r] = s[0]
            s[0] = pow(s[0], 7, P)
            s = C.p2_internal(s)
        for r in range(4):
            s = [(s[i] + Kc["POSEIDON2_RC_EXT"][12 * (4 + r) + i]) % P for i in range(12)]
            w[87 + 12 * r:87 + 12 * (r + 1)] = s
            s = C.p2_external([pow(x, 7, P) for x in s])
        outs = [self._out(row, 12 + i, s[i]) for i in range(12)]
        self.tape += [OP_P2, row] + [t.sid for t in inputs] + [swap.sid] + [t.sid for t in outs]
        return outs

    def permute(self, inputs):
        return self.permute_swapped(inputs, self.zero())

    def permute_poseidon_swapped(self, inputs, swap):
        """one PoseidonGate row ([dep] plonky2 gates/poseidon.rs: the original Poseidon permutation, same wire layout as the Poseidon2
        gate: inputs 0..11, outputs 12..23, swap 24, deltas 25..28, full-round S-box inputs 29.. and 87.., partial-round ones 65..)"""
        row = self._new_row(C.POSEIDON)
        w = self.rows[row].wires
        for i, t in enumerate(inputs):
            self._put(row, i, t)
        self._put(row, 24, swap)
        Kc = K2()
        s = [0] * 12
        for i in range(4):
            delta = swap.v * (inputs[i + 4].v - inputs[i].v) % P
            w[25 + i] = delta
            s[i], s[i + 4] = (inputs[i].v + delta) % P, (inputs[i + 4].v - delta) % P
        for i in range(8, 12):
            s[i] = inputs[i].v
        for r in range(30):
            s = [(s[i] + Kc["POSEIDON_RC"][12 * r + i]) % P for i in range(12)]
            if 4 <= r < 26:
                w[65 + r - 4] = s[0]
                s[0] = pow(s[0], 7, P)
            else:
                if r:
                    base = 29 + 12 * (r - 1) if r < 4 else 87 + 12 * (r - 26)
                    w[base:base + 12] = s
                s = [pow(x, 7, P) for x in s]
            s = C.poseidon_mds(s)
        outs = [self._out(row, 12 + i, s[i]) for i in range(12)]
        self.tape += [OP_POSEIDON, row] + [t.sid for t in inputs] + [swap.sid] + [t.sid for t in outs]
        return outs

    def hash_n_to_m_no_pad(self, inputs, m, hasher=0):
        """hashing.rs hash_n_to_m_no_pad: overwrite-mode absorb (rate 8), squeeze from the front; hasher 1 = PoseidonHash (PoseidonGate rows)"""
        z = self.zero()
        perm = self.permute if hasher == 0 else (lambda st: self.permute_poseidon_swapped(st, z))
        state = [z] * 12
        for i in range(0, len(inputs), 8):
            chunk = inputs[i:i + 8]
            state = perm(list(chunk) + state[len(chunk):])
        if not inputs:
            state = perm(state)
        out = []
        while True:
            for t in state[:8]:
                out.append(t)
                if len(out) == m:
                    return out
            state = perm(state)

    def hash_or_noop(self, inputs):
        if len(inputs) <= 4:
            return list(inputs) + [self.zero()] * (4 - len(inputs))
        return self.hash_n_to_m_no_pad(inputs, 4)

    # ---- BaseSumGate<2>: bits ----------------------------------------------------------------------------------------------
    def split_le_base2(self, x, num_bits):
        """one BaseSumGate<2> row: x = sum limbs_i 2^i with num_bits <= 63 boolean limbs (the gate has 63; the unused
        high limbs are zero)"""
        assert num_bits <= self.BASE_SUM_LIMBS and (not self.strict or x.v < (1 << num_bits))
        row = self._new_row(C.BASE_SUM, self.BASE_SUM_LIMBS, 2)
        self._put(row, 0, x)
        limbs = [self._out(row, 1 + i, (x.v >> i) & 1) for i in range(self.BASE_SUM_LIMBS)]
        self.tape += [OP_BASE_SUM, row, x.sid] + [t.sid for t in limbs]
        for t in limbs[num_bits:]:
            self.assert_zero(t)
        return limbs[:num_bits]

    def split_le(self, x, num_bits):
        """split_join.rs split_le: little-endian bits through ceil(num_bits / 63) BaseSum gates, recombined with
        weights 2^(63 i)"""
        if num_bits <= self.BASE_SUM_LIMBS:
            return self.split_le_base2(x, num_bits)
        lo = self._hint(x.v & ((1 << 63) - 1))
        hi = self._hint(x.v >> 63)
        self.tape += [OP_HINT_LO63, x.sid, lo.sid, OP_HINT_HI, x.sid, hi.sid]
        bits = self.split_le_base2(lo, 63) + self.split_le_base2(hi, num_bits - 63)
        self.connect(self.arithmetic(1 << 63, hi, self.one(), 1, lo), x)
        return bits

    def range_check(self, x, n_bits):
        self.split_le_base2(x, n_bits)

    def split_low_high(self, x, n_log, num_bits):
        """gadgets/range_check.rs split_low_high: x = low + 2^n_log high with low < 2^n_log and high < 2^(num_bits - n_log), both
        range-checked (LowHighGenerator supplies them)"""
        lo = self._hint(x.v & ((1 << n_log) - 1))
        hi = self._hint(x.v >> n_log)
        self.tape += [OP_HINT_SPLIT, x.sid, n_log, lo.sid, hi.sid]
        self.range_check(lo, n_log)
        self.range_check(hi, num_bits - n_log)
        self.connect(self.arithmetic(1 << n_log, hi, self.one(), 1, lo), x)
        return lo, hi

    def is_equal(self, x, y):
        """gadgets/arithmetic.rs is_equal: equal = 1 - (x - y) inv with inv = (x - y)^-1 or 0 (EqualityGenerator), and
        (x - y) equal = 0; the result is a boolean by construction"""
        diff = self.sub(x, y)
        inv_v = pow(diff.v, P - 2, P)
        inv, unused = self._hint(inv_v), self._hint(0)
        one, zero = self.one(), self.zero()
        self.tape += [OP_HINT_DIV_EXT, one.sid, zero.sid, diff.sid, zero.sid, inv.sid, unused.sid]  # (1, 0) / (diff, 0); 0 for diff = 0
        equal = self.sub(one, self.mul(diff, inv))
        self.connect(self.mul(diff, equal), zero)
        return equal

    def not_(self, a):
        return self.sub(self.one(), a)

    def or_(self, a, b):
        """a + b - a b for booleans"""
        return self.sub(self.add(a, b), self.mul(a, b))

    # ---- the leaf circuits' user-logic gates (plonky2-u32, BaseSumGate<4>, MulExtensionGate, ExponentiationGate) ------------------
    # Each method places one operation in a row of its gate (gate parameters = the reference's leaf gate set, circuits.LEAF_KINDS),
    # and records ONE tape instruction for it (MP2G_OP_U32_ARITH ..: the generator runs inside the witness replay, host or device).
    def u32_arithmetic(self, m0, m1, addend, ops=3):
        """U32ArithmeticGate / U32ArithmeticGenerator: m0 m1 + addend = low + 2^32 high over u32 operands. Returns (low, high)."""
        row, i = self._slot(("u32arith", ops), ops, lambda: self._new_row(C.U32_ARITHMETIC, ops))
        out = (m0.v * m1.v + addend.v) % P
        lo, hi, b = out & 0xFFFFFFFF, out >> 32, 6 * i
        self._put(row, b, m0), self._put(row, b + 1, m1), self._put(row, b + 2, addend)
        tlo, thi = self._out(row, b + 3, lo), self._out(row, b + 4, hi)
        w = self.rows[row].wires
        w[b + 5] = pow((0xFFFFFFFF - hi) % P, P - 2, P)
        for j in range(32):
            w[6 * ops + 32 * i + j] = (out >> (2 * j)) & 3
        self.tape += [OP_U32_ARITH, row, i, ops, m0.sid, m1.sid, addend.sid, tlo.sid, thi.sid]
        return tlo, thi

    def u32_sub(self, x, y, borrow_in, ops=6):
        """U32SubtractionGate / U32SubtractionGenerator: x - y - borrow_in = result - 2^32 borrow_out. Returns (result, borrow_out)."""
        row, i = self._slot(("u32sub", ops), ops, lambda: self._new_row(C.U32_SUBTRACTION, ops))
        r0 = (x.v - y.v - borrow_in.v) % P
        bo = 1 if r0 > 1 << 32 else 0
        r, b = (r0 + (bo << 32)) % P, 5 * i
        self._put(row, b, x), self._put(row, b + 1, y), self._put(row, b + 2, borrow_in)
        tr, tb = self._out(row, b + 3, r), self._out(row, b + 4, bo)
        w = self.rows[row].wires
        for j in range(16):
            w[5 * ops + 16 * i + j] = (r >> (2 * j)) & 3
        self.tape += [OP_U32_SUB, row, i, ops, x.sid, y.sid, borrow_in.sid, tr.sid, tb.sid]
        return tr, tb

    def u32_add_many(self, addends, carry_in, ops=5):
        """U32AddManyGate(len(addends), ops) / U32AddManyGenerator: sum + carry_in = result + 2^32 carry_out. Returns (result, carry_out)."""
        na = len(addends)
        per = na + 3
        row, i = self._slot(("u32addmany", na, ops), ops, lambda: self._new_row(C.U32_ADD_MANY, na, ops))
        tot = (sum(a.v for a in addends) + carry_in.v) % P
        res, co, b = tot & 0xFFFFFFFF, tot >> 32, per * i
        for k, a in enumerate(addends):
            self._put(row, b + k, a)
        self._put(row, b + na, carry_in)
        tr, tc = self._out(row, b + na + 1, res), self._out(row, b + na + 2, co)
        w = self.rows[row].wires
        for j in range(16):
            w[per * ops + 18 * i + j] = (res >> (2 * j)) & 3
        for j in range(2):
            w[per * ops + 18 * i + 16 + j] = (co >> (2 * j)) & 3
        self.tape += [OP_U32_ADD_MANY, row, i, ops, na] + [a.sid for a in addends] + [carry_in.sid, tr.sid, tc.sid]
        return tr, tc

    def u32_range_check(self, x, k=7):
        """U32RangeCheckGate(k) / U32RangeCheckGenerator: x < 2^32"""
        row, i = self._slot(("u32rc", k), k, lambda: self._new_row(C.U32_RANGE_CHECK, k))
        self._put(row, i, x)
        w = self.rows[row].wires
        for j in range(16):
            w[k + 16 * i + j] = (x.v >> (2 * j)) & 3
        self.tape += [OP_U32_RANGE_CHECK, row, i, k, x.sid]

    def comparison_le(self, first, second, num_bits=32, num_chunks=16):
        """ComparisonGate(num_bits, num_chunks) / ComparisonGenerator: the boolean first <= second for operands below 2^num_bits"""
        row = self._new_row(C.COMPARISON, num_bits, num_chunks)
        cb = (num_bits + num_chunks - 1) // num_chunks
        a, b = first.v, second.v
        self._put(row, 0, first), self._put(row, 1, second)
        w = self.rows[row].wires
        msd = 0
        for i in range(num_chunks):
            fc, sc = (a >> (cb * i)) & ((1 << cb) - 1), (b >> (cb * i)) & ((1 << cb) - 1)
            diff = (sc - fc) % P
            eq = 1 if diff == 0 else 0
            w[4 + i], w[4 + num_chunks + i] = fc, sc
            w[4 + 2 * num_chunks + i] = 1 if eq else pow(diff, P - 2, P)
            w[4 + 3 * num_chunks + i] = eq
            iv = msd if eq else 0
            w[4 + 4 * num_chunks + i] = iv
            msd = iv if eq else diff
        w[3] = msd
        val = ((1 << cb) + msd) % P
        for i in range(cb + 1):
            w[4 + 5 * num_chunks + i] = (val >> i) & 1
        res = self._out(row, 2, (val >> cb) & 1)
        self.tape += [OP_COMPARISON, row, num_bits, num_chunks, first.sid, second.sid, res.sid]
        return res

    def base_split(self, x, base_bits, n_limbs):
        """BaseSumGate<2^base_bits> with n_limbs limbs / BaseSplitGenerator: the little-endian digits of x (the reference's leaf gate
        set carries BaseSumGate<4> with 20 limbs beside the bit splits)"""
        row = self._new_row(C.BASE_SUM, n_limbs, 1 << base_bits)
        self._put(row, 0, x)
        limbs = [self._out(row, 1 + j, (x.v >> (base_bits * j)) & ((1 << base_bits) - 1)) for j in range(n_limbs)]
        self.tape += [OP_BASE_SPLIT, row, base_bits, n_limbs, x.sid] + [t.sid for t in limbs]
        return limbs

    def mul_ext_gate(self, c0, m0, m1):
        """MulExtensionGate (13 operations a row) / MulExtensionGenerator: c0 m0 m1 over the extension"""
        c0 %= P
        row, i = self._slot(("mulext", c0), 13, lambda: self._new_row(C.MUL_EXT, 13, consts=(c0, 0)))
        b = 6 * i
        self._put(row, b, m0.a), self._put(row, b + 1, m0.b), self._put(row, b + 2, m1.a), self._put(row, b + 3, m1.b)
        o = xscale(xmul(m0.v, m1.v), c0)
        out = E(self._out(row, b + 4, o[0]), self._out(row, b + 5, o[1]))
        self.tape += [OP_MUL_EXT, row, i, c0, m0.a.sid, m0.b.sid, m1.a.sid, m1.b.sid, out.a.sid, out.b.sid]
        return out

    def exponentiation(self, base, bits):
        """ExponentiationGate(len(bits)) / ExponentiationGenerator: base^(sum bits_j 2^j)"""
        nb = len(bits)
        row = self._new_row(C.EXPONENTIATION, nb)
        self._put(row, 0, base)
        for j, bt in enumerate(bits):
            self._put(row, 1 + j, bt)
        w = self.rows[row].wires
        cur = 1
        for i in range(nb):
            prev = 1 if i == 0 else cur * cur % P
            cur = prev * (base.v if bits[nb - 1 - i].v else 1) % P
            w[nb + 2 + i] = cur
        out = self._out(row, nb + 1, cur)
        self.tape += [OP_EXP, row, nb, base.sid] + [bt.sid for bt in bits] + [out.sid]
        return out

    # ---- RandomAccessGate (bits 4, 4 copies, 2 extra constants) ---------------------------------------------------------------
    def random_access(self, index, values):
        assert len(values) == 1 << self.RA_BITS
        vs = len(values)
        row, c = self._slot(("ra",), self.RA_COPIES, lambda: self._new_row(C.RANDOM_ACCESS, self.RA_BITS, self.RA_COPIES, 2))
        base = (2 + vs) * c
        self._put(row, base, index)
        for i, t in enumerate(values):
            self._put(row, base + 2 + i, t)
        routed = (2 + vs) * self.RA_COPIES + 2
        w = self.rows[row].wires
        for i in range(self.RA_BITS):
            w[routed + c * self.RA_BITS + i] = (index.v >> i) & 1
        assert index.v < vs or not self.strict
        out = self._out(row, base + 1, values[index.v % vs].v)
        self.tape += [OP_RA, row, c, index.sid] + [t.sid for t in values] + [out.sid]
        return out

    def random_access_ext(self, index, values):
        return E(self.random_access(index, [v.a for v in values]), self.random_access(index, [v.b for v in values]))

    # ---- ReducingGate: acc <- acc alpha + coeff over base-field coefficients ---------------------------------------------------
    def reduce_base(self, alpha, terms):
        """ReducingFactorTarget::reduce_base: sum_i terms[i] alpha^i for base-field targets. The terms are padded with
        zeros to a multiple of the gate's 43 coefficients and fed highest power first (the padding leads and is inert)."""
        n = self.REDUCING_COEFFS
        rev = list(reversed(terms))
        rev = [self.zero()] * ((-len(rev)) % n) + rev
        acc = self.zero_ext()
        start_accs = 6 + n
        for lo in range(0, len(rev), n):
            row = self._new_row(C.REDUCING, n)
            for k, t in enumerate((alpha.a, alpha.b, acc.a, acc.b)):
                self._put(row, 2 + k, t)
            w = self.rows[row].wires
            cur = acc.v
            for i in range(n):
                self._put(row, 6 + i, rev[lo + i])
                cur = xadd(xmul(cur, alpha.v), (rev[lo + i].v, 0))
                if i < n - 1:
                    w[start_accs + 2 * i], w[start_accs + 2 * i + 1] = cur
            old = acc
            acc = E(self._out(row, 0, cur[0]), self._out(row, 1, cur[1]))
            self.tape += [OP_REDUCING, row, alpha.a.sid, alpha.b.sid, old.a.sid, old.b.sid] + [t.sid for t in rev[lo:lo + n]] + [acc.a.sid, acc.b.sid]
        return acc

    def reduce_ext(self, alpha, terms):
        """ReducingFactorTarget::reduce: sum_i terms[i] alpha^i for extension targets through ReducingExtensionGates
        (32 coefficients per row)"""
        n = self.REDUCING_EXT_COEFFS
        rev = list(reversed(terms))
        rev = [self.zero_ext()] * ((-len(rev)) % n) + rev
        acc = self.zero_ext()
        start_accs = 6 + 2 * n
        for lo in range(0, len(rev), n):
            row = self._new_row(C.REDUCING_EXT, n)
            for k, t in enumerate((alpha.a, alpha.b, acc.a, acc.b)):
                self._put(row, 2 + k, t)
            w = self.rows[row].wires
            cur = acc.v
            for i in range(n):
                e = rev[lo + i]
                self._put(row, 6 + 2 * i, e.a)
                self._put(row, 7 + 2 * i, e.b)
                cur = xadd(xmul(cur, alpha.v), e.v)
                if i < n - 1:
                    w[start_accs + 2 * i], w[start_accs + 2 * i + 1] = cur
            old = acc
            acc = E(self._out(row, 0, cur[0]), self._out(row, 1, cur[1]))
            self.tape += [OP_REDUCING_EXT, row, alpha.a.sid, alpha.b.sid, old.a.sid, old.b.sid] + [x for e in rev[lo:lo + n] for x in (e.a.sid, e.b.sid)] + \
                [acc.a.sid, acc.b.sid]
        return acc

    # ---- CosetInterpolationGate (subgroup_bits 4) ---------------------------------------------------------------------------------
    def interpolate_coset(self, bits, shift, values, point):
        """the value at `point` (extension) of the polynomial through {(shift g^i, values[i])}, g of order 2^bits"""
        deg = C.coset_interpolation_degree(bits)
        npts = 1 << bits
        nint = (npts - 2) // (deg - 1)
        row = self._new_row(C.COSET_INTERPOLATION, bits, deg)
        w = self.rows[row].wires
        w_pt, w_val = 1 + 2 * npts, 3 + 2 * npts
        w_int = w_val + 2
        w_sh = w_int + 4 * nint
        self._put(row, 0, shift)
        for i, v in enumerate(values):
            self._put(row, 1 + 2 * i, v.a)
            self._put(row, 2 + 2 * i, v.b)
        self._put(row, w_pt, point.a)
        self._put(row, w_pt + 1, point.b)
        om = root_of_unity(bits)
        dom = [pow(om, i, P) for i in range(npts)]
        bw = []
        for i in range(npts):
            pr = 1
            for j in range(npts):
                if j != i:
                    pr = pr * (dom[i] - dom[j]) % P
            bw.append(inv(pr))
        sh = xscale(point.v, inv(shift.v))
        w[w_sh], w[w_sh + 1] = sh
        ev, pr = (0, 0), (1, 0)
        start, end = 0, deg
        for c in range(nint + 1):
            for i in range(start, end):
                val = xscale(values[i].v, bw[i])
                term = ((sh[0] - dom[i]) % P, sh[1])
                ev, pr = xadd(xmul(ev, term), xmul(val, pr)), xmul(pr, term)
            if c == nint:
                break
            w[w_int + 2 * c], w[w_int + 2 * c + 1] = ev
            w[w_int + 2 * (nint + c)], w[w_int + 2 * (nint + c) + 1] = pr
            start = 1 + (deg - 1) * (c + 1)
            end = min(start + deg - 1, npts)
        out = E(self._out(row, w_val, ev[0]), self._out(row, w_val + 1, ev[1]))
        self.tape += [OP_COSET, row, bits, shift.sid] + [x for v in values for x in (v.a.sid, v.b.sid)] + [point.a.sid, point.b.sid, out.a.sid, out.b.sid]
        return out

    # ---- gate set ------------------------------------------------------------------------------------------------------------------
    def add_gate_rows(self, kinds, seed=0xC0FFEE06):
        """one satisfied row of every gate (kind, p0, p1, p2) of `kinds` the circuit does not contain yet. plonky2 evaluates every
        gate of a circuit at every point of the LDE domain, so a circuit's proving cost depends on its gate SET and its degree, not
        on how many rows each gate has: this is how a circuit whose own logic is light is given the gate set (and with
        build(min_log_n) the degree) of the reference's heavier circuit of the same role (base degrees k = 12..15).
        A gate whose generator is a tape instruction (the u32 gates, ComparisonGate, BaseSumGate<4>, MulExtensionGate,
        ExponentiationGate) gets its row from that generator, replayed on host or device like the circuit's own logic
        (_generated_gate_row); the other rows' wires are fixed values (circuits.fill_row: a satisfying assignment) written by OP_WIRE from
        constant slots. No copy constraint touches either."""
        present = {(r.kind, r.p0, r.p1, r.p2) for r in self.rows}
        for kind, p0, p1, p2 in kinds:
            if kind in (C.NOOP, C.PUBLIC_INPUT, C.CONSTANT) or (kind, p0, p1, p2) in present:
                continue
            present.add((kind, p0, p1, p2))
            rng = np.random.default_rng([seed, kind, p0, p1, p2])
            if self._generated_gate_row(kind, p0, p1, p2, rng):
                continue
            w = [None] * NUM_WIRES
            # gate constants (0, 0): a RandomAccessGate row's two extra-constant cells stay free for _place_constants
            C.fill_row(Gate(kind, p0, p1, p2, 0, 0, 0), w, (0, 0), lambda col: int(rng.integers(0, P, dtype=np.uint64)), rng, (0, 0, 0, 0))
            row = self._new_row(kind, p0, p1, p2, consts=(0, 0))
            if kind == C.RANDOM_ACCESS:
                for i in range(p2):
                    w[(2 + (1 << p0)) * p1 + i] = 0
            for col, v in enumerate(w):
                v = int(v) % P
                if v:
                    self.rows[row].wires[col] = v
                    sid = self._sid()
                    self.const_slots.append((sid, v))
                    self.tape += [OP_WIRE, row, col, sid]

    def _free(self, v):
        """a target whose value is preloaded into its slot and that no constraint ties to anything: the operand of a padding row's
        generator (its home is the wire the generator's gate puts it in)"""
        t = T(int(v) % P, None, self._sid())
        self.const_slots.append((t.sid, t.v))
        return t

    def _generated_gate_row(self, kind, p0, p1, p2, rng):
        """add_gate_rows for the gates whose generators are tape instructions (round 6): the row is filled by the gate's own
        generator -- MP2G_OP_U32_ARITH .. MP2G_OP_EXP replayed on host or device -- from preloaded operands, every operation of the
        row used, instead of 30-130 plain wire writes. False: no generator opcode for this gate (the caller writes the wires)."""
        u32 = lambda: self._free(int(rng.integers(0, 1 << 32)))
        if kind == C.U32_ARITHMETIC and 1 <= p0 <= 3:
            for _ in range(p0):
                self.u32_arithmetic(u32(), u32(), u32(), ops=p0)
        elif kind == C.U32_SUBTRACTION and 1 <= p0 <= 6:
            for _ in range(p0):
                self.u32_sub(u32(), u32(), self._free(int(rng.integers(0, 2))), ops=p0)
        elif kind == C.U32_ADD_MANY and 1 <= p0 <= 16 and (p0 + 3 + 18) * p1 <= NUM_WIRES:
            for _ in range(p1):
                self.u32_add_many([u32() for _ in range(p0)], self._free(int(rng.integers(0, 1 << 4))), ops=p1)
        elif kind == C.U32_RANGE_CHECK and 1 <= p0 <= 7:
            for _ in range(p0):
                self.u32_range_check(u32(), k=p0)
        elif kind == C.COMPARISON and 1 <= p0 <= 63 and 1 <= p1 <= 16:
            self.comparison_le(self._free(int(rng.integers(0, 1 << p0, dtype=np.uint64))), self._free(int(rng.integers(0, 1 << p0, dtype=np.uint64))), p0, p1)
        elif kind == C.BASE_SUM and p1 == 4 and 2 * p0 <= 63:
            self.base_split(self._free(int(rng.integers(0, 1 << (2 * p0), dtype=np.uint64))), 2, p0)
        elif kind == C.MUL_EXT and p0 == 13:
            fe = lambda: self._free(int(rng.integers(0, P, dtype=np.uint64)))
            for _ in range(13):
                self.mul_ext_gate(1, E(fe(), fe()), E(fe(), fe()))
            del self.open[("mulext", 1)]  # the row is full; user logic that multiplies later opens its own
        elif kind == C.EXPONENTIATION and 1 <= p0 <= 66:
            self.exponentiation(self._free(int(rng.integers(0, P, dtype=np.uint64))), [self._free(int(rng.integers(0, 2))) for _ in range(p0)])
        else:
            return False
        return True

    # ---- public inputs -------------------------------------------------------------------------------------------------------------
    def register_public_inputs(self, targets):
        self.public_inputs += list(targets)

    # ---- build ---------------------------------------------------------------------------------------------------------------------
    def set_domain_separator(self, values):
        """CircuitBuilder::set_domain_separator: field elements hashed (hash_pad) into the circuit digest, which every transcript of
        the circuit's proofs starts from. The framework's wrapped circuits have none (check_circuit_digest_target)."""
        self.domain_separator = [int(x) for x in values]

    def build(self, min_log_n=6):
        """CircuitBuilder::build: the public-inputs hash bound to a PublicInputGate, rows padded with Noops to a
        power of two, selectors, sigma polynomials from the copy classes. Returns a circuits.Circuit."""
        pi_hash = self.hash_n_to_m_no_pad(self.public_inputs, 4, self.hasher)  # C::InnerHasher of the circuit's config
        pi_row = self._new_row(C.PUBLIC_INPUT)
        for i, t in enumerate(pi_hash):
            self._put(pi_row, i, t)
            self.tape += [OP_WIRE, pi_row, i, t.sid]
        self._place_constants()
        n_rows = len(self.rows) + 1  # at least one Noop (as plonky2's blinding-free padding leaves)
        log_n = max(min_log_n, (n_rows - 1).bit_length())
        n = 1 << log_n
        while len(self.rows) < n:
            self._new_row(C.NOOP)
        kinds = {}
        for r in self.rows:
            kinds.setdefault((r.kind, r.p0, r.p1, r.p2), None)
        gates = [Gate(k, p0, p1, p2, 0, 0, 0) for (k, p0, p1, p2) in kinds]
        gates.sort(key=lambda g: (C.gate_degree(g), g.kind, g.p0, g.p1))
        index = {(g.kind, g.p0, g.p1, g.p2): i for i, g in enumerate(gates)}
        instances = [index[(r.kind, r.p0, r.p1, r.p2)] for r in self.rows]
        cols, sel_idx, groups = C.selector_polynomials(gates, instances)
        for i, g in enumerate(gates):
            g.selector_index, (g.group_start, g.group_end) = sel_idx[i], groups[sel_idx[i]]
        wires = np.zeros((NUM_WIRES, n), dtype=np.uint64)
        for r, row in enumerate(self.rows):
            for c, v in enumerate(row.wires):
                if v is not None:
                    wires[c, r] = v
        wN = root_of_unity(log_n)
        xs = np.array([pow(wN, i, P) for i in range(n)], dtype=object)
        ks = [pow(MULT_GEN, j, P) for j in range(NUM_ROUTED)]
        sig = [[ks[j] * int(x) % P for x in xs] for j in range(NUM_ROUTED)]
        classes = {}
        for c in list(self.parent):
            classes.setdefault(self._find(c), set()).add(c)
        for root, members in classes.items():
            cells = sorted(members | {root})
            ids = [ks[c] * int(xs[r]) % P for r, c in cells]
            for (r, c), v in zip(cells, ids[1:] + ids[:1]):
                sig[c][r] = v
        consts = np.array(cols + [[row.consts[k] for row in self.rows] for k in range(2)], dtype=np.uint64)
        ckt = C.Circuit()
        ckt.log_n, ckt.gates, ckt.num_selectors = log_n, gates, len(cols)
        ckt.pi_hash = np.array([t.v for t in pi_hash], dtype=np.uint64)
        ckt.public_inputs = np.array([t.v for t in self.public_inputs], dtype=np.uint64)
        ckt.pre = np.concatenate([consts, np.array(sig, dtype=np.uint64)])
        ckt.wires = wires
        ckt.num_constants = consts.shape[0]
        ckt.instances = instances
        ckt.pi_row = pi_row
        ckt.gate_array = (Gate * len(gates))(*gates)
        ckt.luts, ckt.num_lookup_selectors, ckt.num_lookup_polys = [], 0, 0
        ckt.n_used_rows = n_rows - 1
        # the witness program of this circuit (see the OP_* table above) and the inputs this build was run with
        ckt.tape = np.array(self.tape, dtype=np.uint64)
        ckt.domain_separator = list(getattr(self, "domain_separator", []))  # CircuitBuilder::set_domain_separator: part of the circuit digest
        ckt.n_slots = self.n_slots
        ckt.input_sids = np.array(self.input_sids, dtype=np.uint32)
        ckt.const_slots = np.array([[sid, v] for sid, v in self.const_slots], dtype=np.uint64).reshape(-1, 2)
        ckt.pi_hash_sids = np.array([t.sid for t in pi_hash], dtype=np.uint32)
        ckt.public_input_sids = np.array([t.sid for t in self.public_inputs], dtype=np.uint32)
        ckt.input_values = np.array(self.input_vals, dtype=np.uint64)  # what this build was run with, in input order
        return ckt


# ---- Fiat-Shamir in the circuit ([dep] iop/challenger.rs RecursiveChallenger) -------------------------------------------------------
class RecursiveChallenger:
    def __init__(self, b):
        self.b = b
        self.state = [b.zero()] * 12
        self.inp, self.out = [], []

    def observe(self, targets):
        for t in targets:
            self.out = []
            self.inp.append(t)
            if len(self.inp) == 8:
                self._duplex()

    def observe_ext(self, exts):
        for e in exts:
            self.observe([e.a, e.b])

    def _duplex(self):
        st = list(self.inp) + self.state[len(self.inp):]
        self.inp = []
        self.state = self.b.permute(st)
        self.out = self.state[:8]

    def get(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()

    def get_n(self, n):
        return [self.get() for _ in range(n)]

    def get_ext(self):
        a = self.get()
        return E(a, self.get())


# ---- Merkle proofs in the circuit ([dep] hash/merkle_proofs.rs verify_merkle_proof_to_cap_with_cap_index) ---------------------------------
def verify_merkle_proof_to_cap(b, leaf, index_bits, cap_index, cap, siblings):
    """leaf: base targets; index_bits: little-endian bits of the leaf index below the cap; cap: 2^cap_height hashes
    (4 targets each); siblings bottom-up. The node hash is two_to_one = permute([l || r || 0000])[0..4] with the
    Poseidon2 gate's swap wire choosing the order."""
    state = b.hash_or_noop(leaf)
    z = b.zero()
    for bit, sib in zip(index_bits, siblings):
        out = b.permute_swapped(list(state) + list(sib) + [z] * 4, bit)
        state = out[:4]
    for i in range(4):
        got = b.random_access(cap_index, [h[i] for h in cap]) if len(cap) > 1 else cap[0][i]
        b.connect(got, state[i])


# ---- gate constraints over extension targets ([dep] gates/*.rs eval_unfiltered_circuit) -----------------------------------------------------
def eval_gate_circuit(b, g, consts, wires, pih):
    """constraints of gate descriptor g at the opened point: consts = local constants after the selector prefix
    (extension targets), wires = local wires, pih = public-inputs hash (base targets). Order as eval_unfiltered."""
    k = g.kind
    out = []
    if k == C.NOOP:
        return out
    if k == C.CONSTANT:
        return [b.sub_ext(consts[i], wires[i]) for i in range(g.p0)]
    if k == C.PUBLIC_INPUT:
        return [b.sub_ext(wires[i], b.to_ext(pih[i])) for i in range(4)]
    if k == C.ARITHMETIC:
        for i in range(g.p0):
            m0, m1, ad, o = wires[4 * i:4 * i + 4]
            t = b.mul_ext(b.mul_ext(m0, m1), consts[0])
            t = b.mul_add_ext(ad, consts[1], t)
            out.append(b.sub_ext(o, t))
        return out
    if k == C.BASE_SUM:
        acc = b.zero_ext()
        for i in reversed(range(g.p0)):
            acc = b.mul_const_add_ext(g.p1, acc, wires[1 + i])
        out.append(b.sub_ext(acc, wires[0]))
        for i in range(g.p0):
            if g.p1 == 2:
                out.append(b.mul_sub_ext(wires[1 + i], wires[1 + i], wires[1 + i]))  # limb (limb - 1)
                continue
            pr = wires[1 + i]
            for kk in range(1, g.p1):
                pr = b.mul_ext(pr, b.add_const_ext(wires[1 + i], P - kk))
            out.append(pr)
        return out
    if k == C.ARITHMETIC_EXT:
        for i in range(g.p0):
            m0, m1, ad, o = [(wires[8 * i + 2 * j], wires[8 * i + 2 * j + 1]) for j in range(4)]
            pr = alg_mul(b, m0, m1)
            c = [b.mul_add_ext(ad[j], consts[1], b.mul_ext(pr[j], consts[0])) for j in range(2)]
            out += [b.sub_ext(o[0], c[0]), b.sub_ext(o[1], c[1])]
        return out
    if k == C.POSEIDON2:
        Kc = K2()
        swap = wires[24]
        out.append(b.mul_sub_ext(swap, swap, swap))  # swap (swap - 1)
        s = [None] * 12
        for i in range(4):
            lhs, rhs, delta = wires[i], wires[i + 4], wires[25 + i]
            out.append(b.sub_ext(b.mul_ext(swap, b.sub_ext(rhs, lhs)), delta))
            s[i], s[i + 4] = b.add_ext(lhs, delta), b.sub_ext(rhs, delta)
        for i in range(8, 12):
            s[i] = wires[i]
        s = p2_external_circuit(b, s)
        for r in range(4):
            s = [b.add_const_ext(s[i], Kc["POSEIDON2_RC_EXT"][12 * r + i]) for i in range(12)]
            if r:
                for i in range(12):
                    w = wires[29 + 12 * (r - 1) + i]
                    out.append(b.sub_ext(s[i], w))
                    s[i] = w
            s = p2_external_circuit(b, [pow7_circuit(b, x) for x in s])
        for r in range(22):
            s[0] = b.add_const_ext(s[0], Kc["POSEIDON2_RC_INT"][r])
            w = wires[65 + r]
            out.append(b.sub_ext(s[0], w))
            s[0] = pow7_circuit(b, w)
            s = p2_internal_circuit(b, s)
        for r in range(4):
            for i in range(12):
                s[i] = b.add_const_ext(s[i], Kc["POSEIDON2_RC_EXT"][12 * (4 + r) + i])
                w = wires[87 + 12 * r + i]
                out.append(b.sub_ext(s[i], w))
                s[i] = pow7_circuit(b, w)
            s = p2_external_circuit(b, s)
        for i in range(12):
            out.append(b.sub_ext(s[i], wires[12 + i]))
        return out
    if k in (C.REDUCING, C.REDUCING_EXT):
        nc, ext = g.p0, k == C.REDUCING_EXT
        start_accs = 6 + (2 * nc if ext else nc)
        alpha, acc = (wires[2], wires[3]), (wires[4], wires[5])
        for i in range(nc):
            nxt = (wires[0], wires[1]) if i == nc - 1 else (wires[start_accs + 2 * i], wires[start_accs + 2 * i + 1])
            # acc alpha + coeff - next, the addends riding on the products
            if ext:
                z = (b.sub_ext(wires[6 + 2 * i], nxt[0]), b.sub_ext(wires[7 + 2 * i], nxt[1]))
            else:
                z = (b.sub_ext(wires[6 + i], nxt[0]), b.mul_const_ext(P - 1, nxt[1]))
            out += list(alg_mul_add(b, acc, alpha, z))
            acc = nxt
        return out
    if k == C.RANDOM_ACCESS:
        bits, copies, extra = g.p0, g.p1, g.p2
        vs = 1 << bits
        routed = (2 + vs) * copies + extra
        for c in range(copies):
            w0, b0 = (2 + vs) * c, routed + c * bits
            bt = [wires[b0 + i] for i in range(bits)]
            for x in bt:
                out.append(b.mul_sub_ext(x, x, x))
            idx = b.zero_ext()
            for x in reversed(bt):
                idx = b.mul_const_add_ext(2, idx, x)
            out.append(b.sub_ext(idx, wires[w0]))
            items = [wires[w0 + 2 + i] for i in range(vs)]
            for x in bt:
                items = [b.mul_add_ext(x, b.sub_ext(items[2 * j + 1], items[2 * j]), items[2 * j]) for j in range(len(items) // 2)]
            out.append(b.sub_ext(items[0], wires[w0 + 1]))
        for i in range(extra):
            out.append(b.sub_ext(consts[i], wires[(2 + vs) * copies + i]))
        return out
    if k == C.COSET_INTERPOLATION:
        npts, deg = 1 << g.p0, g.p1
        nint = (npts - 2) // (deg - 1)
        w_pt, w_val = 1 + 2 * npts, 3 + 2 * npts
        w_int = w_val + 2
        w_sh = w_int + 4 * nint
        at = lambda i: (wires[i], wires[i + 1])
        pt, sh = at(w_pt), at(w_sh)
        out += [b.sub_ext(pt[0], b.mul_ext(sh[0], wires[0])), b.sub_ext(pt[1], b.mul_ext(sh[1], wires[0]))]
        om = root_of_unity(g.p0)
        dom = [pow(om, i, P) for i in range(npts)]
        bw = []
        for i in range(npts):
            pr = 1
            for j in range(npts):
                if j != i:
                    pr = pr * (dom[i] - dom[j]) % P
            bw.append(inv(pr))
        ev, pr = (b.zero_ext(), b.zero_ext()), (b.one_ext(), b.zero_ext())
        start, end = 0, deg
        for c in range(nint + 1):
            for i in range(start, end):
                val = tuple(b.mul_const_ext(bw[i], x) for x in at(1 + 2 * i))
                term = (b.add_const_ext(sh[0], P - dom[i]), sh[1])
                ev, pr = alg_mul_add(b, ev, term, alg_mul(b, val, pr)), alg_mul(b, pr, term)
            if c == nint:
                break
            iev, ipr = at(w_int + 2 * c), at(w_int + 2 * (nint + c))
            out += [b.sub_ext(iev[0], ev[0]), b.sub_ext(iev[1], ev[1]), b.sub_ext(ipr[0], pr[0]), b.sub_ext(ipr[1], pr[1])]
            ev, pr = iev, ipr
            start = 1 + (deg - 1) * (c + 1)
            end = min(start + deg - 1, npts)
        val = at(w_val)
        out += [b.sub_ext(val[0], ev[0]), b.sub_ext(val[1], ev[1])]
        return out
    if k == C.MUL_EXT:  # gates/multiplication_extension.rs: 3 D wires per operation
        for i in range(g.p0):
            m0, m1, o = [(wires[6 * i + 2 * j], wires[6 * i + 2 * j + 1]) for j in range(3)]
            pr = alg_mul(b, m0, m1)
            out += [b.sub_ext(o[j], b.mul_ext(pr[j], consts[0])) for j in range(2)]
        return out
    if k == C.EXPONENTIATION:  # gates/exponentiation.rs: base 0, bits 1..nb (most significant first in the loop), output nb+1, intermediates
        nb = g.p0
        base, one = wires[0], b.one_ext()
        for i in range(nb):
            prev = one if i == 0 else b.mul_ext(wires[nb + 2 + i - 1], wires[nb + 2 + i - 1])
            bit = wires[1 + (nb - 1 - i)]
            mulby = b.add_ext(b.mul_ext(bit, base), b.sub_ext(one, bit))
            out.append(b.mul_sub_ext(prev, mulby, wires[nb + 2 + i]))
        out.append(b.sub_ext(wires[nb + 1], wires[nb + 2 + nb - 1]))
        return out
    if k in (C.U32_ARITHMETIC, C.U32_RANGE_CHECK, C.U32_SUBTRACTION, C.U32_ADD_MANY, C.COMPARISON):
        return eval_u32_gate_circuit(b, g, wires)
    raise NotImplementedError(f"no in-circuit evaluator for gate kind {k}")


def _limb_range(b, limb, base):
    """limb (limb - 1) ... (limb - (base - 1))"""
    pr = limb
    for x in range(1, base):
        pr = b.mul_ext(pr, b.add_const_ext(limb, P - x))
    return pr


def _limb_sum(b, limbs, base):
    """sum_j limbs[j] base^j by Horner from the most significant limb"""
    acc = b.zero_ext()
    for x in reversed(limbs):
        acc = b.mul_const_add_ext(base, acc, x)
    return acc


def eval_u32_gate_circuit(b, g, wires):
    """the plonky2-u32 / plonky2_crypto gates a leaf circuit of the reference carries (mp2-common/src/serialization/
    circuit_data_serialization.rs:254-262), constraint by constraint in the order of oracle/gates_body.inc (the prover's evaluators)"""
    k, out = g.kind, []
    one, two32 = b.one_ext(), 1 << 32
    if k == C.U32_ARITHMETIC:  # per op m0, m1, addend, output_low, output_high, inverse; 32 two-bit limbs of the 64-bit output
        ops = g.p0
        for i in range(ops):
            m0, m1, ad, lo, hi, inv_w = wires[6 * i:6 * i + 6]
            computed = b.mul_add_ext(m0, m1, ad)
            diff = b.sub_ext(b.constant_ext((0xFFFFFFFF, 0)), hi)
            hi_not_max = b.mul_sub_ext(inv_w, diff, one)
            out.append(b.mul_ext(hi_not_max, lo))
            out.append(b.sub_ext(b.mul_const_add_ext(two32, hi, lo), computed))
            limbs = [wires[6 * ops + 32 * i + j] for j in range(32)]
            out += [_limb_range(b, limbs[j], 4) for j in reversed(range(32))]
            out.append(b.sub_ext(_limb_sum(b, limbs[:16], 4), lo))
            out.append(b.sub_ext(_limb_sum(b, limbs[16:], 4), hi))
        return out
    if k == C.U32_RANGE_CHECK:  # inputs 0..k, 16 two-bit limbs per input after them
        n_in = g.p0
        for i in range(n_in):
            limbs = [wires[n_in + 16 * i + j] for j in range(16)]
            out.append(b.sub_ext(_limb_sum(b, limbs, 4), wires[i]))
            out += [_limb_range(b, x, 4) for x in limbs]
        return out
    if k == C.U32_SUBTRACTION:  # per op x, y, borrow_in, result, borrow_out; 16 two-bit limbs of the result
        ops = g.p0
        for i in range(ops):
            x, y, bi, res, bo = wires[5 * i:5 * i + 5]
            initial = b.sub_ext(b.sub_ext(x, y), bi)
            out.append(b.sub_ext(res, b.mul_const_add_ext(two32, bo, initial)))
            limbs = [wires[5 * ops + 16 * i + j] for j in range(16)]
            out += [_limb_range(b, limbs[j], 4) for j in reversed(range(16))]
            out.append(b.sub_ext(_limb_sum(b, limbs, 4), res))
            out.append(b.mul_ext(bo, b.sub_ext(one, bo)))
        return out
    if k == C.U32_ADD_MANY:  # per op p0 addends, carry_in, result, carry_out; 16 result + 2 carry limbs
        na, ops, per = g.p0, g.p1, g.p0 + 3
        for i in range(ops):
            w = wires[per * i:per * i + per]
            computed = w[na]
            for j in range(na):
                computed = b.add_ext(computed, w[j])
            res, co = w[na + 1], w[na + 2]
            out.append(b.sub_ext(b.mul_const_add_ext(two32, co, res), computed))
            limbs = [wires[per * ops + 18 * i + j] for j in range(18)]
            out += [_limb_range(b, limbs[j], 4) for j in reversed(range(18))]
            out.append(b.sub_ext(_limb_sum(b, limbs[:16], 4), res))
            out.append(b.sub_ext(_limb_sum(b, limbs[16:], 4), co))
        return out
    # ComparisonGate: first <= second over p0 bits in p1 chunks
    nc = g.p1
    cb = (g.p0 + nc - 1) // nc
    cs = 1 << cb
    fc, sc, ed, ce, iv = [[wires[4 + q * nc + i] for i in range(nc)] for q in range(5)]
    bits = [wires[4 + 5 * nc + i] for i in range(cb + 1)]
    out.append(b.sub_ext(_limb_sum(b, fc, cs), wires[0]))
    out.append(b.sub_ext(_limb_sum(b, sc, cs), wires[1]))
    msd = b.zero_ext()
    for i in range(nc):
        out.append(_limb_range(b, fc[i], cs))
        out.append(_limb_range(b, sc[i], cs))
        diff = b.sub_ext(sc[i], fc[i])
        out.append(b.sub_ext(b.mul_ext(diff, ed[i]), b.sub_ext(one, ce[i])))
        out.append(b.mul_ext(ce[i], diff))
        out.append(b.sub_ext(iv[i], b.mul_ext(ce[i], msd)))
        msd = b.mul_add_ext(b.sub_ext(one, ce[i]), diff, iv[i])
    out.append(b.sub_ext(wires[3], msd))
    bc = _limb_sum(b, bits, 2)
    out += [b.mul_ext(x, b.sub_ext(one, x)) for x in bits]
    out.append(b.sub_ext(b.add_const_ext(wires[3], cs), bc))
    out.append(b.sub_ext(wires[2], bits[cb]))
    return out


def alg_mul(b, x, y):
    """ExtensionAlgebra product (a0 + a1 X)(b0 + b1 X), X^2 = 7, over extension targets"""
    a = b.mul_add_ext(x[0], y[0], b.arithmetic_ext(W7, x[1], y[1], 0, x[1]))
    return (a, b.mul_add_ext(x[0], y[1], b.mul_ext(x[1], y[0])))


def alg_mul_add(b, x, y, z):
    """x y + z in the extension algebra, the addend folded into the products' own operations"""
    a = b.mul_add_ext(x[0], y[0], b.arithmetic_ext(W7, x[1], y[1], 1, z[0]))
    return (a, b.mul_add_ext(x[0], y[1], b.mul_add_ext(x[1], y[0], z[1])))


def alg_add(b, x, y):
    return (b.add_ext(x[0], y[0]), b.add_ext(x[1], y[1]))


def pow7_circuit(b, x):
    x2 = b.mul_ext(x, x)
    x4 = b.mul_ext(x2, x2)
    return b.mul_ext(b.mul_ext(x, x2), x4)


def p2_external_circuit(b, s):
    """circ(2 M4, M4, M4) with M4 by the addition chain of HorizenLabs' matmul_m4 (8 operations per block instead of 16
    multiply-adds): t0 = a + b, t1 = c + d, t2 = 2b + t1, t3 = 2d + t0, t4 = 4 t1 + t3, t5 = 4 t0 + t2 -> (t3 + t5, t5, t2 + t4, t4)"""
    t = []
    for c in range(3):
        x0, x1, x2, x3 = s[4 * c:4 * c + 4]
        t0, t1 = b.add_ext(x0, x1), b.add_ext(x2, x3)
        t2, t3 = b.mul_const_add_ext(2, x1, t1), b.mul_const_add_ext(2, x3, t0)
        t4, t5 = b.mul_const_add_ext(4, t1, t3), b.mul_const_add_ext(4, t0, t2)
        t += [b.add_ext(t3, t5), t5, b.add_ext(t2, t4), t4]
    sums = [b.add_ext(b.add_ext(t[i], t[4 + i]), t[8 + i]) for i in range(4)]
    return [b.add_ext(t[4 * c + i], sums[i]) for c in range(3) for i in range(4)]


def p2_internal_circuit(b, s):
    d = K2()["POSEIDON2_DIAG_M1"]
    tot = s[0]
    for x in s[1:]:
        tot = b.add_ext(tot, x)
    return [b.mul_const_add_ext(d[i], s[i], tot) for i in range(12)]


# ---- the verifier ([dep] plonk/recursive_verifier.rs verify_proof, fri/recursive_verifier.rs) -------------------------------------------------
class InnerCircuit:
    """what the verifier circuit needs to know about the circuit whose proofs it checks: CommonCircuitData (FRI
    parameters, gate table with selector groups, constant / wire / routed counts) and, as constants of the wrap
    circuit, VerifierOnlyCircuitData (constants_sigmas cap, circuit digest)."""

    def __init__(self, ckt, fp, constants_sigmas_cap, circuit_digest, n_public_inputs):
        self.ckt, self.fp = ckt, fp
        self.cap = [[int(x) for x in h] for h in np.asarray(constants_sigmas_cap).reshape(-1, 4)]
        self.circuit_digest = [int(x) for x in circuit_digest]
        self.n_public_inputs = n_public_inputs


def verify_proof_circuit(b, inner, caps, openings, fri, public_inputs, verifier_data=None):
    """Adds the constraints `proof is a valid proof of `inner` with these public inputs` to builder b and returns
    the public-input targets. caps [n_oracles][16][4] (oracle 0 ignored: it belongs to the verifier data), openings
    [n_open][2] (FRI batch order), fri = the flat FriProof words (include/mp2g.h layout), as numpy / int arrays.
    verifier_data = (cap targets [16][4], circuit digest targets [4]) for a universal verifier whose verifier data are
    witnesses; None = the constants of inner (verify_proof_fixed_circuit, the first wrap step)."""
    fp, ckt = inner.fp, inner.ckt
    assert fp.num_lookup_polys == 0, "lookup tables: not in the recursive verifier yet"
    k, lg = fp.log_n, fp.log_n + fp.rate_bits
    n = 1 << k
    nc = fp.zs_count
    capn = 1 << fp.cap_height
    ws = [fp.oracle_w[o] for o in range(4)]
    num_routed, degree = NUM_ROUTED, 8
    chunks = num_routed // degree
    num_consts = ws[0] - num_routed
    V, VE = b.add_virtual, b.add_virtual_ext
    # ---- proof targets (add_virtual_proof_with_pis) and the constant verifier data
    pis = [V(int(x)) for x in public_inputs]
    cap_t = [None] + [[[V(int(x)) for x in np.asarray(caps[o]).reshape(capn, 4)[h]] for h in range(capn)] for o in range(1, 4)]
    if verifier_data is None:
        cap_t[0] = [[b.constant(x) for x in h] for h in inner.cap]
        digest = [b.constant(x) for x in inner.circuit_digest]
    else:
        cap_t[0], digest = verifier_data
    op = [VE((int(e[0]), int(e[1]))) for e in openings]
    o_w, o_z, o_q, o_next = ws[0], ws[0] + ws[1], ws[0] + ws[1] + ws[2], sum(ws)
    fri = [int(x) for x in fri]
    pos = 0

    def take(cnt):
        nonlocal pos
        out = [V(x) for x in fri[pos:pos + cnt]]
        pos += cnt
        return out
    hashes = lambda ts: [ts[4 * i:4 * i + 4] for i in range(len(ts) // 4)]
    exts = lambda ts: [E(ts[2 * i], ts[2 * i + 1]) for i in range(len(ts) // 2)]
    commit_caps = [hashes(take(4 * capn)) for _ in range(fp.n_layers)]
    rounds = []
    for _ in range(fp.num_queries):
        init = []
        for o in range(4):
            leaf = take(ws[o])
            init.append((leaf, hashes(take(4 * (lg - fp.cap_height)))))
        steps, clg = [], lg
        for i in range(fp.n_layers):
            ab = fp.arity_bits[i]
            clg -= ab
            ev = exts(take(2 << ab))
            steps.append((ev, hashes(take(4 * (clg - fp.cap_height)))))
        rounds.append((init, steps))
    deg_bits = k - sum(fp.arity_bits[i] for i in range(fp.n_layers))
    final_poly = exts(take(2 << deg_bits))
    pow_witness = take(1)[0]
    assert pos == len(fri)

    # ---- challenges (plonk/get_challenges.rs)
    pi_hash = b.hash_n_to_m_no_pad(pis, 4)
    ch = RecursiveChallenger(b)
    ch.observe(digest)
    ch.observe(pi_hash)
    flat = lambda cap: [t for h in cap for t in h]
    ch.observe(flat(cap_t[1]))
    betas, gammas = ch.get_n(nc), ch.get_n(nc)
    ch.observe(flat(cap_t[2]))
    alphas = ch.get_n(nc)
    ch.observe(flat(cap_t[3]))
    zeta = ch.get_ext()
    ch.observe_ext(op)
    fri_alpha = ch.get_ext()
    fri_betas = []
    for cap in commit_caps:
        ch.observe(flat(cap))
        fri_betas.append(ch.get_ext())
    ch.observe_ext(final_poly)
    ch.observe([pow_witness])
    pow_response = ch.get()
    query_indices = ch.get_n(fp.num_queries)

    # ---- the PLONK identity at zeta (plonk/vanishing_poly.rs eval_vanishing_poly_circuit)
    zeta_n = b.exp_power_of_2_ext(zeta, k)
    z_h = b.add_const_ext(zeta_n, P - 1)
    l0 = b.div_ext(z_h, b.mul_const_ext(n % P, b.add_const_ext(zeta, P - 1)))
    consts_o, sigmas_o = op[:num_consts], op[num_consts:o_w]
    wires_o = op[o_w:o_z]
    zs_o, pps_o = op[o_z:o_z + nc], op[o_z + nc:o_q]
    quot_o = op[o_q:o_next]
    zs_next = op[o_next:o_next + nc]
    terms = [b.mul_ext(l0, b.add_const_ext(zs_o[c], P - 1)) for c in range(nc)]
    k_is = [pow(MULT_GEN, j, P) for j in range(num_routed)]
    num_prods = chunks - 1
    for c in range(nc):
        beta_e, gamma_e = b.to_ext(betas[c]), b.to_ext(gammas[c])
        beta_zeta = b.mul_ext(beta_e, zeta)
        for chn in range(chunks):
            num, den = None, None
            for j in range(chn * degree, (chn + 1) * degree):
                wg = b.add_ext(wires_o[j], gamma_e)
                f_num = b.mul_const_add_ext(k_is[j], beta_zeta, wg)
                f_den = b.mul_add_ext(beta_e, sigmas_o[j], wg)
                num = f_num if num is None else b.mul_ext(num, f_num)
                den = f_den if den is None else b.mul_ext(den, f_den)
            prev = zs_o[c] if chn == 0 else pps_o[c * num_prods + chn - 1]
            nxt = zs_next[c] if chn == chunks - 1 else pps_o[c * num_prods + chn]
            terms.append(b.mul_sub_ext(prev, num, b.mul_ext(nxt, den)))
    # gate constraints. plonky2 sums filter_g c_{g,j} into slot j and alpha-reduces the slots; the same field element is
    # sum_g filter_g (sum_j alpha^j c_{g,j}) alpha^(#permutation terms): each gate's constraints are alpha-reduced first
    # (ReducingExtensionGate rows, 32 coefficients each), the filter multiplies the reduced value once.
    gates = ckt.gates
    nsel = ckt.num_selectors
    per_gate = []
    for gi, g in enumerate(gates):
        cons = eval_gate_circuit(b, g, consts_o[nsel:], wires_o, pi_hash)
        if not cons:
            continue
        s_sel = consts_o[g.selector_index]
        filt = None
        for r in range(g.group_start, g.group_end):
            if r != gi:
                f = b.arithmetic_ext(P - 1, s_sel, b.one_ext(), r, b.one_ext())  # r - s
                filt = f if filt is None else b.mul_ext(filt, f)
        if nsel > 1:
            f = b.add_const_ext(b.mul_const_ext(P - 1, s_sel), 0xFFFFFFFF)  # UNUSED_SELECTOR - s
            filt = f if filt is None else b.mul_ext(filt, f)
        per_gate.append((filt, cons))
    for a in range(nc):
        alpha_e = b.to_ext(alphas[a])
        gsum = None
        for filt, cons in per_gate:
            r = b.reduce_ext(alpha_e, cons) if len(cons) > 8 else b.reduce_with_powers_ext(cons, alpha_e)
            v = r if filt is None else b.mul_ext(filt, r)
            gsum = v if gsum is None else b.add_ext(gsum, v)
        van = b.reduce_with_powers_ext(terms + ([gsum] if gsum is not None else []), alpha_e)
        tz = b.reduce_with_powers_ext(quot_o[8 * a:8 * a + 8], zeta_n)
        b.connect_ext(van, b.mul_ext(z_h, tz))

    # ---- FRI (fri/recursive_verifier.rs verify_fri_proof)
    # proof of work: the response has pow_bits leading zeros
    b.range_check(pow_response, 64 - fp.pow_bits)
    # precomputed reduced openings: sum_j alpha^j v_j per batch
    n_zeta = o_next
    red = [b.reduce_ext(fri_alpha, op[:n_zeta]), b.reduce_with_powers_ext(op[n_zeta:], fri_alpha)]
    g_k = root_of_unity(k)
    zeta_next = b.mul_const_ext(g_k, zeta)
    alpha_pow_next = fri_alpha
    for _ in range(len(op) - n_zeta - 1):
        alpha_pow_next = b.mul_ext(alpha_pow_next, fri_alpha)  # alpha^(number of polynomials in the g*zeta batch)
    w_lg = root_of_unity(lg)
    with b.parallel_sections() as region:
        for q in range(fp.num_queries):
            with region.section():  # the query rounds read the challenges and reduced openings above and nothing of each other
                init, steps = rounds[q]
                bits = b.split_le(query_indices[q], 64)[:lg]
                cap_index = b.le_sum(bits[lg - fp.cap_height:lg])
                for o in range(4):
                    leaf, sib = init[o]
                    verify_merkle_proof_to_cap(b, leaf, bits[:lg - fp.cap_height], cap_index, cap_t[o], sib)
                # subgroup_x = g * w^(bit-reversed index)
                sx = b.mul_const(MULT_GEN, b.exp_from_bits_const_base(w_lg, list(reversed(bits))))
                # fri_combine_initial: (reduce(all leaf evals) - red0) / (x - zeta), shifted, + the g*zeta batch
                leaf_all = [t for o in range(4) for t in init[o][0]]
                sx_e = b.to_ext(sx)
                num0 = b.sub_ext(b.reduce_base(fri_alpha, leaf_all), red[0])
                q0 = b.div_ext(num0, b.sub_ext(sx_e, zeta))
                num1 = b.sub_ext(b.reduce_base(fri_alpha, init[2][0][:nc]) if nc > 21 else
                                 b.reduce_with_powers_ext([b.to_ext(t) for t in init[2][0][:nc]], fri_alpha), red[1])
                q1 = b.div_ext(num1, b.sub_ext(sx_e, zeta_next))
                old_eval = b.mul_add_ext(q0, alpha_pow_next, q1)
                xbits = bits
                clg = lg
                for i in range(fp.n_layers):
                    ab = fp.arity_bits[i]
                    evals, sib = steps[i]
                    within_bits, coset_bits = xbits[:ab], xbits[ab:]
                    within = b.le_sum(within_bits)
                    b.connect_ext(b.random_access_ext(within, evals), old_eval)
                    # compute_evaluation: interpolate the coset at beta
                    g_ab = root_of_unity(ab)
                    rev = list(evals)
                    rev = [rev[int(format(j, f"0{ab}b")[::-1], 2)] for j in range(1 << ab)]
                    start = b.exp_from_bits_const_base(inv(g_ab), list(reversed(within_bits)))
                    coset_start = b.mul(start, sx)
                    old_eval = b.interpolate_coset(ab, coset_start, rev, fri_betas[i])
                    clg -= ab
                    leaf = [t for e in evals for t in (e.a, e.b)]
                    verify_merkle_proof_to_cap(b, leaf, coset_bits[:clg - fp.cap_height], cap_index, commit_caps[i], sib)
                    sx = b.exp_power_of_2(sx, ab)
                    xbits = coset_bits
                # final polynomial at the folded point
                fe = b.reduce_ext(b.to_ext(sx), final_poly) if len(final_poly) > 12 else b.reduce_with_powers_ext(final_poly, b.to_ext(sx))
                b.connect_ext(fe, old_eval)
    return pis


# ---- the circuits of recursion-framework/tests/integration.rs -------------------------------------------------------------------
def map_circuit(inputs, builder=None):
    """MapCircuitWires::circuit_logic (integration.rs:75-93): public inputs = (sum of the even inputs, H(inputs))"""
    b = builder or Builder()
    ins = [b.add_virtual(int(x)) for x in inputs]
    one = b.one()
    acc = b.zero()
    for t in ins:
        is_odd = b.split_le(t, 64)[0]
        acc = b.mul_add(b.sub(one, is_odd), t, acc)
    b.register_public_inputs([acc] + b.hash_n_to_m_no_pad(ins, 4))
    return b.build()


def wrap_circuit(inner, caps, openings, fri, public_inputs, strict=True):
    """WrapCircuit::build_wrap_circuit, first step (wrap_circuit.rs:58-99): verify the inner proof with the inner
    circuit's verifier data as constants and expose the inner proof's public inputs"""
    b = Builder(strict)
    pis = verify_proof_circuit(b, inner, caps, openings, fri, public_inputs)
    b.register_public_inputs(pis)
    return b.build()


# ---- the recursion framework (recursion-framework/src: circuit_builder.rs, universal_verifier_gadget/*, framework.rs) --------------------
# Every circuit of the framework is wrapped until its proof has the shape all the others have, so that one "universal"
# verifier (verifier data as witnesses + membership of their digest in the circuit set) can check any of them. The fixed
# point is RECURSION_THRESHOLD = 12 row bits as in the reference (universal_verifier_gadget/mod.rs:34): the verifier built
# here needs 3847 rows for a 2^12-row standard-config proof and 4022 for a 2^13-row one (its constants ride in the
# RandomAccessGate rows' spare constant slots, so it has no ConstantGate rows at all), a reduce circuit with two universal
# verifiers fits 2^13 rows, and its wrap lands on 2^12 again.
RECURSION_THRESHOLD = 12
CIRCUIT_SET_CAP_HEIGHT = 0
DOMAIN_SEPARATOR_PAD = [1, 0, 0, 0, 0, 0, 0, 1]  # hash_pad(&[]) input


def common_data(ckt):
    """what must be equal for two circuits to share a universal verifier: CommonCircuitData (here: degree, gate table
    with its selector groups, number of constants)"""
    return (ckt.log_n, ckt.num_selectors, ckt.num_constants,
            tuple((g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end) for g in ckt.gates))


def check_circuit_digest(b, cap_t, digest_t, degree_bits):
    """circuit_set.rs:136-158 check_circuit_digest_target: digest == H(cap || H_pad([]) || degree_bits)"""
    dom = b.hash_n_to_m_no_pad([b.constant(x) for x in DOMAIN_SEPARATOR_PAD], 4)
    h = b.hash_n_to_m_no_pad([t for hh in cap_t for t in hh] + dom + [b.constant(degree_bits)], 4)
    for x, y in zip(h, digest_t):
        b.connect(x, y)


def universal_verifier_circuit(b, rec, circuit_set_t, set_size, proof, vd, membership):
    """verifier_gadget.rs:129-166: verify `proof` (caps, openings, fri, public_inputs) of a circuit with common data
    `rec` under verifier data given as witnesses vd = (cap [16][4], digest [4]); check the digest against the cap,
    its membership in the circuit set (index bits little-endian, siblings bottom-up) and that the proof exposes the
    same circuit set. Returns the proof's public-input targets."""
    caps, openings, fri, public_inputs = proof
    cap_t = [[b.add_virtual(int(x)) for x in h] for h in np.asarray(vd[0]).reshape(-1, 4)]
    digest_t = [b.add_virtual(int(x)) for x in vd[1]]
    pis = verify_proof_circuit(b, rec, caps, openings, fri, public_inputs, verifier_data=(cap_t, digest_t))
    check_circuit_digest(b, cap_t, digest_t, RECURSION_THRESHOLD)
    bits, siblings = membership
    height = max(0, (set_size - 1).bit_length()) - CIRCUIT_SET_CAP_HEIGHT
    assert len(bits) == height and len(siblings) == height
    bit_t = []
    for x in bits:
        t = b.add_virtual(int(x))
        b.assert_bool(t)  # add_virtual_bool_target_safe
        bit_t.append(t)
    sib_t = [[b.add_virtual(int(x)) for x in sb] for sb in siblings]
    verify_merkle_proof_to_cap(b, digest_t, bit_t, None, [circuit_set_t], sib_t)
    n_own = len(pis) - 4
    for x, y in zip(circuit_set_t, pis[n_own:]):
        b.connect(x, y)
    return pis


class FrameworkCircuit:
    """CircuitWithUniversalVerifier (circuit_builder.rs:264-311): NUM_VERIFIERS universal verifiers, the circuit's
    own logic, the circuit-set digest as the last public inputs; then the wrap chain down to the threshold shape.
    `logic(b, child_public_inputs, inputs)` returns the circuit's own public-input targets (CircuitLogicWires)."""

    def __init__(self, name, num_verifiers, logic, num_public_inputs, min_log_n=6, extra_gates=()):
        """min_log_n / extra_gates: pad the base circuit with no-op rows to 2^min_log_n rows and give it one row of every gate of
        `extra_gates` it lacks (Builder.add_gate_rows) -- the knobs of the base-degree sweep (the reference's real base
        degrees come from CircuitWithUniversalVerifier::wrapped_circuit_size, circuit_builder.rs:323-325, and are 12..15)"""
        self.name, self.num_verifiers, self.logic, self.num_public_inputs = name, num_verifiers, logic, num_public_inputs
        self.min_log_n, self.extra_gates = max(6, int(min_log_n)), tuple(extra_gates)

    def build_base(self, fw, child_proofs, child_vds, memberships, inputs, set_digest, strict=True):
        if self.num_verifiers and fw.rec.fp.num_lookup_polys:
            # verify_proof_circuit has no in-circuit lookup argument (check_lookup_constraints_circuit): a circuit with lookup
            # tables can be proved by prove() but not wrapped or checked by the universal verifier (Limits)
            raise NotImplementedError("the in-circuit verifier does not check the lookup argument: circuits with lookup tables cannot enter the framework")
        b = Builder(strict)
        set_t = [b.add_virtual(int(x)) for x in set_digest]  # CircuitSetTarget::build_target: a virtual cap of height 0
        child_pis = []
        for proof, vd, mem in zip(child_proofs, child_vds, memberships):
            pis = universal_verifier_circuit(b, fw.rec, set_t, fw.set_size, proof, vd, mem)
            child_pis.append(pis[:len(pis) - 4])
        own = self.logic(b, child_pis, inputs)
        assert len(own) == self.num_public_inputs
        b.register_public_inputs(list(own) + set_t)
        if self.extra_gates:
            b.add_gate_rows(self.extra_gates)
        return b.build(min_log_n=self.min_log_n)


def build_wrap_chain(prover, fri_params, base_vd):
    """WrapCircuit::build_wrap_circuit (universal_verifier_gadget/wrap_circuit.rs:45-120): the circuits that re-prove a
    proof of `base_vd` = (circuit, constants_sigmas cap, circuit digest) until the threshold size is reached, built over
    dummy proofs; returns [(circuit, cap, digest)] per wrap step"""
    chain = []
    cur = base_vd
    for _ in range(4):
        inner = InnerCircuit(cur[0], fri_params(cur[0]), cur[1], cur[2], len(cur[0].public_inputs))
        b = Builder(strict=False)
        pis = verify_proof_circuit(b, inner, *dummy_proof(inner))
        b.register_public_inputs(pis)
        w = b.build(min_log_n=RECURSION_THRESHOLD)
        cap, digest = prover.verifier_data(w)
        cur = (w, cap, digest)
        chain.append(cur)
        if w.log_n == RECURSION_THRESHOLD:
            return chain
    raise AssertionError("the wrap chain does not reach the threshold size")


def wrap_proof_chain(prover, fri_params, chain, base, proof):
    """WrapCircuit::wrap_proof (wrap_circuit.rs:122-148): chain = [base verifier data] + build_wrap_chain(...); every step
    fills the verifier circuit with the previous proof and proves it. A proof the step's verifier does not accept makes the
    witness inconsistent: the strict builder raises, as the reference's prove() panics."""
    cur_ckt, cur_proof = base, proof
    for step, (wckt, wcap, wdig) in enumerate(chain[1:]):
        prev = chain[step]
        inner = InnerCircuit(cur_ckt, fri_params(cur_ckt), prev[1], prev[2], len(cur_ckt.public_inputs))
        b = Builder()
        pis = verify_proof_circuit(b, inner, *cur_proof)
        b.register_public_inputs(pis)
        w = b.build(min_log_n=RECURSION_THRESHOLD)
        assert np.array_equal(w.pre, wckt.pre)
        caps, openings, proof = prover.prove(w)
        cur_ckt, cur_proof = w, (caps, openings, proof, w.public_inputs)
    return cur_proof


def dummy_proof(inner):
    fp = inner.fp
    n_open = sum(fp.oracle_w[o] for o in range(4)) + fp.zs_count
    return (np.zeros((4, 4 << fp.cap_height), dtype=np.uint64), np.zeros((n_open, 2), dtype=np.uint64),
            np.zeros(fp.proof_words, dtype=np.uint64), np.zeros(inner.n_public_inputs, dtype=np.uint64))


def dummy_circuit(num_gates_log, num_public_inputs):
    """universal_verifier_gadget/mod.rs:47-63: `num_public_inputs` free public inputs, padded with no-ops to 2^num_gates_log rows"""
    b = Builder()
    b.register_public_inputs([b.add_virtual(0) for _ in range(num_public_inputs)])
    return b.build(min_log_n=num_gates_log)


class WrapCircuit:
    """universal_verifier_gadget/wrap_circuit.rs:29-155: the chain of circuits shrinking proofs of one base circuit to the
    RECURSION_THRESHOLD size. prover / fri_params as for RecursiveCircuits."""

    def __init__(self, base, prover, fri_params):
        self.prover, self.fri_params = prover, fri_params
        cap, digest = prover.verifier_data(base)
        self.chain = [(base, cap, digest)] + build_wrap_chain(prover, fri_params, (base, cap, digest))

    def final_proof_circuit_data(self):
        """(circuit, constants_sigmas cap, circuit digest) of the last wrap step"""
        return self.chain[-1]

    def wrap_proof(self, base, proof):
        """proof = (caps, openings, fri, public_inputs) of `base` (the witness-filled instance of the base circuit)"""
        assert np.array_equal(base.pre, self.chain[0][0].pre), "not a proof of this wrap circuit's base circuit"
        return wrap_proof_chain(self.prover, self.fri_params, self.chain, base, proof)


class RecursiveCircuits:
    """framework.rs RecursiveCircuits + the per-circuit wrap chains: build every circuit's structure once (dummy
    witnesses), collect the digests of the final wrap circuits into the circuit set, then generate_proof().
    prover(ckt) -> (caps, openings, proof, constants_sigmas_cap, circuit_digest) is the proving back end (the HIP
    prover in production, the oracle in the CPU tests)."""

    def __init__(self, circuits, prover, fri_params):
        self.prover, self.fri_params = prover, fri_params
        self.circuits = {c.name: c for c in circuits}
        self.set_size = len(circuits)
        # common data of every final wrap circuit: wrap a dummy proof of a dummy circuit, as
        # build_data_for_universal_verifier does (universal_verifier_gadget/mod.rs:66-87)
        n_pi = circuits[0].num_public_inputs + 4
        assert all(c.num_public_inputs + 4 == n_pi for c in circuits)
        dummy = self._dummy_circuit(n_pi)
        chain = self._wrap_structure(dummy)
        self.rec = InnerCircuit(chain[-1][0], fri_params(chain[-1][0]), chain[-1][1], chain[-1][2], n_pi)
        self.rec_common = common_data(chain[-1][0])
        # structure of every circuit and of its wrap chain; verifier data of the final wrap circuit
        self.chains, self.vds = {}, {}
        zero_set = [0, 0, 0, 0]
        for c in circuits:
            proofs = [self._dummy_proof(self.rec) for _ in range(c.num_verifiers)]
            vds = [(np.zeros((16, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64))] * c.num_verifiers
            height = max(0, (self.set_size - 1).bit_length())
            mems = [([0] * height, [[0, 0, 0, 0]] * height)] * c.num_verifiers
            base = c.build_base(self, proofs, vds, mems, None, zero_set, strict=False)
            chain = self._wrap_structure(base)
            assert common_data(chain[-1][0]) == self.rec_common, f"{c.name}: the final wrap circuit does not have the shared shape"
            self.chains[c.name] = [self._verifier_data(base)] + chain
            self.vds[c.name] = (chain[-1][1], chain[-1][2])
        # the circuit set: Merkle tree (cap height 0) over the final wrap circuits' digests, padded with [0] leaves
        self.digests = [self.vds[c.name][1] for c in circuits]
        self.set_levels = self._set_tree(self.digests)
        self.set_digest = self.set_levels[-1][0]

    # -- structure helpers (dummy witnesses, strict off)
    def _dummy_circuit(self, n_pi):
        b = Builder()
        b.register_public_inputs([b.add_virtual(0) for _ in range(n_pi)])
        return b.build(min_log_n=6)

    def _verifier_data(self, ckt):
        """(circuit, constants_sigmas cap, circuit digest): the preprocessed commitment by the proving back end"""
        cap, digest = self.prover.verifier_data(ckt)
        return (ckt, cap, digest)

    def _dummy_proof(self, inner):
        return dummy_proof(inner)

    def _wrap_structure(self, base):
        return build_wrap_chain(self.prover, self.fri_params, self._verifier_data(base))

    def _set_tree(self, digests):
        size = 1 << max(0, (len(digests) - 1).bit_length())
        leaves = [[int(x) for x in d] for d in digests] + [[0, 0, 0, 0]] * (size - len(digests))  # hash_or_noop of the 1-limb pad leaf [0]
        levels = [leaves]
        while len(levels[-1]) > 1:
            prev = levels[-1]
            levels.append([self.prover.two_to_one(prev[2 * i], prev[2 * i + 1]) for i in range(len(prev) // 2)])
        return levels

    def membership(self, digest):
        """CircuitSet::set_circuit_membership_target (circuit_set.rs:205-237): little-endian index bits + siblings"""
        idx = next((i for i, d in enumerate(self.digests) if [int(x) for x in d] == [int(x) for x in digest]), None)
        if idx is None:
            raise KeyError("circuit digest not found")  # the reference's error (circuit_set.rs set_circuit_membership_target)
        bits, sib = [], []
        for lv in self.set_levels[:-1]:
            bits.append(idx & 1)
            sib.append(lv[idx ^ 1])
            idx >>= 1
        return bits, sib

    # -- proving
    def generate_proof(self, name, child_proofs, child_names, inputs):
        """RecursiveCircuits::generate_proof (framework.rs): base proof of circuit `name` over the children's final
        proofs, then its wrap chain. Returns the final proof (caps, openings, fri, public_inputs)."""
        c = self.circuits[name]
        vds = [self.vds[n] for n in child_names]
        mems = [self.membership(vd[1]) for vd in vds]
        base = c.build_base(self, child_proofs, vds, mems, inputs, self.set_digest)
        assert np.array_equal(base.pre, self.chains[name][0][0].pre), "the circuit structure depends on the witness"
        caps, openings, proof = self.prover.prove(base)
        return wrap_proof_chain(self.prover, self.fri_params, self.chains[name], base, (caps, openings, proof, base.public_inputs))


# ---- parameter files: the built framework without rebuilding it (framework.rs derives Serialize / Deserialize for RecursiveCircuits;
# the container here is numpy's .npz -- plain arrays, loaded with allow_pickle=False -- not bincode of plonky2's CircuitData)
_CKT_ARRAYS = ("pre", "tape", "input_sids", "const_slots", "pi_hash_sids", "public_input_sids", "pi_hash", "public_inputs", "input_values")
_CKT_SCALARS = ("log_n", "num_selectors", "num_constants", "pi_row", "n_slots", "n_used_rows")
PARAMS_VERSION = 1


def circuit_to_arrays(ckt, prefix, out):
    """everything of a built circuit but its witness: preprocessed polynomials, gate table with selector groups, witness program"""
    for k in _CKT_ARRAYS:
        out[prefix + k] = np.asarray(getattr(ckt, k))
    out[prefix + "scalars"] = np.array([int(getattr(ckt, k)) for k in _CKT_SCALARS], dtype=np.int64)
    out[prefix + "gates"] = np.array([[g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end] for g in ckt.gates],
                                     dtype=np.uint32).reshape(-1, 7)
    out[prefix + "instances"] = np.asarray(ckt.instances, dtype=np.uint32)
    out[prefix + "domain_separator"] = np.asarray(ckt.domain_separator, dtype=np.uint64)


def circuit_from_arrays(data, prefix):
    ckt = C.Circuit()
    for k in _CKT_ARRAYS:
        setattr(ckt, k, data[prefix + k])
    for k, v in zip(_CKT_SCALARS, data[prefix + "scalars"]):
        setattr(ckt, k, int(v))
    ckt.gates = [Gate(*(int(x) for x in row)) for row in data[prefix + "gates"]]
    ckt.gate_array = (Gate * len(ckt.gates))(*ckt.gates)
    ckt.instances = [int(x) for x in data[prefix + "instances"]]
    ckt.domain_separator = [int(x) for x in data[prefix + "domain_separator"]]
    ckt.luts, ckt.num_lookup_selectors, ckt.num_lookup_polys = [], 0, 0
    ckt.wires = None  # a parameter file holds circuits, not witnesses
    return ckt


def _params_to_bytes(self):
    """RecursiveCircuits -> bytes: every circuit's base + wrap chain (preprocessed polynomials, gate tables, witness programs),
    their verifier data, the circuit set. The circuit logic (Python callables), the prover and the FRI parameter rule are not
    part of the file: from_bytes takes them again."""
    import io
    out = {"version": np.array([PARAMS_VERSION, self.set_size], dtype=np.int64),
           "names": np.array([n.encode() for n in self.circuits]),
           "shape": np.array([[c.num_verifiers, c.num_public_inputs, len(self.chains[n])] for n, c in self.circuits.items()], dtype=np.int64),
           "set_digest": np.asarray(self.set_digest, dtype=np.uint64)}
    for n in self.circuits:
        for step, (ckt, cap, digest) in enumerate(self.chains[n]):
            pre = f"{n}/{step}/"
            circuit_to_arrays(ckt, pre, out)
            out[pre + "cap"], out[pre + "digest"] = np.asarray(cap, dtype=np.uint64), np.asarray(digest, dtype=np.uint64)
    circuit_to_arrays(self.rec.ckt, "rec/", out)
    out["rec/cap"], out["rec/digest"] = np.array(self.rec.cap, dtype=np.uint64), np.array(self.rec.circuit_digest, dtype=np.uint64)
    buf = io.BytesIO()
    np.savez_compressed(buf, **out)
    return buf.getvalue()


def _params_from_bytes(cls, data, circuits, prover, fri_params):
    """bytes of to_bytes() -> RecursiveCircuits, without building a circuit: `circuits` are the FrameworkCircuit objects (names,
    verifier counts and public-input counts must be the file's), prover / fri_params as for the constructor. The circuit set is
    re-hashed by the prover and must give the file's digest (a file made with the other hasher fails here)."""
    import io
    z = np.load(io.BytesIO(data), allow_pickle=False)
    version, set_size = (int(x) for x in z["version"])
    if version != PARAMS_VERSION:
        raise ValueError(f"parameter file version {version}, expected {PARAMS_VERSION}")
    names = [n.decode() for n in z["names"]]
    shape = z["shape"]
    if names != [c.name for c in circuits] or any((c.num_verifiers, c.num_public_inputs) != (int(s[0]), int(s[1])) for c, s in zip(circuits, shape)):
        raise ValueError("the parameter file was built for another set of circuits")
    self = cls.__new__(cls)
    self.prover, self.fri_params = prover, fri_params
    self.circuits = {c.name: c for c in circuits}
    self.set_size = set_size
    self.chains, self.vds = {}, {}
    for n, s in zip(names, shape):
        self.chains[n] = [(circuit_from_arrays(z, f"{n}/{step}/"), z[f"{n}/{step}/cap"], z[f"{n}/{step}/digest"]) for step in range(int(s[2]))]
        self.vds[n] = (self.chains[n][-1][1], self.chains[n][-1][2])
    rec = circuit_from_arrays(z, "rec/")
    self.rec = InnerCircuit(rec, fri_params(rec), z["rec/cap"], z["rec/digest"], circuits[0].num_public_inputs + 4)
    self.rec_common = common_data(rec)
    self.digests = [self.vds[n][1] for n in names]
    self.set_levels = self._set_tree(self.digests)
    self.set_digest = self.set_levels[-1][0]
    if [int(x) for x in self.set_digest] != [int(x) for x in z["set_digest"]]:
        raise ValueError("the circuit set of the parameter file does not hash to its digest with this prover")
    return self


def split_hash_element_to_low_high(b, element):
    """mp2-common/src/poseidon.rs:59-72: the low and the high 32 bits of a hash limb, with the check that makes the split unique for
    a CANONICAL field element: high = 2^32 - 1 forces low = 0 (p = 2^64 - 2^32 + 1)"""
    lo, hi = b.split_low_high(element, 32, 64)
    low_zero = b.is_equal(lo, b.zero())
    high_high = b.is_equal(hi, b.constant(0xFFFFFFFF))
    b.connect(b.or_(low_zero, b.not_(high_high)), b.one())
    return lo, hi


def flatten_poseidon_hash_target(b, h):
    """poseidon.rs:75-89: the 4 limbs as 8 u32 targets, big-endian per limb (high, low)"""
    out = []
    for t in h:
        lo, hi = split_hash_element_to_low_high(b, t)
        out += [hi, lo]
    return out


def hash_to_int_target(b, h):
    """poseidon.rs:105-117: the 128-bit scalar of a hash as four u32 limbs, least significant first (low, high of limb 0, then
    of limb 1) -- what field_hashed_scalar_mul multiplies a curve point by"""
    out = []
    for t in h[:2]:
        lo, hi = split_hash_element_to_low_high(b, t)
        out += [lo, hi]
    return out


def hash_maybe_swap(b, inputs, do_swap):
    """mp2-common/src/poseidon.rs:134-172: H(inputs[0] || inputs[1]) or, when do_swap is set, H(inputs[1] || inputs[0]) -- one
    Poseidon2 gate whose swap wire does the exchange (the node hash of every Merkle path the reference's tree circuits walk).
    inputs: two lists of 4 targets; do_swap: a boolean target. Returns 4 targets."""
    z = b.zero()
    out = b.permute_swapped(list(inputs[0]) + list(inputs[1]) + [z] * 4, do_swap)
    return out[:4]


class TestingRecursiveCircuits:
    """framework_testing.rs:71-245: the framework plus a dummy circuit that exposes NUM_PUBLIC_INPUTS unconstrained public inputs,
    so that a circuit with universal verifiers can be tested (or timed) on input proofs with chosen public inputs instead of
    proofs of the real circuits below it."""
    __test__ = False  # not a pytest class
    DUMMY = "dummy circuit"

    def __init__(self, circuits, prover, fri_params):
        n = circuits[0].num_public_inputs

        def dummy_logic(b, child_pis, inputs):  # DummyCircuitWires::circuit_logic: add_virtual_public_input_arr
            return [b.add_virtual(int(x)) for x in (inputs if inputs is not None else [0] * n)]

        self.fw = RecursiveCircuits(list(circuits) + [FrameworkCircuit(self.DUMMY, 0, dummy_logic, n)], prover, fri_params)

    def generate_input_proofs(self, public_inputs):
        """one dummy proof per row of public_inputs: verifiable by every circuit of the set"""
        return self.fw.generate_proofs_batch(self.DUMMY, [([], [], pis) for pis in public_inputs])

    def generate_proof(self, name, input_proofs, custom_inputs):
        return self.fw.generate_proofs_batch(name, [(list(input_proofs), [self.DUMMY] * len(input_proofs), custom_inputs)])[0]

    def generate_proof_from_public_inputs(self, name, public_inputs, custom_inputs):
        return self.generate_proof(name, self.generate_input_proofs(public_inputs), custom_inputs)


class RecursiveCircuitsVerifierGadget:
    """RecursiveCircuitsVerifierGagdet (framework.rs:186-262): what a circuit OUTSIDE a set of recursive circuits uses to verify a
    proof generated with that set's framework `fw`. The set's digest enters as constants (CircuitSetTarget::from_circuit_set_digest)."""

    def __init__(self, fw):
        self.fw = fw

    def dummy_inputs(self):
        """(proof, verifier data, membership) of the right shapes with zero values: what the structure pass of a circuit feeds the gadget"""
        fw = self.fw
        height = max(0, (fw.set_size - 1).bit_length()) - CIRCUIT_SET_CAP_HEIGHT
        vd = (np.zeros((1 << fw.rec.fp.cap_height, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64))
        return fw._dummy_proof(fw.rec), vd, ([0] * height, [np.zeros(4, dtype=np.uint64)] * height)

    def verify_proof_in_circuit_set(self, b, proof, vd, membership):
        """any circuit of the set: verifier data as witnesses, digest check, membership in the set, same set in the proof
        (verify_proof_in_circuit_set, framework.rs:219-234). Returns the verified proof's public-input targets."""
        set_t = [b.constant(int(x)) for x in self.fw.set_digest]
        return universal_verifier_circuit(b, self.fw.rec, set_t, self.fw.set_size, proof, vd, membership)

    def verify_proof_fixed_circuit_in_circuit_set(self, b, proof, fixed_vd):
        """one fixed circuit of the set: its verifier data are constants of the verifying circuit; the proof must expose the set's
        digest (verify_proof_fixed_circuit_in_circuit_set + check_circuit_set_equality, framework.rs:238-261)"""
        set_t = [b.constant(int(x)) for x in self.fw.set_digest]
        cap_t = [[b.constant(int(x)) for x in h] for h in np.asarray(fixed_vd[0]).reshape(-1, 4)]
        digest_t = [b.constant(int(x)) for x in fixed_vd[1]]
        pis = verify_proof_circuit(b, self.fw.rec, *proof, verifier_data=(cap_t, digest_t))
        for x, y in zip(set_t, pis[len(pis) - 4:]):
            b.connect(x, y)
        return pis


class FinalWrapCircuit:
    """verifiable-db/src/api.rs:148-214 WrapCircuitParams: the one circuit whose proofs leave the framework (towards the Groth16
    wrapper). Built over PoseidonGoldilocksConfig (`type WrapC`): it verifies a proof of ANY circuit of the set `fw` with the
    universal verifier gadget (verify_proof_in_circuit_set: Poseidon2 hashing in-circuit, since the inner proof is the default
    config's) and re-exposes the verified proof's own public inputs; the circuit itself is committed, challenged and digested with
    the ORIGINAL Poseidon, and its public inputs are hashed by PoseidonGate rows (C::InnerHasher of WrapC). `prover` must be a
    Poseidon-variant back end (framework.GpuProver(ctx, variant=POSEIDON)); fri_params(ckt) the FRI shape under that hasher."""

    def __init__(self, fw, prover, fri_params, num_public_inputs=None):
        self.fw, self.prover, self.fri_params = fw, prover, fri_params
        self.gadget = RecursiveCircuitsVerifierGadget(fw)
        self.n_pi = num_public_inputs if num_public_inputs is not None else fw.rec.n_public_inputs - 4
        self.ckt = self._build(self.gadget.dummy_inputs(), strict=False)
        self.cap, self.digest = prover.verifier_data(self.ckt)

    def _build(self, inputs, strict=True):
        b = Builder(strict, hasher=1)
        pis = self.gadget.verify_proof_in_circuit_set(b, *inputs)
        b.register_public_inputs(pis[:self.n_pi])  # get_public_input_targets::<F, N>: the verified proof's own public inputs
        return b.build(min_log_n=RECURSION_THRESHOLD)

    def program(self):
        from . import WitnessProgram
        if not hasattr(self, "_prog"):
            self._prog = WitnessProgram(self.ckt)
        return self._prog

    def generate_proof(self, proof, name):
        """WrapCircuitParams::generate_proof by the eager builder: (caps, openings, fri, public_inputs) of the wrap proof"""
        vd = self.fw.vds[name]
        w = self._build((proof, vd, self.fw.membership(vd[1])))
        assert np.array_equal(w.pre, self.ckt.pre)
        caps, openings, fri = self.prover.prove(w)
        return caps, openings, fri, w.public_inputs

    def generate_proofs_batch(self, proofs, names, capture=None):
        """the same for a batch through the recorded witness program (device replay with a GPU prover)"""
        rows = []
        for pr, name in zip(proofs, names):
            vd = self.fw.vds[name]
            rows.append(universal_inputs(pr, vd, self.fw.membership(vd[1])))
        cur = np.stack(rows)
        prog = self.program()
        if getattr(self.prover, "device_witness", False):
            return self.prover.prove_chain([self.ckt], [prog], cur, capture=capture, name="final wrap")
        wires, pi_hash, pis = prog.run(cur)
        outs = self.prover.prove_batch(self.ckt, wires, pi_hash)
        if capture is not None:
            for i, (c, o, p) in enumerate(outs):
                capture.append(("final wrap", 0, self.ckt, self.digest, wires[i].copy(), pi_hash[i].copy(), c, o, p))
        return [(c, o, p, pis[i]) for i, (c, o, p) in enumerate(outs)]


def map_logic(b, child_pis, inputs):
    """MapCircuitWires::circuit_logic (integration.rs:75-93)"""
    ins = [b.add_virtual(int(x)) for x in (inputs if inputs is not None else [0, 0, 0, 0])]
    one = b.one()
    acc = b.zero()
    for t in ins:
        is_odd = b.split_le(t, 64)[0]
        acc = b.mul_add(b.sub(one, is_odd), t, acc)
    return [acc] + b.hash_n_to_m_no_pad(ins, 4)


def reduce_logic(b, child_pis, inputs):
    """ReduceCircuitWires::circuit_logic (integration.rs:108-127)"""
    acc = b.zero()
    for pis in child_pis:
        acc = b.add(acc, pis[0])
    return [acc] + b.hash_n_to_m_no_pad([t for pis in child_pis for t in pis[1:5]], 4)


# ---- batched generate_proof: the recorded witness programs instead of the Python builder --------------------------------------------
def proof_inputs(proof):
    """the input vector of verify_proof_circuit's virtual targets for one proof (caps, openings, fri, public_inputs), in
    add_virtual order: public inputs, the three proof caps, openings, the FRI proof words"""
    caps, openings, fri, pis = proof
    return np.concatenate([np.asarray(pis, dtype=np.uint64).ravel(), np.asarray(caps, dtype=np.uint64)[1:4].ravel(),
                           np.asarray(openings, dtype=np.uint64).ravel(), np.asarray(fri, dtype=np.uint64).ravel()])


def universal_inputs(proof, vd, membership):
    """... of universal_verifier_circuit: verifier data (cap, digest), the proof, the membership proof (bits, siblings)"""
    bits, sib = membership
    return np.concatenate([np.asarray(vd[0], dtype=np.uint64).ravel(), np.asarray(vd[1], dtype=np.uint64).ravel(), proof_inputs(proof),
                           np.asarray(bits, dtype=np.uint64).ravel(), np.asarray(sib, dtype=np.uint64).ravel()])


class DeviceProof:
    """a final framework proof that stays in device memory: the address and length (u64 words) of its public inputs, its three proof
    caps, its openings and its FRI proof words, in recursion.proof_inputs order. A parent's generate_proofs_batch copies them into
    its witness inputs on the device (GpuProver.prove_chain); between ranks they travel as device tensors (sharding.send_device_proof).
    `keep` holds whatever owns the memory."""

    def __init__(self, parts, keep=None):
        self.parts, self.keep = [(int(p), int(n)) for p, n in parts], keep
        self.n_words = sum(n for _, n in self.parts)

    def to_host(self, ctx):
        """(caps [4][cap words], openings [n][2], fri words, public inputs): the host form of the same proof"""
        import ctypes
        from . import _ck, load
        out = []
        for ptr, n in self.parts:
            a = np.empty(n, dtype=np.uint64)
            _ck(load().mp2g_d2h(ctx.h, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(n * 8)))
            out.append(a)
        pis, caps3, openings, fri = out
        caps = np.concatenate([np.zeros(caps3.size // 3, dtype=np.uint64), caps3]).reshape(4, -1)  # oracle 0's cap is verifier data, not proof
        return caps, openings.reshape(-1, 2), fri, pis


def _generate_proofs_batch(self, name, jobs, threads=0, session=None, capture=None):
    """RecursiveCircuits.generate_proof for a batch of nodes of circuit `name`: jobs = [(child_proofs, child_names, inputs)].
    Witnesses come from the circuits' recorded programs (csrc/witness.hip: host threads, one proof each), proving from
    prover.prove_batch(circuit, wires [B][135][n], pi_hash [B][4]). Returns the final proofs, one per job.
    `session` (ProofSession): the prover and the host wire matrices to use instead of the framework's own -- independent trees
    (the reference's independent rows / blocks) run in one thread each with a session of their own, so that one tree's witness
    generation fills the time another's prove() spends on the GPU; circuits, verifier data and witness programs are shared.
    `capture` (a list): receives, per job and chain step, (name, step, circuit, circuit digest, wire matrix [135][n], pi_hash, caps,
    openings, proof) -- what a checker needs to prove the same witness again."""
    sess = session if session is not None else self.default_session()
    progs = self.witness_programs(name)
    on_device = bool(getattr(sess.prover, "device_witness", False))
    rows, patches = [], []
    for j, (child_proofs, child_names, inputs) in enumerate(jobs):
        parts = [np.asarray(self.set_digest, dtype=np.uint64)]
        for pr, cn in zip(child_proofs, child_names):
            vd = self.vds[cn]
            if isinstance(pr, DeviceProof) and not on_device:
                pr = pr.to_host(sess.prover.ctx)
            if isinstance(pr, DeviceProof):  # the proof words are copied in on the device; the host fills what surrounds them
                bits, sib = self.membership(vd[1])
                head = np.concatenate([np.asarray(vd[0], dtype=np.uint64).ravel(), np.asarray(vd[1], dtype=np.uint64).ravel()])
                patches.append((j, sum(p.size for p in parts) + head.size, pr))
                parts += [head, np.zeros(pr.n_words, dtype=np.uint64), np.asarray(bits, dtype=np.uint64).ravel(), np.asarray(sib, dtype=np.uint64).ravel()]
            else:
                parts.append(universal_inputs(pr, vd, self.membership(vd[1])))
        if inputs is not None:
            parts.append(np.asarray(inputs, dtype=np.uint64).ravel())
        rows.append(np.concatenate(parts))
    cur = np.stack(rows)
    if getattr(sess.prover, "device_witness", False):
        assert cur.shape[1] == progs[0].n_inputs, f"{name}: {cur.shape[1]} inputs for a program of {progs[0].n_inputs}"
        return sess.prover.prove_chain([c[0] for c in self.chains[name]], progs, cur, capture=capture, name=name, patches=patches)
    proofs = None
    for step, prog in enumerate(progs):
        assert cur.shape[1] == prog.n_inputs, f"{name} step {step}: {cur.shape[1]} inputs for a program of {prog.n_inputs}"
        rows = bool(getattr(sess.prover, "rows_layout", False))
        wires, pi_hash, pis = prog.run(cur, threads, out=sess.wire_buffer(name, step, cur.shape[0], prog.log_n, rows), rows=rows)
        outs = sess.prover.prove_batch(self.chains[name][step][0], wires, pi_hash)
        proofs = [(c, o, p, pis[i]) for i, (c, o, p) in enumerate(outs)]
        if capture is not None:
            ckt, _, digest = self.chains[name][step]
            for i, (c, o, p) in enumerate(outs):
                w = np.ascontiguousarray(wires[i].T) if rows else wires[i].copy()
                capture.append((name, step, ckt, digest, w, pi_hash[i].copy(), c, o, p))
        if step + 1 < len(progs):
            cur = np.stack([proof_inputs(p) for p in proofs])
    return proofs


class ProofSession:
    """what one thread of generate_proofs_batch owns: its prover (a GPU context / stream of its own) and the host wire
    matrices it fills, one per (circuit, batch size), reused from call to call. With a GPU prover the matrices live in pinned
    host memory and go up asynchronously on the prover's stream (prover.pinned_wires)."""

    def __init__(self, prover):
        self.prover, self.buffers = prover, {}

    def wire_buffer(self, name, step, batch, log_n, rows=False):
        cap = max(batch, getattr(self.prover, "capacity", 0) or 0)  # a prover with a capacity serves every narrower batch from one matrix
        key = (name, step, cap)
        buf = self.buffers.get(key)
        if buf is None:
            shape = (cap, 1 << log_n, 135) if rows else (cap, 135, 1 << log_n)
            make = getattr(self.prover, "pinned_wires", None)
            buf = self.buffers[key] = make(shape) if make is not None else np.empty(shape, dtype=np.uint64)
        return buf[:batch]


def _default_session(self):
    if not hasattr(self, "_session"):
        self._session = ProofSession(self.prover)
    return self._session


def _witness_programs(self, name):
    """the recorded witness programs of a circuit's chain (base, wraps); read-only once made, shared by every session"""
    from . import WitnessProgram
    if not hasattr(self, "programs"):
        self.programs = {}
    progs = self.programs.get(name)
    if progs is None:
        progs = self.programs[name] = [WitnessProgram(c[0]) for c in self.chains[name]]
    return progs


RecursiveCircuits.default_session = _default_session
RecursiveCircuits.to_bytes = _params_to_bytes
RecursiveCircuits.from_bytes = classmethod(_params_from_bytes)
RecursiveCircuits.witness_programs = _witness_programs
RecursiveCircuits.generate_proofs_batch = _generate_proofs_batch
