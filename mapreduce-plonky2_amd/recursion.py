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
    """a base-field target: its value and (once it sits in a routed wire) its home cell"""
    __slots__ = ("v", "cell")

    def __init__(self, v, cell=None):
        self.v, self.cell = v % P, cell


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


class Builder:
    """standard_recursion_config: 135 wires, 80 routed, 2 gate constants per row"""
    ARITH_OPS, ARITH_EXT_OPS = 20, 10
    RA_BITS, RA_COPIES = 4, 4
    BASE_SUM_LIMBS = 63
    REDUCING_COEFFS = 43
    REDUCING_EXT_COEFFS = 32

    def __init__(self):
        self.rows = []
        self.parent = {}
        self.open = {}      # slot key -> [row index, used slots]
        self.consts = {}    # value -> T
        self.public_inputs = []
        self.ctx_counts = {}

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

    def _out(self, row, col, v):
        self.rows[row].wires[col] = v % P
        return T(v, (row, col))

    def connect(self, a, b):
        assert a.v == b.v, f"connect: {a.v} != {b.v}"
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
        """add_virtual_target: a witness value; its home is the first wire it is used in"""
        return T(v)

    def add_virtual_ext(self, v):
        return E(T(v[0]), T(v[1]))

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
        v %= P
        t = self.consts.get(v)
        if t is None:
            row, i = self._slot(("const",), 2, lambda: self._new_row(C.CONSTANT, 2))
            self.rows[row].consts[i] = v
            t = self.consts[v] = self._out(row, i, v)
        return t

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
        return self._out(row, 4 * i + 3, c0 * m0.v * m1.v + c1 * ad.v)

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
        return E(self._out(row, 8 * i + 6, r[0]), self._out(row, 8 * i + 7, r[1]))

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

    def mul_const_ext(self, k, a):
        return self.arithmetic_ext(k, a, self.one_ext(), 0, a)

    def mul_const_add_ext(self, k, a, b):
        return self.arithmetic_ext(k, a, self.one_ext(), 1, b)

    def add_const_ext(self, a, k):
        """a + k for a base-field constant k"""
        return self.arithmetic_ext(1, a, self.one_ext(), k, self.one_ext())

    def scalar_mul_ext(self, s, a):
        """base target s times extension a"""
        return self.mul_ext(self.to_ext(s), a)

    def div_ext(self, num, den):
        """q with q * den = num (the quotient is a witness, the product is constrained)"""
        q = self.add_virtual_ext(xmul(num.v, xinv(den.v)))
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
        return [self._out(row, 12 + i, s[i]) for i in range(12)]

    def permute(self, inputs):
        return self.permute_swapped(inputs, self.zero())

    def hash_n_to_m_no_pad(self, inputs, m):
        """hashing.rs hash_n_to_m_no_pad: overwrite-mode absorb (rate 8), squeeze from the front"""
        z = self.zero()
        state = [z] * 12
        for i in range(0, len(inputs), 8):
            chunk = inputs[i:i + 8]
            state = self.permute(list(chunk) + state[len(chunk):])
        if not inputs:
            state = self.permute(state)
        out = []
        while True:
            for t in state[:8]:
                out.append(t)
                if len(out) == m:
                    return out
            state = self.permute(state)

    def hash_or_noop(self, inputs):
        if len(inputs) <= 4:
            return list(inputs) + [self.zero()] * (4 - len(inputs))
        return self.hash_n_to_m_no_pad(inputs, 4)

    # ---- BaseSumGate<2>: bits ----------------------------------------------------------------------------------------------
    def split_le_base2(self, x, num_bits):
        """one BaseSumGate<2> row: x = sum limbs_i 2^i with num_bits <= 63 boolean limbs (the gate has 63; the unused
        high limbs are zero)"""
        assert num_bits <= self.BASE_SUM_LIMBS and x.v < (1 << num_bits)
        row = self._new_row(C.BASE_SUM, self.BASE_SUM_LIMBS, 2)
        self._put(row, 0, x)
        bits = []
        for i in range(self.BASE_SUM_LIMBS):
            t = self._out(row, 1 + i, (x.v >> i) & 1)
            if i < num_bits:
                bits.append(t)
            else:
                self.assert_zero(t)
        return bits

    def split_le(self, x, num_bits):
        """split_join.rs split_le: little-endian bits through ceil(num_bits / 63) BaseSum gates, recombined with
        weights 2^(63 i)"""
        if num_bits <= self.BASE_SUM_LIMBS:
            return self.split_le_base2(x, num_bits)
        lo = self.add_virtual(x.v & ((1 << 63) - 1))
        hi = self.add_virtual(x.v >> 63)
        bits = self.split_le_base2(lo, 63) + self.split_le_base2(hi, num_bits - 63)
        self.connect(self.arithmetic(1 << 63, hi, self.one(), 1, lo), x)
        return bits

    def range_check(self, x, n_bits):
        self.split_le_base2(x, n_bits)

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
        assert index.v < vs
        return self._out(row, base + 1, values[index.v].v)

    def random_access_ext(self, index, values):
        return E(self.random_access(index, [v.a for v in values]), self.random_access(index, [v.b for v in values]))

    # ---- ReducingGate: acc <- acc alpha + coeff over base-field coefficients ---------------------------------------------------
    def reduce_base(self, alpha, coeffs, acc=None):
        """sum_i coeffs[i] alpha^(len - 1 - i) + acc alpha^len (ReducingFactorTarget::reduce_base order is handled by
        the caller: this is the gate's Horner step)"""
        acc = acc or self.zero_ext()
        n = self.REDUCING_COEFFS
        for lo in range(0, len(coeffs), n):
            chunk = coeffs[lo:lo + n]
            row = self._new_row(C.REDUCING, n)
            for k, t in enumerate((alpha.a, alpha.b, acc.a, acc.b)):
                self._put(row, 2 + k, t)
            w = self.rows[row].wires
            cur = acc.v
            start_accs = 6 + n
            for i in range(n):
                if i < len(chunk):
                    self._put(row, 6 + i, chunk[i])
                    cv = chunk[i].v
                else:
                    # unused tail of the row: plonky2 pads the coefficient list with zeros and keeps multiplying by alpha;
                    # here the tail is cut by placing the short chunk at the END of the row (leading zero coefficients)
                    raise AssertionError
                cur = xadd(xmul(cur, alpha.v), (cv, 0))
                if i == n - 1:
                    pass
                else:
                    w[start_accs + 2 * i], w[start_accs + 2 * i + 1] = cur
            acc = E(self._out(row, 0, cur[0]), self._out(row, 1, cur[1]))
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
        return E(self._out(row, w_val, ev[0]), self._out(row, w_val + 1, ev[1]))

    # ---- public inputs -------------------------------------------------------------------------------------------------------------
    def register_public_inputs(self, targets):
        self.public_inputs += list(targets)

    # ---- build ---------------------------------------------------------------------------------------------------------------------
    def build(self, min_log_n=6):
        """CircuitBuilder::build: the public-inputs hash bound to a PublicInputGate, rows padded with Noops to a
        power of two, selectors, sigma polynomials from the copy classes. Returns a circuits.Circuit."""
        pi_hash = self.hash_n_to_m_no_pad(self.public_inputs, 4)
        pi_row = self._new_row(C.PUBLIC_INPUT)
        for i, t in enumerate(pi_hash):
            self._put(pi_row, i, t)
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
        return ckt
