"""Synthetic gate-level circuits with satisfied witnesses: the host-side stand-in for what plonky2's
CircuitBuilder + witness generators hand to prove() -- constants (selectors first), sigma polynomials,
the wire matrix and the gate table (CommonCircuitData::gates + SelectorsInfo).

Used by bench.py (the workload generator), the examples and the tests. Follows [dep] plonky2
gates/selectors.rs selector_polynomials for the selector layout and gates/*.rs for the wire layouts.
Pure Python over ints mod p; the witness of a 2^13-row circuit takes a few seconds to fill. Nothing here
touches the GPU library or the CPU oracle.
"""
import ctypes
import os
import re

import numpy as np

from . import Gate, MULT_GEN, P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rand_field(shape, seed):
    """Deterministic field-element stream: SplitMix64, values >= p rejected and re-drawn from
    the continuation of the same stream."""
    n = int(np.prod(shape))

    def draw(first, count):
        idx = np.arange(first, first + count, dtype=np.uint64)
        with np.errstate(over="ignore"):
            z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    z = draw(1, n)
    nxt = n + 1
    bad = np.flatnonzero(z >= np.uint64(P))
    while bad.size:  # probability 2^-32 per element
        z[bad] = draw(nxt, bad.size)
        nxt += bad.size
        bad = bad[z[bad] >= np.uint64(P)]
    return z.reshape(shape)


(NOOP, CONSTANT, PUBLIC_INPUT, ARITHMETIC, BASE_SUM, ARITHMETIC_EXT, MUL_EXT, POSEIDON2, EXPONENTIATION, REDUCING, REDUCING_EXT,
 RANDOM_ACCESS, POSEIDON, POSEIDON_MDS, COSET_INTERPOLATION, U32_ARITHMETIC, U32_RANGE_CHECK, U32_SUBTRACTION, U32_ADD_MANY,
 COMPARISON, LOOKUP, LOOKUP_TABLE, U32_INTERLEAVE, UNINTERLEAVE_TO_B32, UNINTERLEAVE_TO_U32) = range(25)
# the lookup argument under standard_recursion_config: LookupGate::num_slots = 80 / 2, LookupTableGate::num_slots = 80 / 3,
# ceil(40 / (8 - 1)) = 6 partial Sum/LDC polynomials + RE = 7 lookup polynomials per challenge round
NUM_LU_SLOTS, NUM_LUT_SLOTS, NUM_LOOKUP_POLYS = 40, 26, 7
LOOKUP_SELECTORS = 4  # TransSre, TransLdc, InitSre, LastLdc; then one "ends" selector per table
UNUSED_SELECTOR = 0xFFFFFFFF
NUM_WIRES, NUM_ROUTED, MAX_DEGREE = 135, 80, 8


def gate_degree(g):
    """Gate::degree()"""
    return {NOOP: 0, CONSTANT: 1, PUBLIC_INPUT: 1, ARITHMETIC: 3, BASE_SUM: g.p1, ARITHMETIC_EXT: 3, MUL_EXT: 3, POSEIDON2: 7,
            EXPONENTIATION: 4, REDUCING: 2, REDUCING_EXT: 2, RANDOM_ACCESS: g.p0 + 1, POSEIDON: 7, POSEIDON_MDS: 1, COSET_INTERPOLATION: g.p1,
            U32_ARITHMETIC: 4, U32_RANGE_CHECK: 4, U32_SUBTRACTION: 4, U32_ADD_MANY: 4,
            COMPARISON: 1 << ((g.p0 + max(g.p1, 1) - 1) // max(g.p1, 1)), LOOKUP: 0, LOOKUP_TABLE: 0, U32_INTERLEAVE: 2,
            UNINTERLEAVE_TO_B32: 2, UNINTERLEAVE_TO_U32: 2}[g.kind]


def gate_num_constraints(g):
    """Gate::num_constraints()"""
    return {NOOP: 0, CONSTANT: g.p0, PUBLIC_INPUT: 4, ARITHMETIC: g.p0, BASE_SUM: 1 + g.p0, ARITHMETIC_EXT: 2 * g.p0, MUL_EXT: 2 * g.p0,
            POSEIDON2: 123, EXPONENTIATION: g.p0 + 1, REDUCING: 2 * g.p0, REDUCING_EXT: 2 * g.p0,
            RANDOM_ACCESS: (g.p0 + 2) * g.p1 + g.p2, POSEIDON: 123, POSEIDON_MDS: 24,
            COSET_INTERPOLATION: 4 + 4 * (((1 << g.p0) - 2) // max(g.p1 - 1, 1)),
            U32_ARITHMETIC: 36 * g.p0, U32_RANGE_CHECK: 17 * g.p0, U32_SUBTRACTION: 19 * g.p0, U32_ADD_MANY: 21 * g.p1,
            COMPARISON: 6 + 5 * g.p1 + (g.p0 + max(g.p1, 1) - 1) // max(g.p1, 1), LOOKUP: 0, LOOKUP_TABLE: 0,
            U32_INTERLEAVE: 34 * g.p0, UNINTERLEAVE_TO_B32: 67 * g.p0, UNINTERLEAVE_TO_U32: 67 * g.p0}[g.kind]


_consts = None


def poseidon2_constants():
    global _consts
    if _consts is None:
        # the generated constant tables (tools/gen_constants.py writes the same numbers for product and oracle)
        src = open(os.path.join(ROOT, "mapreduce-plonky2_amd", "csrc", "perm_constants.h")).read()
        out = {}
        for name in ("POSEIDON2_RC_EXT", "POSEIDON2_RC_INT", "POSEIDON2_DIAG_M1", "POSEIDON_RC", "POSEIDON_MDS_CIRC", "POSEIDON_MDS_DIAG"):
            body = re.search(name + r"\[\d+\] = \{(.*?)\};", src, re.S).group(1)
            out[name] = [int(x.rstrip("ULu"), 0) for x in re.findall(r"0x[0-9a-fA-F]+U?L*|\d+U?L*", body)]
        _consts = out
    return _consts


M4 = ((5, 7, 1, 3), (4, 6, 1, 1), (1, 3, 5, 7), (1, 1, 4, 6))


def p2_external(s):
    t = [sum(s[4 * c + j] * M4[i][j] for j in range(4)) % P for c in range(3) for i in range(4)]
    sums = [(t[i] + t[4 + i] + t[8 + i]) % P for i in range(4)]
    return [(t[4 * c + i] + sums[i]) % P for c in range(3) for i in range(4)]


def p2_internal(s):
    d = poseidon2_constants()["POSEIDON2_DIAG_M1"]
    tot = sum(s) % P
    return [(s[i] * d[i] + tot) % P for i in range(12)]


def poseidon_mds(s):
    C = poseidon2_constants()
    circ, diag = C["POSEIDON_MDS_CIRC"], C["POSEIDON_MDS_DIAG"]
    return [(sum(s[(i + r) % 12] * circ[i] for i in range(12)) + s[r] * diag[r]) % P for r in range(12)]


def ext_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def ext_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def coset_interpolation_degree(subgroup_bits, max_degree=MAX_DEGREE):
    """CosetInterpolationGate::with_max_degree: the smallest degree that needs no more intermediates"""
    n_points = 1 << subgroup_bits
    n_intermediates = (n_points - 2) // (max_degree - 1)
    return (n_points - 2) // (n_intermediates + 1) + 2


def ext_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def fill_row(g, w, consts, inp, rng, pi_hash):
    """Write a satisfying assignment of gate `g` into the wire list `w` (length NUM_WIRES).
    inp(col) yields the value of a free routed input cell (random, or copied from an earlier cell)."""
    k = g.kind
    rnd = lambda: int(rng.integers(0, P, dtype=np.uint64))
    if k == CONSTANT:
        for i in range(g.p0):
            w[i] = consts[i]
    elif k == PUBLIC_INPUT:
        for i in range(4):
            w[i] = int(pi_hash[i])
    elif k == ARITHMETIC:
        for i in range(g.p0):
            m0, m1, ad = inp(4 * i), inp(4 * i + 1), inp(4 * i + 2)
            w[4 * i], w[4 * i + 1], w[4 * i + 2] = m0, m1, ad
            w[4 * i + 3] = (m0 * m1 % P * consts[0] + ad * consts[1]) % P
    elif k == BASE_SUM:
        limbs = [int(rng.integers(0, g.p1)) for _ in range(g.p0)]
        w[0] = sum(l * g.p1 ** i for i, l in enumerate(limbs)) % P
        w[1:1 + g.p0] = limbs
    elif k in (ARITHMETIC_EXT, MUL_EXT):
        per = 8 if k == ARITHMETIC_EXT else 6
        for i in range(g.p0):
            b = per * i
            for c in range(per - 2):
                w[b + c] = inp(b + c)
            prod = ext_mul((w[b], w[b + 1]), (w[b + 2], w[b + 3]))
            o = (prod[0] * consts[0] % P, prod[1] * consts[0] % P)
            if k == ARITHMETIC_EXT:
                o = ext_add(o, (w[b + 4] * consts[1] % P, w[b + 5] * consts[1] % P))
            w[b + per - 2], w[b + per - 1] = o
    elif k == POSEIDON2:
        C = poseidon2_constants()
        for i in range(12):
            w[i] = inp(i)
        swap = int(rng.integers(0, 2))
        w[24] = swap
        s = [0] * 12
        for i in range(4):
            delta = swap * (w[i + 4] - w[i]) % P
            w[25 + i] = delta
            s[i], s[i + 4] = (w[i] + delta) % P, (w[i + 4] - delta) % P
        s[8:12] = w[8:12]
        s = p2_external(s)
        for r in range(4):
            s = [(s[i] + C["POSEIDON2_RC_EXT"][12 * r + i]) % P for i in range(12)]
            if r:
                w[29 + 12 * (r - 1):29 + 12 * r] = s
            s = p2_external([pow(x, 7, P) for x in s])
        for r in range(22):
            s[0] = (s[0] + C["POSEIDON2_RC_INT"][r]) % P
            w[65 + r] = s[0]
            s[0] = pow(s[0], 7, P)
            s = p2_internal(s)
        for r in range(4):
            s = [(s[i] + C["POSEIDON2_RC_EXT"][12 * (4 + r) + i]) % P for i in range(12)]
            w[87 + 12 * r:87 + 12 * (r + 1)] = s
            s = p2_external([pow(x, 7, P) for x in s])
        w[12:24] = s
    elif k == POSEIDON:
        C = poseidon2_constants()
        for i in range(12):
            w[i] = inp(i)
        swap = int(rng.integers(0, 2))
        w[24] = swap
        s = [0] * 12
        for i in range(4):
            delta = swap * (w[i + 4] - w[i]) % P
            w[25 + i] = delta
            s[i], s[i + 4] = (w[i] + delta) % P, (w[i + 4] - delta) % P
        s[8:12] = w[8:12]
        for r in range(30):
            s = [(s[i] + C["POSEIDON_RC"][12 * r + i]) % P for i in range(12)]
            if 4 <= r < 26:
                w[65 + r - 4] = s[0]
                s[0] = pow(s[0], 7, P)
            else:
                if r:
                    base = 29 + 12 * (r - 1) if r < 4 else 87 + 12 * (r - 26)
                    w[base:base + 12] = s
                s = [pow(x, 7, P) for x in s]
            s = poseidon_mds(s)
        w[12:24] = s
    elif k == POSEIDON_MDS:
        for i in range(24):
            w[i] = inp(i)
        for c in range(2):
            o = poseidon_mds([w[2 * i + c] for i in range(12)])
            for i in range(12):
                w[24 + 2 * i + c] = o[i]
    elif k == COSET_INTERPOLATION:
        npts, deg = 1 << g.p0, g.p1
        nint = (npts - 2) // (deg - 1)
        w_pt, w_val = 1 + 2 * npts, 3 + 2 * npts
        w_int = w_val + 2
        w_sh = w_int + 4 * nint
        om = pow(7277203076849721926, 1 << (32 - g.p0), P)
        dom = [pow(om, i, P) for i in range(npts)]
        bw = []
        for i in range(npts):
            pr = 1
            for j in range(npts):
                if j != i:
                    pr = pr * (dom[i] - dom[j]) % P
            bw.append(pow(pr, P - 2, P))
        shift = inp(0)
        w[0] = shift
        for c in range(1, 1 + 2 * npts):
            w[c] = inp(c)
        sh = (inp(w_sh) if w_sh < NUM_ROUTED else rnd(), rnd())  # shifted point: not routed
        w[w_sh], w[w_sh + 1] = sh
        w[w_pt], w[w_pt + 1] = sh[0] * shift % P, sh[1] * shift % P
        ev, pr = (0, 0), (1, 0)
        start, end = 0, deg
        for c in range(nint + 1):
            for i in range(start, end):
                val = (w[1 + 2 * i] * bw[i] % P, w[2 + 2 * i] * bw[i] % P)
                term = ((sh[0] - dom[i]) % P, sh[1])
                ev, pr = ext_add(ext_mul(ev, term), ext_mul(val, pr)), ext_mul(pr, term)
            if c == nint:
                break
            w[w_int + 2 * c], w[w_int + 2 * c + 1] = ev
            w[w_int + 2 * (nint + c)], w[w_int + 2 * (nint + c) + 1] = pr
            start = 1 + (deg - 1) * (c + 1)
            end = min(start + deg - 1, npts)
        w[w_val], w[w_val + 1] = ev
    elif k == U32_ARITHMETIC:
        ops = g.p0
        for i in range(ops):
            b = 6 * i
            # u32 operands (a copied cell may hold a field element: reduce it so the row stays a valid u32 op)
            m0, m1, ad = [int(rng.integers(0, 1 << 32)) for _ in range(3)]
            if i == 0 and rng.random() < 0.5:
                m0 = m1 = ad = (1 << 32) - 1  # largest product: output_high = 2^32 - 1, output_low = 0 ... exercised below
            out = m0 * m1 + ad
            lo, hi = out & 0xFFFFFFFF, out >> 32
            w[b:b + 5] = [m0, m1, ad, lo, hi]
            diff = (0xFFFFFFFF - hi) % P
            w[b + 5] = pow(diff, P - 2, P) if diff else rnd()
            for j in range(32):
                w[6 * ops + 32 * i + j] = (out >> (2 * j)) & 3
    elif k == U32_RANGE_CHECK:
        kk = g.p0
        for i in range(kk):
            v = int(rng.integers(0, 1 << 32))
            w[i] = v
            for j in range(16):
                w[kk + 16 * i + j] = (v >> (2 * j)) & 3
    elif k == U32_SUBTRACTION:
        ops = g.p0
        for i in range(ops):
            x, y, bi = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32)), int(rng.integers(0, 2))
            r = x - y - bi
            bo = 1 if r < 0 else 0
            r += bo << 32
            w[5 * i:5 * i + 5] = [x, y, bi, r, bo]
            for j in range(16):
                w[5 * ops + 16 * i + j] = (r >> (2 * j)) & 3
    elif k == U32_ADD_MANY:
        na, ops = g.p0, g.p1
        per = na + 3
        for i in range(ops):
            adds = [int(rng.integers(0, 1 << 32)) for _ in range(na)]
            ci = int(rng.integers(0, 1 << 4))
            tot = sum(adds) + ci
            res, co = tot & 0xFFFFFFFF, tot >> 32
            w[per * i:per * i + per] = adds + [ci, res, co]
            for j in range(16):
                w[per * ops + 18 * i + j] = (res >> (2 * j)) & 3
            for j in range(2):
                w[per * ops + 18 * i + 16 + j] = (co >> (2 * j)) & 3
    elif k == COMPARISON:
        nb, nch = g.p0, g.p1
        cb = (nb + nch - 1) // nch
        cs = 1 << cb
        a, b = int(rng.integers(0, 1 << nb, dtype=np.uint64)), int(rng.integers(0, 1 << nb, dtype=np.uint64))
        if rng.random() < 0.3:
            b = a
        w[0], w[1] = a, b
        fc = [(a >> (cb * i)) & (cs - 1) for i in range(nch)]
        sc = [(b >> (cb * i)) & (cs - 1) for i in range(nch)]
        msd = 0
        for i in range(nch):
            diff = (sc[i] - fc[i]) % P
            eq = 1 if diff == 0 else 0
            w[4 + i], w[4 + nch + i] = fc[i], sc[i]
            w[4 + 2 * nch + i] = rnd() if eq else pow(diff, P - 2, P)  # equality dummy: inverse of the difference
            w[4 + 3 * nch + i] = eq
            iv = eq * msd % P
            w[4 + 4 * nch + i] = iv
            msd = (iv + (1 - eq) * diff) % P
        w[3] = msd
        val = (cs + msd) % P  # 2^chunk_bits + most significant difference, in [1, 2^(cb+1))
        for i in range(cb + 1):
            w[4 + 5 * nch + i] = (val >> i) & 1
        w[2] = (val >> cb) & 1
    elif k == U32_INTERLEAVE:
        ops = g.p0
        for i in range(ops):
            x = int(rng.integers(0, 1 << 32))
            w[2 * i] = x
            w[2 * i + 1] = sum(((x >> b) & 1) << (2 * b) for b in range(32))
            for j in range(32):  # bit wires most significant first
                w[2 * ops + 32 * i + j] = (x >> (31 - j)) & 1
    elif k in (UNINTERLEAVE_TO_B32, UNINTERLEAVE_TO_U32):
        ops = g.p0
        for i in range(ops):
            x = int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))
            x %= P  # a 64-bit pattern that is also a field element
            ev = [(x >> (2 * b)) & 1 for b in range(32)]
            od = [(x >> (2 * b + 1)) & 1 for b in range(32)]
            sh = 2 if k == UNINTERLEAVE_TO_B32 else 1
            w[3 * i] = x
            w[3 * i + 1] = sum(b << (sh * j) for j, b in enumerate(ev))
            w[3 * i + 2] = sum(b << (sh * j) for j, b in enumerate(od))
            for j in range(64):
                w[3 * ops + 64 * i + j] = (x >> (63 - j)) & 1
    elif k == EXPONENTIATION:
        nb = g.p0
        base = inp(0)
        bits = [int(rng.integers(0, 2)) for _ in range(nb)]
        w[0] = base
        w[1:1 + nb] = bits
        cur = 1
        for i in range(nb):
            prev = 1 if i == 0 else cur * cur % P
            cur = prev * (base if bits[nb - 1 - i] else 1) % P
            w[nb + 2 + i] = cur
        w[nb + 1] = cur
    elif k in (REDUCING, REDUCING_EXT):
        nc, ext = g.p0, k == REDUCING_EXT
        start_accs = 6 + (2 * nc if ext else nc)
        for c in range(2, 6):
            w[c] = inp(c)
        alpha, acc = (w[2], w[3]), (w[4], w[5])
        for i in range(nc):
            if ext:
                w[6 + 2 * i], w[7 + 2 * i] = inp(6 + 2 * i), inp(7 + 2 * i)
                coeff = (w[6 + 2 * i], w[7 + 2 * i])
            else:
                w[6 + i] = inp(6 + i)
                coeff = (w[6 + i], 0)
            acc = ext_add(ext_mul(acc, alpha), coeff)
            if i == nc - 1:
                w[0], w[1] = acc
            else:
                w[start_accs + 2 * i], w[start_accs + 2 * i + 1] = acc
    elif k == RANDOM_ACCESS:
        bits, copies, extra = g.p0, g.p1, g.p2
        vs = 1 << bits
        routed = (2 + vs) * copies + extra
        for c in range(copies):
            b = (2 + vs) * c
            idx = int(rng.integers(0, vs))
            for i in range(vs):
                w[b + 2 + i] = inp(b + 2 + i)
            w[b], w[b + 1] = idx, w[b + 2 + idx]
            for i in range(bits):
                w[routed + c * bits + i] = (idx >> i) & 1
        for i in range(extra):
            w[(2 + vs) * copies + i] = consts[i]
    for i in range(NUM_WIRES):
        if w[i] is None:
            w[i] = rnd()  # unused cells are unconstrained


def fill_lookup_row(g, w, r, lut_info, rng):
    """LookupGate rows (gates/lookup.rs: looking_inp 2i, looking_out 2i+1) and LookupTableGate rows
    (gates/lookup_table.rs: looked_inp 3i, looked_out 3i+1, multiplicity 3i+2; entry (first_lut_row - row) * slots + i)
    of the table the row belongs to. Multiplicities are counted while the LookupGate rows (which come first) are filled."""
    info = next(t for t in lut_info if t["last_lu_row"] <= r <= t["first_lut_row"])
    table = info["table"]
    if g.kind == LOOKUP:
        mult = info.setdefault("mult", [0] * len(table))
        done = (r - info["last_lu_row"]) * NUM_LU_SLOTS
        for i in range(NUM_LU_SLOTS):
            idx = int(rng.integers(0, len(table))) if done + i < info["n_lookups"] else 0  # padding = the first entry
            w[2 * i], w[2 * i + 1] = int(table[idx][0]), int(table[idx][1])
            mult[idx] += 1
    else:
        mult = info["mult"]
        for i in range(NUM_LUT_SLOTS):
            e = (info["first_lut_row"] - r) * NUM_LUT_SLOTS + i
            w[3 * i], w[3 * i + 1], w[3 * i + 2] = (int(table[e][0]), int(table[e][1]), mult[e]) if e < len(table) else (0, 0, 0)
    for i in range(NUM_WIRES):
        if w[i] is None:
            w[i] = int(rng.integers(0, P, dtype=np.uint64))


def selector_polynomials(gates, instances, max_degree=MAX_DEGREE + 1):
    """gates/selectors.rs selector_polynomials: `gates` sorted by degree; instances[row] = gate index.
    CircuitBuilder::build calls it with max_degree = quotient_degree_factor + 1 (a filtered constraint may reach
    degree 9: the quotient of a degree-9n vanishing polynomial by Z_H still fits the 8n coset).
    Returns (selector columns, selector_indices, groups)."""
    n, num_gates = len(instances), len(gates)
    degs = [gate_degree(g) for g in gates]
    if max(degs) + num_gates - 1 <= max_degree:
        return [[i for i in instances]], [0] * num_gates, [(0, num_gates)]
    groups, start = [], 0
    while start < num_gates:
        size = 0
        while start + size < num_gates and size + degs[start + size] < max_degree:
            size += 1
        groups.append((start, start + size))
        start += size
    sel_idx = [next(j for j, (a, b) in enumerate(groups) if a <= i < b) for i in range(num_gates)]
    cols = [[(g if a <= g < b else UNUSED_SELECTOR) for g in instances] for (a, b) in groups]
    return cols, sel_idx, groups


class Circuit:
    """gates: list of Gate with selector fields filled; pre = constants ‖ sigmas [num_constants + 80][n];
    wires [135][n]; num_selectors; pi_hash."""


def build(log_n, kinds, seed, copy_prob=0.35, luts=None):
    """A random satisfied circuit using every gate kind in `kinds` (list of (kind, p0, p1, p2)), rows dealt
    round-robin (row 0 = PublicInput when present), with random copy constraints between routed cells.
    luts: list of (table, n_lookups) with table = [(input, output)] of u16 pairs: the circuit then ends with the
    rows CircuitBuilder::add_all_lookups appends per table -- LookupGate rows holding n_lookups random lookups (the
    last row padded with the table's first entry), the LookupTableGate rows (the table runs downwards from the last
    of them, multiplicities filled as prove()'s set_lookup_wires does) and one Noop row -- and the constants gain the
    4 + len(luts) lookup selectors (gates/selectors.rs selectors_lookup / selector_ends_lookups)."""
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    gates = [Gate(k, p0, p1, p2, 0, 0, 0) for (k, p0, p1, p2) in kinds]
    gates.sort(key=lambda g: (gate_degree(g), g.kind, g.p0))  # CircuitBuilder sorts gates by (degree, id)
    order = [i for i, g in enumerate(gates) if g.kind not in (LOOKUP, LOOKUP_TABLE)]
    # rows of the lookup argument, at the end of the circuit
    lookup_rows, n_tail = [], 0
    for table, n_lookups in (luts or []):
        n_lu = max(1, -(-n_lookups // NUM_LU_SLOTS))
        n_lut = -(-len(table) // NUM_LUT_SLOTS)
        lookup_rows.append([n_lu, n_lut])
        n_tail += n_lu + n_lut + 1
    assert n_tail < n - 1, "the lookup rows do not fit"
    instances = [order[i % len(order)] for i in range(n)]
    rng.shuffle(instances)
    pi_rows = [i for i, g in enumerate(gates) if g.kind == PUBLIC_INPUT]
    if pi_rows:
        instances[0] = pi_rows[0]
        instances = [instances[0]] + [g if g != pi_rows[0] else order[(r + 1) % len(order)] if gates[order[(r + 1) % len(order)]].kind != PUBLIC_INPUT else order[0]
                                      for r, g in enumerate(instances[1:], 1)]
    lut_info = []
    if luts:
        gi = {g.kind: i for i, g in enumerate(gates)}
        row = n - n_tail
        for (table, n_lookups), (n_lu, n_lut) in zip(luts, lookup_rows):
            info = {"table": np.array(table, dtype=np.uint16).reshape(-1, 2), "n_lookups": n_lookups, "last_lu_row": row,
                    "last_lut_row": row + n_lu, "first_lut_row": row + n_lu + n_lut - 1}
            for r in range(row, row + n_lu):
                instances[r] = gi[LOOKUP]
            for r in range(row + n_lu, row + n_lu + n_lut):
                instances[r] = gi[LOOKUP_TABLE]
            instances[row + n_lu + n_lut] = gi[NOOP]
            row += n_lu + n_lut + 1
            lut_info.append(info)
    cols, sel_idx, groups = selector_polynomials(gates, instances)
    for i, g in enumerate(gates):
        g.selector_index, (g.group_start, g.group_end) = sel_idx[i], groups[sel_idx[i]]
    num_selectors = len(cols)
    pi_hash = rand_field(4, seed + 1)
    gate_consts = [[int(x) for x in rand_field(2, seed * 1000 + r)] for r in range(n)]
    wires = [[None] * NUM_WIRES for _ in range(n)]
    parent = {}

    def find(c):
        while parent.get(c, c) != c:
            c = parent[c]
        return c

    filled = []  # routed cells (row, col) with a value
    for r in range(n):
        g = gates[instances[r]]
        w = wires[r]

        def inp(col, r=r, w=w):
            if col < NUM_ROUTED and filled and rng.random() < copy_prob:
                rr, cc = filled[int(rng.integers(0, len(filled)))]
                parent[find((r, col))] = find((rr, cc))
                return wires[rr][cc]
            return int(rng.integers(0, P, dtype=np.uint64))

        if g.kind in (LOOKUP, LOOKUP_TABLE):
            fill_lookup_row(g, w, r, lut_info, rng)
        else:
            fill_row(g, w, gate_consts[r], inp, rng, pi_hash)
        if g.kind != PUBLIC_INPUT:  # nothing is copied from the public-input row: proofs of one circuit may differ there
            filled += [(r, c) for c in range(0, NUM_ROUTED, 7)]
    # sigma: identity with each equivalence class of copy-constrained cells rotated by one
    wN = pow(7277203076849721926, 1 << (32 - log_n), P)
    xs = [pow(wN, i, P) for i in range(n)]
    ks = [pow(MULT_GEN, j, P) for j in range(NUM_ROUTED)]
    sig = [[ks[j] * xs[i] % P for i in range(n)] for j in range(NUM_ROUTED)]
    classes = {}
    for c in list(parent):
        classes.setdefault(find(c), set()).add(c)
    for root, members in classes.items():
        cells = sorted(members | {root})
        vals = {wires[r][c] for r, c in cells}
        assert len(vals) == 1
        ids = [ks[c] * xs[r] % P for r, c in cells]
        for (r, c), v in zip(cells, ids[1:] + ids[:1]):
            sig[c][r] = v
    lookup_sel = []
    if lut_info:
        lookup_sel = [[0] * n for _ in range(LOOKUP_SELECTORS + len(lut_info))]
        for t, info in enumerate(lut_info):
            for r in range(info["last_lut_row"], info["first_lut_row"] + 1):
                lookup_sel[0][r] = 1  # TransSre
            for r in range(info["last_lu_row"], info["last_lut_row"]):
                lookup_sel[1][r] = 1  # TransLdc
            lookup_sel[2][info["first_lut_row"] + 1] = 1  # InitSre
            lookup_sel[3][info["last_lu_row"]] = 1        # LastLdc
            lookup_sel[LOOKUP_SELECTORS + t][info["last_lut_row"]] = 1  # end of table t: RE must equal the table's polynomial
    consts = np.array(cols + lookup_sel + [[gate_consts[r][k] for r in range(n)] for k in range(2)], dtype=np.uint64)
    ckt = Circuit()
    ckt.luts, ckt.num_lookup_selectors = lut_info, len(lookup_sel)
    ckt.num_lookup_polys = NUM_LOOKUP_POLYS if lut_info else 0
    ckt.log_n, ckt.gates, ckt.num_selectors, ckt.pi_hash = log_n, gates, num_selectors, pi_hash
    ckt.pre = np.concatenate([consts, np.array(sig, dtype=np.uint64)])
    ckt.wires = np.array(wires, dtype=np.uint64).T.copy()
    ckt.num_constants = consts.shape[0]
    ckt.instances = instances
    ckt.pi_row = 0 if pi_rows else None  # the PublicInputGate row (mp2g_prover_bind_public_inputs)
    ckt.gate_array = (Gate * len(gates))(*gates)
    return ckt


ALL_KINDS = [(NOOP, 0, 0, 0), (CONSTANT, 2, 0, 0), (PUBLIC_INPUT, 0, 0, 0), (ARITHMETIC, 20, 0, 0), (BASE_SUM, 63, 2, 0),
             (BASE_SUM, 20, 4, 0), (ARITHMETIC_EXT, 10, 0, 0), (MUL_EXT, 13, 0, 0), (POSEIDON2, 0, 0, 0),
             (EXPONENTIATION, 66, 0, 0), (REDUCING, 43, 0, 0), (REDUCING_EXT, 32, 0, 0), (RANDOM_ACCESS, 4, 4, 2),
             (POSEIDON, 0, 0, 0), (POSEIDON_MDS, 0, 0, 0), (COSET_INTERPOLATION, 4, coset_interpolation_degree(4), 0),
             (U32_ARITHMETIC, 3, 0, 0), (U32_RANGE_CHECK, 7, 0, 0), (U32_SUBTRACTION, 6, 0, 0), (U32_ADD_MANY, 3, 5, 0),
             (COMPARISON, 32, 16, 0), (U32_INTERLEAVE, 3, 0, 0), (UNINTERLEAVE_TO_B32, 2, 0, 0), (UNINTERLEAVE_TO_U32, 2, 0, 0)]
# the lookup gates (mp2-common/src/serialization/circuit_data_serialization.rs:246-247): no constraints of their own, they
# come with lookup tables and the lookup argument of prove() (build(..., luts=...)); with ALL_KINDS: all 26 registered gates
LOOKUP_KINDS = [(LOOKUP, NUM_LU_SLOTS, 0, 0), (LOOKUP_TABLE, NUM_LUT_SLOTS, 0, 0)]
# an extraction-leaf-like gate set (BASELINE configs[0]): the leaf set + Keccak's interleave gates + the lookup gates
EXTRACTION_KINDS = [(U32_INTERLEAVE, 3, 0, 0), (UNINTERLEAVE_TO_B32, 2, 0, 0), (UNINTERLEAVE_TO_U32, 2, 0, 0)] + LOOKUP_KINDS


def bits_lookup_tables():
    """two of the tables mp2-v1/src/values_extraction/gadgets/column_gadget.rs:53-68 registers: the first 5 bits of a
    byte-sized value (inputs 0..=263) and the last 3 bits (inputs 0..=256), as big-endian integers"""
    first5 = [(v, (v & 0xFF) >> 3) for v in range(256 + 8)]
    last3 = [(v, v & 7) for v in range(256 + 1)]
    return [first5, last3]


# Gate sets as the reference composes its circuits. A wrap circuit (recursion-framework/src/universal_verifier_gadget/
# wrap_circuit.rs) is plonky2's recursive verifier and nothing else; a leaf circuit of the table build (cells / rows tree,
# values extraction) adds the user logic: u32 / u256 arithmetic and comparisons next to the verifier gadget. Poseidon
# (original) and PoseidonMds only occur under the Poseidon config of the final wrap (verifiable-db/src/api.rs:148).
VERIFIER_KINDS = [(NOOP, 0, 0, 0), (CONSTANT, 2, 0, 0), (PUBLIC_INPUT, 0, 0, 0), (ARITHMETIC, 20, 0, 0), (BASE_SUM, 63, 2, 0),
                  (ARITHMETIC_EXT, 10, 0, 0), (MUL_EXT, 13, 0, 0), (POSEIDON2, 0, 0, 0), (EXPONENTIATION, 66, 0, 0),
                  (REDUCING, 43, 0, 0), (REDUCING_EXT, 32, 0, 0), (RANDOM_ACCESS, 4, 4, 2),
                  (COSET_INTERPOLATION, 4, coset_interpolation_degree(4), 0)]
LEAF_KINDS = VERIFIER_KINDS + [(BASE_SUM, 20, 4, 0), (U32_ARITHMETIC, 3, 0, 0), (U32_RANGE_CHECK, 7, 0, 0), (U32_SUBTRACTION, 6, 0, 0),
                               (U32_ADD_MANY, 3, 5, 0), (COMPARISON, 32, 16, 0)]
