"""ctypes loader for the CPU oracle (oracle/liboracle.so) -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.environ.get("ORC_LIB", os.path.join(ORACLE_DIR, "liboracle.so"))  # ORC_LIB: bench.py's natively built copy
P = 0xFFFFFFFF00000001
MULT_GEN = 14293326489335486720
_u64p = ctypes.POINTER(ctypes.c_uint64)
_lib = None


def build():
    """incremental `make` of oracle/liboracle.so, serialised across processes: the ranks of a multi-GPU run (and pytest-xdist workers)
    all come through here, and two `make`s writing one .so at the same time leave a file neither of them meant"""
    import fcntl
    with open(os.path.join(ORACLE_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def lib():
    global _lib
    if _lib is None:
        if "ORC_LIB" not in os.environ:
            build()  # incremental: a no-op when liboracle.so is newer than its sources (a stale .so has bitten before)
        _lib = ctypes.CDLL(LIB)
        _lib.orc_merkle_levels_len.restype = ctypes.c_size_t
        _lib.orc_fri_proof_words.restype = ctypes.c_size_t
        _lib.orc_n_openings.restype = ctypes.c_size_t
        _lib.orc_bitrev.restype = ctypes.c_size_t
    return _lib


def arr(a, dtype=np.uint64):
    return np.ascontiguousarray(a, dtype=dtype)


def p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def sz(x):
    return ctypes.c_size_t(int(x))


class SplitMix64:
    """Deterministic field-element stream: SplitMix64, reject >= p."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)


def rand_field(shape, seed):
    """SplitMix64 stream with rejection of values >= p: the workload generator's stream
    (mapreduce-plonky2_amd/circuits.py rand_field), shared so that tests and bench draw the same inputs."""
    import importlib
    return importlib.import_module("mapreduce-plonky2_amd.circuits").rand_field(shape, seed)


# ---- thin wrappers ---------------------------------------------------------------------------
def perm(state, variant=0):
    s = arr(state).copy()
    lib().orc_perm(variant, p(s))
    return s


def hash_n_to_m_no_pad(inp, m, variant=0):
    a = arr(inp)
    out = np.empty(m, dtype=np.uint64)
    lib().orc_hash_n_to_m_no_pad(variant, p(a), sz(a.size), p(out), sz(m))
    return out


def hash_no_pad_batch(inputs, out_len=4, variant=0):
    a = arr(inputs)
    count, in_len = a.shape
    out = np.empty((count, out_len), dtype=np.uint64)
    lib().orc_hash_no_pad_batch(variant, p(a), sz(in_len), sz(count), sz(out_len), p(out))
    return out


def fft(a, inverse=False, coset_shift=0):
    a = arr(a).copy()
    batch, n = (1, a.shape[0]) if a.ndim == 1 else a.shape
    log_n = int(n).bit_length() - 1
    lib().orc_fft_batch(p(a), log_n, sz(batch), int(inverse), ctypes.c_uint64(coset_shift))
    return a


def bitrev_perm(n):
    bits = int(n).bit_length() - 1
    i = np.arange(n, dtype=np.uint64)
    r = np.zeros(n, dtype=np.uint64)
    for b in range(bits):
        r |= ((i >> np.uint64(b)) & np.uint64(1)) << np.uint64(bits - 1 - b)
    return r.astype(np.int64)


def lde_leaves(coeffs, rate_bits):
    c = arr(coeffs)
    w, n = c.shape
    out = np.empty((n << rate_bits, w), dtype=np.uint64)
    lib().orc_lde_leaves(p(c), int(n).bit_length() - 1, sz(w), rate_bits, p(out))
    return out


def merkle_build(leaves, cap_h, variant=0):
    a = arr(leaves)
    L, leaf_len = a.shape
    log_l = int(L).bit_length() - 1
    levels = np.empty(lib().orc_merkle_levels_len(log_l, cap_h), dtype=np.uint64)
    lib().orc_merkle_build(variant, p(a), sz(leaf_len), log_l, cap_h, p(levels))
    return levels


def merkle_cap(levels, cap_h):
    return levels[-(4 << cap_h):].reshape(-1, 4)


def merkle_prove(levels, log_leaves, cap_h, idx):
    sib = np.empty((log_leaves - cap_h, 4), dtype=np.uint64)
    lib().orc_merkle_prove(p(levels), log_leaves, cap_h, sz(idx), p(sib))
    return sib


def merkle_verify(leaf, idx, siblings, cap, variant=0):
    leaf, siblings, cap = arr(leaf), arr(siblings), arr(cap)
    return bool(lib().orc_merkle_verify(variant, p(leaf), sz(leaf.size), sz(idx), p(siblings), siblings.shape[0], p(cap)))


class FriParams(ctypes.Structure):
    """Mirror of orc_fri_params (oracle/fri.h) and mp2g_fri_params (include/mp2g.h)."""
    _fields_ = [("variant", ctypes.c_uint32), ("log_n", ctypes.c_uint32), ("rate_bits", ctypes.c_uint32),
                ("cap_height", ctypes.c_uint32), ("pow_bits", ctypes.c_uint32), ("num_queries", ctypes.c_uint32),
                ("n_layers", ctypes.c_uint32), ("arity_bits", ctypes.c_uint32 * 8), ("n_oracles", ctypes.c_uint32),
                ("oracle_w", ctypes.c_uint32 * 8), ("zs_oracle", ctypes.c_uint32), ("zs_count", ctypes.c_uint32),
                ("num_lookup_polys", ctypes.c_uint32)]


def standard_params(log_n, oracle_w=(84, 135, 20, 16), variant=0, rate_bits=3, cap_height=4, pow_bits=16,
                    num_queries=28, zs_oracle=2, zs_count=2, arity=4, final_poly_bits=5, num_lookup_polys=0):
    """standard_recursion_config (mp2-common/src/lib.rs:45-47) FRI parameters for degree 2^log_n."""
    fp = FriParams()
    fp.variant, fp.log_n, fp.rate_bits, fp.cap_height = variant, log_n, rate_bits, cap_height
    fp.pow_bits, fp.num_queries = pow_bits, num_queries
    ab = (ctypes.c_uint32 * 8)()
    fp.n_layers = lib().orc_reduction_arity_bits(log_n, rate_bits, cap_height, arity, final_poly_bits, ab)
    fp.arity_bits = ab
    fp.n_oracles = len(oracle_w)
    for i, w in enumerate(oracle_w):
        fp.oracle_w[i] = w
    fp.zs_oracle, fp.zs_count, fp.num_lookup_polys = zs_oracle, zs_count, num_lookup_polys
    return fp


def pcs_prove(fp, values, circuit_digest, pi_hash, num_routed=0, degree=8, quotient=False, want_challenges=False):
    """values: list of [w_o][n] arrays. Returns (caps, openings, proof). num_routed > 0: the Z /
    partial-product oracle is computed from the wires and sigmas (values[2] is ignored)."""
    vals = [arr(v) for v in values]
    ptrs = (ctypes.c_void_p * len(vals))(*[v.ctypes.data for v in vals])
    capw = 4 << fp.cap_height
    caps = np.zeros((fp.n_oracles, capw), dtype=np.uint64)
    openings = np.zeros((lib().orc_n_openings(ctypes.byref(fp)), 2), dtype=np.uint64)
    proof = np.zeros(lib().orc_fri_proof_words(ctypes.byref(fp)), dtype=np.uint64)
    cd, ph = arr(circuit_digest), arr(pi_hash)
    bgao = np.zeros(8, dtype=np.uint64)
    lib().orc_pcs_prove(ctypes.byref(fp), ptrs, p(cd), p(ph), num_routed, degree, int(bool(quotient)), p(bgao), p(caps), p(openings), p(proof))
    if want_challenges:
        return caps, openings, proof, bgao
    return caps, openings, proof


def pcs_verify(fp, circuit_digest, pi_hash, caps, openings, proof):
    cd, ph = arr(circuit_digest), arr(pi_hash)
    caps, openings, proof = arr(caps), arr(openings), arr(proof)
    return lib().orc_pcs_verify(ctypes.byref(fp), p(cd), p(ph), p(caps), p(openings), p(proof))


def partial_products_and_zs(wires, sigmas, betas, gammas, degree=8):
    wires, sigmas, betas, gammas = arr(wires), arr(sigmas), arr(betas), arr(gammas)
    num_routed, n = sigmas.shape
    nc = betas.size
    out = np.zeros((nc * (num_routed // degree), n), dtype=np.uint64)
    lib().orc_partial_products_and_zs(p(wires), p(sigmas), int(n).bit_length() - 1, num_routed, degree, p(betas), p(gammas), nc, p(out))
    return out


def plonk_identity_check(fp, num_routed, degree, openings, bgao):
    """plonk/verifier.rs vanishing(zeta) == Z_H(zeta) t(zeta) for a gate-less circuit; 0 = holds."""
    o = arr(openings)
    class G2(ctypes.Structure):
        _fields_ = [("c", ctypes.c_uint64 * 2)]
    z = G2()
    z.c[0], z.c[1] = int(bgao[6]), int(bgao[7])
    b, g, a = arr(bgao[0:2]), arr(bgao[2:4]), arr(bgao[4:6])
    return lib().orc_plonk_identity_check(ctypes.byref(fp), num_routed, degree, p(o), z, p(b), p(g), p(a))


def copy_constraint_circuit(log_n, num_routed, wires_w, n_cycles, seed):
    """A satisfied copy-constraint-only circuit: identity sigma with `n_cycles` random 3-cycles of
    routed cells forced to equal values. Returns (sigma values [num_routed][n], wires [wires_w][n])."""
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    w = pow(7277203076849721926, 1 << (32 - log_n), P)
    xs = [pow(w, i, P) for i in range(n)]
    ks = [pow(MULT_GEN, j, P) for j in range(num_routed)]
    sig = np.array([[ks[j] * x % P for x in xs] for j in range(num_routed)], dtype=np.uint64)
    wires = rand_field((wires_w, n), seed)
    cells = rng.permutation(num_routed * n)[:3 * n_cycles].reshape(-1, 3)
    for a, b, c in cells:
        (ja, ia), (jb, ib), (jc, ic) = divmod(int(a), n), divmod(int(b), n), divmod(int(c), n)
        wires[jb, ib] = wires[ja, ia]
        wires[jc, ic] = wires[ja, ia]
        sa, sb, sc = sig[ja, ia], sig[jb, ib], sig[jc, ic]
        sig[ja, ia], sig[jb, ib], sig[jc, ic] = sb, sc, sa
    return sig, wires
