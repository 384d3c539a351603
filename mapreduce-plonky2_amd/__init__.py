"""Host-side (Python) mirror of the prover interface of Lagrange-Labs/mapreduce-plonky2's hot
path, over the C ABI of libmp2gpu (include/mp2g.h).

The reference is Rust: the Python layer here is only the test / bench harness. The names follow
plonky2's as used behind `prove()` (recursion-framework/src/circuit_builder.rs:308):
PolynomialBatch.from_values / from_coeffs, MerkleTree.new / prove, Hasher.hash_no_pad.
There is no CPU fallback: loading fails loudly when the HIP library is missing, and
Context() fails when no GPU is visible.
"""
import ctypes
import time
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MP2G_LIB", os.path.join(_HERE, "libmp2gpu.so"))
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mp2g.h")

P = 0xFFFFFFFF00000001
MULT_GEN = 14293326489335486720
POSEIDON2, POSEIDON = 0, 1

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_lib = None


class Mp2gError(RuntimeError):
    pass


def load():
    """dlopen libmp2gpu.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mp2gError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`; "
                            "there is no CPU fallback for the product path")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.mp2g_last_error.restype = ctypes.c_char_p
        _lib.mp2g_ctx_stream.restype = ctypes.c_void_p
        _lib.mp2g_stat_leaf_permutations.restype = ctypes.c_uint64
        _lib.mp2g_ctx_stream.argtypes = [ctypes.c_void_p]
    return _lib


def _ck(rc):
    if rc != 0:
        raise Mp2gError(load().mp2g_last_error().decode())


def _arr(a, dtype=np.uint64):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        ptr = ctypes.c_void_p()
        _ck(load().mp2g_dev_alloc(ctx.h, ctypes.c_size_t(nbytes), ctypes.byref(ptr)))
        self.ptr = ptr
        ctx._adopt(self)

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        _ck(load().mp2g_h2d(self.ctx.h, self.ptr, _p(a), ctypes.c_size_t(a.nbytes)))
        return self

    def upload_at(self, a, offset):
        a = np.ascontiguousarray(a)
        assert offset + a.nbytes <= self.nbytes
        _ck(load().mp2g_h2d(self.ctx.h, ctypes.c_void_p(self.ptr.value + offset), _p(a), ctypes.c_size_t(a.nbytes)))
        return self

    def download(self, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _ck(load().mp2g_d2h(self.ctx.h, _p(out), self.ptr, ctypes.c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.ptr and self.ctx.h:
            load().mp2g_dev_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One per GPU (mp2g_ctx)."""

    def __init__(self, device=0):
        self.h = ctypes.c_void_p()
        self._children = weakref.WeakSet()
        _ck(load().mp2g_ctx_create(int(device), ctypes.byref(self.h)))

    def _adopt(self, obj):
        self._children.add(obj)
        return obj

    def close(self):
        """Free every handle that lives on this context, then the context itself (handles keep
        a raw pointer to their context, so they must never outlive it)."""
        if self.h:
            for obj in list(self._children):
                obj.free()
            load().mp2g_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def make_current(self):
        """select this context's device for the calling thread (HIP's current device is per thread; call at the start of a worker thread)"""
        _ck(load().mp2g_ctx_make_current(self.h))

    def mem_info(self):
        """(free, total) bytes of the context's device"""
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        _ck(load().mp2g_ctx_mem_info(self.h, ctypes.byref(free), ctypes.byref(total)))
        return int(free.value), int(total.value)

    def sync(self):
        _ck(load().mp2g_ctx_sync(self.h))

    def stream(self):
        return load().mp2g_ctx_stream(self.h)

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, a):
        a = np.ascontiguousarray(a)
        return DeviceBuffer(self, a.nbytes).upload(a)

    def host_alloc(self, nbytes):
        """Pinned host buffer as a numpy uint8 view (freed with host_free)."""
        ptr = ctypes.c_void_p()
        _ck(load().mp2g_host_alloc(self.h, ctypes.c_size_t(nbytes), ctypes.byref(ptr)))
        buf = (ctypes.c_uint8 * nbytes).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.uint8), ptr

    def host_free(self, ptr):
        _ck(load().mp2g_host_free(self.h, ptr))

    def h2d_async(self, d_dst, host_ptr, nbytes, dst_offset=0):
        _ck(load().mp2g_h2d_async(self.h, ctypes.c_void_p(d_dst.ptr.value + dst_offset), host_ptr, ctypes.c_size_t(nbytes)))

    def d2d_2d(self, d_dst, dst_offset, dst_pitch, d_src, src_offset, src_pitch, width, rows):
        """stream-ordered strided device copy (all sizes in bytes): `rows` pieces of `width` bytes"""
        _ck(load().mp2g_d2d_2d(self.h, ctypes.c_void_p(d_dst.ptr.value + dst_offset), ctypes.c_size_t(dst_pitch),
                               ctypes.c_void_p(d_src.ptr.value + src_offset), ctypes.c_size_t(src_pitch), ctypes.c_size_t(width), ctypes.c_size_t(rows)))

    def d2h_raw(self, src_ptr, shape, dtype=np.uint64):
        """download from a raw device address"""
        out = np.empty(shape, dtype=dtype)
        _ck(load().mp2g_d2h(self.h, _p(out), ctypes.c_void_p(int(src_ptr)), ctypes.c_size_t(out.nbytes)))
        return out

    def d2d_raw(self, d_dst, dst_offset, src_ptr, nbytes):
        """stream-ordered device copy of nbytes from a raw device address into a DeviceBuffer"""
        _ck(load().mp2g_d2d_2d(self.h, ctypes.c_void_p(d_dst.ptr.value + dst_offset), ctypes.c_size_t(nbytes), ctypes.c_void_p(int(src_ptr)),
                               ctypes.c_size_t(nbytes), ctypes.c_size_t(nbytes), ctypes.c_size_t(1)))

    def wires_from_rows_dev(self, d_rows, d_wires, log_n, batch, num_wires=135):
        """[batch][n][num_wires] (the witness executor's row layout) -> [batch][num_wires][n] (the prover's), on the device"""
        _ck(load().mp2g_wires_from_rows_dev(self.h, d_rows.ptr, d_wires.ptr, log_n, num_wires, batch))

    def timer_start(self):
        _ck(load().mp2g_timer_start(self.h))

    def timer_stop(self):
        ms = ctypes.c_float()
        _ck(load().mp2g_timer_stop(self.h, ctypes.byref(ms)))
        return ms.value

    # ---- plonky2_field fft.rs ---------------------------------------------------------------
    def ntt(self, data, inverse=False, coset_shift=0, bitrev_out=False):
        a = _arr(data).copy()
        batch, n = (1, a.shape[0]) if a.ndim == 1 else a.shape
        log_n = int(n).bit_length() - 1
        assert 1 << log_n == n
        _ck(load().mp2g_ntt(self.h, _p(a), log_n, batch, int(inverse), ctypes.c_uint64(coset_shift), int(bitrev_out)))
        return a

    def ntt_dev(self, d_in, d_out, log_n, batch, inverse=False, coset_shift=0, bitrev_out=False):
        _ck(load().mp2g_ntt_dev(self.h, d_in.ptr, d_out.ptr, log_n, batch, int(inverse), ctypes.c_uint64(coset_shift), int(bitrev_out)))

    def lde_leaves(self, coeffs, rate_bits):
        c = _arr(coeffs)
        w, n = c.shape
        log_n = int(n).bit_length() - 1
        out = np.empty((n << rate_bits, w), dtype=np.uint64)
        _ck(load().mp2g_lde_leaves(self.h, _p(c), log_n, w, rate_bits, _p(out)))
        return out

    def lde_dev(self, d_coeffs, log_n, w, rate_bits, d_values):
        _ck(load().mp2g_lde_dev(self.h, d_coeffs.ptr, log_n, w, rate_bits, d_values.ptr))

    # ---- Hasher -------------------------------------------------------------------------------
    def hash_no_pad_batch(self, inputs, out_len=4, variant=POSEIDON2):
        a = _arr(inputs)
        count, in_len = a.shape
        out = np.empty((count, out_len), dtype=np.uint64)
        _ck(load().mp2g_hash_no_pad_batch(self.h, variant, _p(a), in_len, count, out_len, _p(out)))
        return out

    def hash_no_pad(self, values, variant=POSEIDON2):
        return self.hash_no_pad_batch(_arr(values).reshape(1, -1), 4, variant)[0]


class MerkleTree:
    """plonky2 hash/merkle_tree.rs MerkleTree."""

    def __init__(self, ctx, leaves, cap_height, variant=POSEIDON2):
        a = _arr(leaves)
        L, leaf_len = a.shape
        self.log_leaves = int(L).bit_length() - 1
        assert 1 << self.log_leaves == L
        self.ctx, self.cap_height, self.leaf_len = ctx, cap_height, leaf_len
        self.h = ctypes.c_void_p()
        _ck(load().mp2g_merkle_build(ctx.h, variant, _p(a), leaf_len, self.log_leaves, cap_height, ctypes.byref(self.h)))
        ctx._adopt(self)

    @property
    def cap(self):
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        _ck(load().mp2g_merkle_cap(self.h, _p(out)))
        return out

    def prove(self, indices):
        idx = _arr(indices, np.uint32)
        depth = self.log_leaves - self.cap_height
        leaves = np.empty((len(idx), self.leaf_len), dtype=np.uint64)
        sib = np.empty((len(idx), depth, 4), dtype=np.uint64)
        _ck(load().mp2g_merkle_open(self.h, _p(idx), len(idx), _p(leaves), _p(sib)))
        return leaves, sib

    def free(self):
        if self.h and self.ctx.h:
            load().mp2g_merkle_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PolynomialBatch:
    """plonky2 fri/oracle.rs PolynomialBatch: coefficients + LDE Merkle tree."""

    def __init__(self, ctx, handle, log_n, w, rate_bits, cap_height):
        self.ctx, self.h = ctx, handle
        self.log_n, self.w, self.rate_bits, self.cap_height = log_n, w, rate_bits, cap_height
        ctx._adopt(self)

    @classmethod
    def from_values(cls, ctx, values, rate_bits=3, cap_height=4, variant=POSEIDON2):
        v = _arr(values)
        w, n = v.shape
        log_n = int(n).bit_length() - 1
        h = ctypes.c_void_p()
        _ck(load().mp2g_commit_from_values(ctx.h, variant, _p(v), log_n, w, rate_bits, cap_height, ctypes.byref(h)))
        return cls(ctx, h, log_n, w, rate_bits, cap_height)

    @classmethod
    def from_values_dev(cls, ctx, d_values, log_n, w, rate_bits=3, cap_height=4, variant=POSEIDON2):
        h = ctypes.c_void_p()
        _ck(load().mp2g_commit_from_values_dev(ctx.h, variant, d_values.ptr, log_n, w, rate_bits, cap_height, ctypes.byref(h)))
        return cls(ctx, h, log_n, w, rate_bits, cap_height)

    @classmethod
    def from_coeffs_dev(cls, ctx, d_coeffs, log_n, w, rate_bits=3, cap_height=4, variant=POSEIDON2):
        h = ctypes.c_void_p()
        _ck(load().mp2g_commit_from_coeffs_dev(ctx.h, variant, d_coeffs.ptr, log_n, w, rate_bits, cap_height, ctypes.byref(h)))
        return cls(ctx, h, log_n, w, rate_bits, cap_height)

    def recommit_from_values_dev(self, d_values):
        _ck(load().mp2g_recommit_from_values_dev(self.ctx.h, self.h, d_values.ptr))

    def rehash_dev(self, parts=3):
        """MerkleTree::new over the LDE values the batch holds: parts 1 = leaf sponges, 2 = tree levels, 3 = both"""
        _ck(load().mp2g_batch_rehash_dev(self.ctx.h, self.h, parts))

    @property
    def cap(self):
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        _ck(load().mp2g_batch_cap(self.h, _p(out)))
        return out

    @property
    def coeffs(self):
        out = np.empty((self.w, 1 << self.log_n), dtype=np.uint64)
        _ck(load().mp2g_batch_coeffs(self.h, _p(out)))
        return out

    def eval_ext(self, point):
        """Evaluate every polynomial of the batch at an extension-field point (the openings)."""
        pt = _arr(point)
        out = np.empty((self.w, 2), dtype=np.uint64)
        _ck(load().mp2g_batch_eval_ext(self.h, _p(pt), _p(out)))
        return out

    def open(self, indices):
        idx = _arr(indices, np.uint32)
        depth = self.log_n + self.rate_bits - self.cap_height
        leaves = np.empty((len(idx), self.w), dtype=np.uint64)
        sib = np.empty((len(idx), depth, 4), dtype=np.uint64)
        _ck(load().mp2g_batch_open(self.h, _p(idx), len(idx), _p(leaves), _p(sib)))
        return leaves, sib

    def free(self):
        if self.h and self.ctx.h:
            load().mp2g_batch_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FriParams(ctypes.Structure):
    """mp2g_fri_params (include/mp2g.h)."""
    _fields_ = [("variant", ctypes.c_uint32), ("log_n", ctypes.c_uint32), ("rate_bits", ctypes.c_uint32),
                ("cap_height", ctypes.c_uint32), ("pow_bits", ctypes.c_uint32), ("num_queries", ctypes.c_uint32),
                ("n_layers", ctypes.c_uint32), ("arity_bits", ctypes.c_uint32 * 8), ("n_oracles", ctypes.c_uint32),
                ("oracle_w", ctypes.c_uint32 * 8), ("zs_oracle", ctypes.c_uint32), ("zs_count", ctypes.c_uint32),
                ("num_lookup_polys", ctypes.c_uint32)]

    @property
    def proof_words(self):
        f = load().mp2g_fri_proof_words
        f.restype = ctypes.c_size_t
        return f(ctypes.byref(self))

    @property
    def n_openings(self):
        f = load().mp2g_fri_n_openings
        f.restype = ctypes.c_size_t
        return f(ctypes.byref(self))

    @property
    def cap_words(self):
        return 4 << self.cap_height


def standard_recursion_params(log_n, oracle_w=(84, 135, 20, 16), variant=POSEIDON2, rate_bits=3, cap_height=4,
                              pow_bits=16, num_queries=28, zs_oracle=2, zs_count=2, arity_bits=4, final_poly_bits=5,
                              num_lookup_polys=0):
    """FRI parameters of standard_recursion_config (mp2-common/src/lib.rs:45-47) for 2^log_n rows:
    135 wires, 2 challenges (=> 20 Z/partial-product and 16 quotient-chunk polynomials),
    ConstantArityBits(4, 5). 84 = constants + 80 sigma polynomials of a typical circuit."""
    fp = FriParams()
    fp.variant, fp.log_n, fp.rate_bits, fp.cap_height = variant, log_n, rate_bits, cap_height
    fp.pow_bits, fp.num_queries = pow_bits, num_queries
    ab = (ctypes.c_uint32 * 8)()
    fp.n_layers = load().mp2g_reduction_arity_bits(log_n, rate_bits, cap_height, arity_bits, final_poly_bits, ab)
    fp.arity_bits = ab
    fp.n_oracles = len(oracle_w)
    for i, w in enumerate(oracle_w):
        fp.oracle_w[i] = w
    fp.zs_oracle, fp.zs_count, fp.num_lookup_polys = zs_oracle, zs_count, num_lookup_polys
    return fp


class Challenger:
    """plonky2 iop/challenger.rs Challenger; `count` transcripts in lockstep on the device."""

    def __init__(self, ctx, variant=POSEIDON2, count=1):
        self.ctx, self.count = ctx, count
        self.h = ctypes.c_void_p()
        _ck(load().mp2g_challenger_create(ctx.h, variant, count, ctypes.byref(self.h)))
        ctx._adopt(self)

    def observe_elements(self, elems):
        a = _arr(elems).reshape(self.count, -1)
        _ck(load().mp2g_challenger_observe(self.h, _p(a), a.shape[1]))

    def get_n_challenges(self, n):
        out = np.empty((self.count, n), dtype=np.uint64)
        _ck(load().mp2g_challenger_get(self.h, n, _p(out)))
        return out

    def free(self):
        if self.h and self.ctx.h:
            load().mp2g_challenger_free(self.h)
        self.h = None


def fri_fold(ctx, evals, arity_bits, beta, shift):
    a = _arr(evals)
    m = a.shape[0]
    log_m = int(m).bit_length() - 1
    out = np.empty((m >> arity_bits, 2), dtype=np.uint64)
    b = _arr(beta)
    _ck(load().mp2g_fri_fold(ctx.h, _p(a), log_m, arity_bits, _p(b), ctypes.c_uint64(shift), _p(out)))
    return out


def fri_pow(ctx, state, pos, bits, variant=POSEIDON2):
    s = _arr(state)
    w = ctypes.c_uint64()
    _ck(load().mp2g_fri_pow(ctx.h, variant, _p(s), pos, bits, ctypes.byref(w)))
    return w.value


def pcs_prove(ctx, fp, values, circuit_digest, pi_hash):
    """Single proof, host arrays: the PCS skeleton of prove(). Returns (caps, openings, proof)."""
    vals = [_arr(v) for v in values]
    ptrs = (ctypes.c_void_p * len(vals))(*[v.ctypes.data for v in vals])
    caps = np.empty((fp.n_oracles, fp.cap_words), dtype=np.uint64)
    openings = np.empty((fp.n_openings, 2), dtype=np.uint64)
    proof = np.empty(fp.proof_words, dtype=np.uint64)
    cd, ph = _arr(circuit_digest), _arr(pi_hash)
    _ck(load().mp2g_pcs_prove(ctx.h, ctypes.byref(fp), ptrs, _p(cd), _p(ph), _p(caps), _p(openings), _p(proof)))
    return caps, openings, proof


class Gate(ctypes.Structure):
    """mp2g_gate: one entry of CommonCircuitData::gates with its selector group."""
    _fields_ = [("kind", ctypes.c_uint32), ("p0", ctypes.c_uint32), ("p1", ctypes.c_uint32), ("p2", ctypes.c_uint32),
                ("selector_index", ctypes.c_uint32), ("group_start", ctypes.c_uint32), ("group_end", ctypes.c_uint32)]

    @property
    def num_constraints(self):
        return load().mp2g_gate_num_constraints(ctypes.byref(self))

    @property
    def degree(self):
        return load().mp2g_gate_degree(ctypes.byref(self))


(GATE_NOOP, GATE_CONSTANT, GATE_PUBLIC_INPUT, GATE_ARITHMETIC, GATE_BASE_SUM, GATE_ARITHMETIC_EXT, GATE_MUL_EXT, GATE_POSEIDON2,
 GATE_EXPONENTIATION, GATE_REDUCING, GATE_REDUCING_EXT, GATE_RANDOM_ACCESS, GATE_POSEIDON, GATE_POSEIDON_MDS,
 GATE_COSET_INTERPOLATION, GATE_U32_ARITHMETIC, GATE_U32_RANGE_CHECK, GATE_U32_SUBTRACTION, GATE_U32_ADD_MANY,
 GATE_COMPARISON, GATE_LOOKUP, GATE_LOOKUP_TABLE, GATE_U32_INTERLEAVE, GATE_UNINTERLEAVE_TO_B32, GATE_UNINTERLEAVE_TO_U32) = range(25)


class Lookup(ctypes.Structure):
    """mp2g_lookup: one lookup table and its rows (plonky2's LookupWire + the table of CommonCircuitData::luts)."""
    _fields_ = [("last_lu_row", ctypes.c_uint32), ("last_lut_row", ctypes.c_uint32), ("first_lut_row", ctypes.c_uint32),
                ("table_len", ctypes.c_uint32), ("table", ctypes.c_void_p)]


def lookup_array(luts):
    """(Lookup * n) from the builder's lookup descriptions (dicts with the row fields and a uint16 [len][2] table);
    the second value keeps the table arrays alive"""
    tabs = [np.ascontiguousarray(t["table"], dtype=np.uint16) for t in luts]
    arr = (Lookup * max(1, len(luts)))()
    for i, (t, tab) in enumerate(zip(luts, tabs)):
        arr[i] = Lookup(t["last_lu_row"], t["last_lut_row"], t["first_lut_row"], tab.shape[0], tab.ctypes.data)
    return arr, tabs


def eval_gate_constraints(ctx, gates, num_selectors, consts, wires, pi_hash):
    """Filtered gate constraints C_j at arbitrary points: consts [num_constants][npts], wires [w][npts]
    -> [max_j][npts]. All zero on H for a satisfied witness (plonky2's prove() panics otherwise)."""
    c, w, ph = _arr(consts), _arr(wires), _arr(pi_hash)
    arr = (Gate * len(gates))(*gates)
    max_j = max(g.num_constraints for g in gates)
    out = np.zeros((max_j, w.shape[1]), dtype=np.uint64)
    _ck(load().mp2g_eval_gate_constraints(ctx.h, arr, len(gates), num_selectors, _p(c), c.shape[0], _p(w), w.shape[0],
                                          ctypes.c_uint64(w.shape[1]), _p(ph), _p(out)))
    return out


class BatchedProver:
    """mp2g_prover: `batch` same-shape proofs per call, device resident."""

    def __init__(self, ctx, fp, batch):
        self.ctx, self.fp, self.batch = ctx, fp, batch
        self.h = ctypes.c_void_p()
        _ck(load().mp2g_prover_create(ctx.h, ctypes.byref(fp), batch, ctypes.byref(self.h)))
        ctx._adopt(self)
        self.d_caps = ctx.alloc(batch * fp.n_oracles * fp.cap_words * 8)
        self.d_openings = ctx.alloc(batch * fp.n_openings * 2 * 8)
        self.d_proof = ctx.alloc(batch * fp.proof_words * 8)

    def set_preprocessed(self, d_values):
        _ck(load().mp2g_prover_set_preprocessed_dev(self.h, d_values.ptr))

    def enable_permutation(self, num_routed=80, degree=8):
        """Compute the Z / partial-product oracle on the device from the wires and the sigma
        polynomials (prove()'s permutation argument); d_values[1] may then be None."""
        _ck(load().mp2g_prover_enable_permutation(self.h, num_routed, degree))

    def enable_quotient(self):
        """Also compute the quotient chunks on the device: permutation terms, plus the gate constraints
        once set_gates() has given the gate table; d_values[2] may then be None."""
        _ck(load().mp2g_prover_enable_quotient(self.h))

    def set_gates(self, gates, num_selectors):
        """Gate table of the circuit (CommonCircuitData::gates + SelectorsInfo): the quotient then
        includes the gate constraint terms -- prove() of a circuit built from the supported gates."""
        arr = (Gate * len(gates))(*gates)
        _ck(load().mp2g_prover_set_gates(self.h, arr, len(gates), num_selectors))

    def set_lookups(self, luts):
        """Lookup tables of the circuit (CommonCircuitData::luts + ProverOnlyCircuitData::lookup_rows): prove() then
        draws the lookup challenges, computes the RE / Sum / LDC polynomials into the Z oracle and adds the lookup
        terms to the quotient. Needs set_gates() with the lookup gates in the table."""
        arr, keep = lookup_array(luts)
        _ck(load().mp2g_prover_set_lookups(self.h, arr, len(luts)))

    def bind_public_inputs(self, row):
        """PublicInputGate's generator on the device: wires 0..3 of `row` of every proof's wire matrix are
        overwritten (in place) with that proof's public-inputs hash before the commitment."""
        _ck(load().mp2g_prover_bind_public_inputs(self.h, ctypes.c_int64(-1 if row is None else int(row))))

    def enable_witness_check(self, on=True):
        """Check gate and copy constraints of every witness on the device (plonky2 panics on a bad one)."""
        _ck(load().mp2g_prover_enable_witness_check(self.h, int(on)))

    def set_active(self, n):
        """prove only the first n <= batch witnesses from the next prove() on (no reallocation: every buffer is proof-major)"""
        _ck(load().mp2g_prover_set_active(self.h, int(n)))
        self.active = int(n)

    def witness_status(self):
        """Per-proof flags of the last prove() (bit 0 copy constraint, bit 1 gate constraint); raises
        Mp2gError naming the first bad proof, like prove()'s panic in the reference."""
        flags = np.zeros(self.batch, dtype=np.uint32)
        rc = load().mp2g_prover_witness_status(self.h, _p(flags))
        if rc:
            err = Mp2gError(load().mp2g_last_error().decode())
            err.flags = flags
            raise err
        return flags

    def enable_graph(self, on=True):
        """Replay the launch sequence as a hipGraph from the third prove() with the same buffers on."""
        _ck(load().mp2g_prover_enable_graph(self.h, int(on)))

    STAGES = ("wires_commit", "z_partial_products", "quotient", "openings", "fri_commit_phase", "proof_of_work", "fri_queries")

    def enable_timing(self, on=True):
        _ck(load().mp2g_prover_enable_timing(self.h, int(on)))

    def stage_ms(self):
        """{stage: milliseconds} of the last prove() (HIP events on the prover's stream); synchronises."""
        out = (ctypes.c_float * len(self.STAGES))()
        _ck(load().mp2g_prover_stage_ms(self.h, out))
        return dict(zip(self.STAGES, [float(x) for x in out]))

    def prove(self, d_values, d_circuit_digest, d_pi_hash):
        """d_values: device buffers [batch][w_o][n] for oracles 1..; asynchronous."""
        ptrs = (ctypes.c_void_p * len(d_values))(*[(d.ptr.value if d is not None else None) for d in d_values])
        _ck(load().mp2g_prover_prove_dev(self.h, ptrs, d_circuit_digest.ptr, d_pi_hash.ptr, self.d_caps.ptr,
                                         self.d_openings.ptr, self.d_proof.ptr))

    def results(self):
        fp, B = self.fp, getattr(self, "active", self.batch)
        return (self.d_caps.download((B, fp.n_oracles, fp.cap_words)), self.d_openings.download((B, fp.n_openings, 2)),
                self.d_proof.download((B, fp.proof_words)))

    def free(self):
        if self.h and self.ctx.h:
            load().mp2g_prover_free(self.h)
        self.h = None


class WitnessProgram:
    """mp2g_witness_program: the witness generator of one circuit built by recursion.Builder (its recorded tape),
    replayed on the host for batches of input vectors. No GPU involved."""

    def __init__(self, ckt):
        tape, ins, cs = _arr(ckt.tape), _arr(ckt.input_sids, np.uint32), _arr(ckt.const_slots).reshape(-1, 2)
        self.log_n, self.n_inputs = ckt.log_n, int(ins.size)
        self.probe = _arr(np.concatenate([ckt.pi_hash_sids, ckt.public_input_sids]), np.uint32)
        self.n_public_inputs = int(ckt.public_input_sids.size)
        self.h = ctypes.c_void_p()
        _ck(load().mp2g_witness_program_create(_p(tape), ctypes.c_size_t(tape.size), int(ckt.n_slots), int(ckt.log_n), _p(ins), int(ins.size),
                                               _p(cs), int(cs.shape[0]), ctypes.byref(self.h)))
        _ck(load().mp2g_witness_program_set_probe(self.h, _p(self.probe), int(self.probe.size)))
        self.n_levels = int(load().mp2g_witness_program_num_levels(self.h))

    def run_dev(self, ctx, d_inputs, batch, d_wires, d_probe):
        """the same replay on the device, stream ordered on ctx's stream: d_inputs [batch][n_inputs] -> d_wires [batch][135][n] (the
        prover's layout) and d_probe [batch][4 + n_public_inputs] (public-inputs hash, then the public inputs); asynchronous"""
        _ck(load().mp2g_witness_program_run_dev(self.h, ctx.h, d_inputs.ptr, int(batch), d_wires.ptr, d_probe.ptr))

    def run(self, inputs, threads=0, out=None, rows=False):
        """inputs [batch][n_inputs] -> (wires [batch][135][n], pi_hash [batch][4], public_inputs [batch][n_pi]);
        rows=True: the wires as [batch][n][135] (one contiguous row per gate row: faster to fill; Context.wires_from_rows_dev
        gives the prover's layout on the device)"""
        a = _arr(inputs).reshape(-1, self.n_inputs)
        B = a.shape[0]
        shape = (B, 1 << self.log_n, 135) if rows else (B, 135, 1 << self.log_n)
        wires = out if out is not None else np.empty(shape, dtype=np.uint64)
        assert wires.shape == shape
        probe = np.empty((B, self.probe.size), dtype=np.uint64)
        fn = load().mp2g_witness_program_run_rows if rows else load().mp2g_witness_program_run
        _ck(fn(self.h, _p(a), B, int(threads), _p(wires), _p(self.probe), int(self.probe.size), _p(probe)))
        return wires, probe[:, :4], probe[:, 4:]

    def free(self):
        if self.h:
            load().mp2g_witness_program_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ChainPatch(ctypes.Structure):
    _fields_ = [("job", ctypes.c_uint32), ("offset", ctypes.c_uint32), ("n_words", ctypes.c_uint32), ("pad_", ctypes.c_uint32), ("d_src", ctypes.c_void_p)]


class ProofChain:
    """mp2g_chain: generate_proof (base prove() + wrap chain) for batches of nodes of one framework circuit, on the device.
    provers: BatchedProver per step (set up for its circuit), programs: WitnessProgram per step, d_digests: DeviceBuffer per step."""

    def __init__(self, ctx, provers, programs, d_digests, capacity):
        n = len(provers)
        self.ctx, self.n_steps, self.capacity = ctx, n, capacity
        self.seconds_in_run = 0.0
        self.provers, self.programs, self.d_digests = list(provers), list(programs), list(d_digests)  # keep the handles alive
        self.fps = [p.fp for p in provers]
        pr = (ctypes.c_void_p * n)(*[p.h for p in provers])
        pg = (ctypes.c_void_p * n)(*[p.h for p in programs])
        fp = (FriParams * n)(*self.fps)
        dg = (ctypes.c_void_p * n)(*[d.ptr for d in d_digests])
        self.h = ctypes.c_void_p()
        _ck(load().mp2g_chain_create(ctx.h, n, pr, pg, fp, dg, int(capacity), ctypes.byref(self.h)))
        ctx._adopt(self)

    def run(self, inputs, patches=()):
        """inputs [B][n_inputs] (host); patches [(job, word offset, device address, n words)] -> (caps [B][4][cap words], openings
        [B][n][2], FRI proofs [B][words], public inputs [B][n_pi]) of the last step; raises on an unsatisfied witness"""
        a = _arr(inputs)
        B = a.shape[0]
        fp, prog = self.fps[-1], self.programs[-1]
        caps = np.empty((B, fp.n_oracles, fp.cap_words), dtype=np.uint64)
        openings = np.empty((B, fp.n_openings, 2), dtype=np.uint64)
        proofs = np.empty((B, fp.proof_words), dtype=np.uint64)
        pis = np.empty((B, prog.n_public_inputs), dtype=np.uint64)
        arr = (ChainPatch * max(1, len(patches)))(*[ChainPatch(int(j), int(off), int(n), 0, int(ptr)) for j, off, ptr, n in patches])
        t0 = time.perf_counter()
        _ck(load().mp2g_chain_run(self.h, _p(a), B, arr, len(patches), _p(caps), _p(openings), _p(proofs), _p(pis)))
        self.seconds_in_run += time.perf_counter() - t0  # inside the library (the GIL is released): what is left of a worker's time is host glue
        self.last_batch = B
        return caps, openings, proofs, pis

    def step_buffers(self, step):
        """raw device addresses of a step's (wires, probe, caps, openings, proof) after a run"""
        ptrs = [ctypes.c_void_p() for _ in range(5)]
        _ck(load().mp2g_chain_step_buffers(self.h, int(step), *[ctypes.byref(p) for p in ptrs]))
        return [p.value for p in ptrs]

    def device_proof(self, b=0):
        """[(device address, words)] x 4 of proof b of the last run: public inputs, the three caps, openings, FRI proof"""
        parts, words = (ctypes.c_void_p * 4)(), (ctypes.c_uint32 * 4)()
        _ck(load().mp2g_chain_device_proof(self.h, int(b), parts, words))
        return [(int(parts[i]), int(words[i])) for i in range(4)]

    def free(self):
        if self.h and self.ctx.h:
            load().mp2g_chain_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ForestCircuit(ctypes.Structure):
    _fields_ = [("n_inputs", ctypes.c_uint32), ("n_children", ctypes.c_uint32), ("child_offset", ctypes.c_uint32 * 4), ("n_const", ctypes.c_uint32)]


class Forest:
    """mp2g_forest: the native scheduler of a tree build (csrc/forest.hip). ctxs: one Context per worker; descriptors: [(n_inputs,
    [child offsets], n_const)] per circuit; chains[w][c]: the ProofChain of circuit c on worker w's context (None = never used there)."""

    def __init__(self, ctxs, descriptors, chains, slot_words, pool_slots):
        self.ctxs, self.chains = list(ctxs), [list(row) for row in chains]  # keep the handles alive
        nw, nc = len(ctxs), len(descriptors)
        d = (ForestCircuit * nc)()
        for i, (n_in, offs, n_const) in enumerate(descriptors):
            d[i].n_inputs, d[i].n_children, d[i].n_const = int(n_in), len(offs), int(n_const)
            for k, o in enumerate(offs):
                d[i].child_offset[k] = int(o)
        cx = (ctypes.c_void_p * nw)(*[c.h for c in ctxs])
        ch = (ctypes.c_void_p * (nw * nc))(*[(self.chains[w][c].h if self.chains[w][c] is not None else None) for w in range(nw) for c in range(nc)])
        self.h = ctypes.c_void_p()
        _ck(load().mp2g_forest_create(nw, cx, nc, d, ch, int(slot_words), int(pool_slots), ctypes.byref(self.h)))
        for c in self.ctxs:  # the forest holds raw pointers to every worker's context and chains: whichever closes first frees it
            c._adopt(self)
        self.n_const = [int(x[2]) for x in descriptors]
        self.n_children = [len(x[1]) for x in descriptors]

    def add_nodes(self, circuit, ids, child_ids, consts, keep=None):
        ids = _arr(ids)
        n = ids.size
        kids = _arr(child_ids).reshape(n, self.n_children[circuit]) if self.n_children[circuit] else None
        cs = _arr(consts).reshape(n, self.n_const[circuit])
        kp = np.ascontiguousarray(keep, dtype=np.uint8) if keep is not None else None
        _ck(load().mp2g_forest_add_nodes(self.h, int(circuit), int(n), _p(ids), _p(kids) if kids is not None else None, _p(cs), _p(kp) if kp is not None else None))

    def prove(self, units):
        """units: lists of node ids, independent of each other; the workers take them from a queue (the call releases the GIL)"""
        flat = _arr([i for u in units for i in u])
        offs = _arr(np.cumsum([0] + [len(u) for u in units]), np.uint32)
        _ck(load().mp2g_forest_prove(self.h, _p(flat), _p(offs), len(units)))

    def prove_plan(self, plan, group_nodes, n_satellites=0, satellite_shift=40):
        """drain an UpdatePlan (workplan.py) inside the library: the Ready items of every wave grouped into units of ~group_nodes plan
        nodes, proved, marked done, until the plan is finished. A plan node k = forest node k + its satellites ((j + 1) << shift) | k.
        Returns the items of every wave."""
        waves, per = ctypes.c_uint32(), (ctypes.c_uint32 * 64)()
        _ck(load().mp2g_forest_prove_plan(self.h, plan.h, int(group_nodes), int(n_satellites), int(satellite_shift), ctypes.byref(waves), per, 64))
        return [int(per[i]) for i in range(min(64, waves.value))]

    def proof_words(self, node_id):
        n = ctypes.c_uint32()
        _ck(load().mp2g_forest_proof(self.h, ctypes.c_uint64(int(node_id)), None, ctypes.byref(n)))
        out = np.empty(n.value, dtype=np.uint64)
        _ck(load().mp2g_forest_proof(self.h, ctypes.c_uint64(int(node_id)), _p(out), ctypes.byref(n)))
        return out

    def device_proof(self, node_id):
        """(device address, words) of a proved node's proof in the pool: public inputs, caps of oracles 1..3, openings, FRI words"""
        ptr, n = ctypes.c_void_p(), ctypes.c_uint32()
        _ck(load().mp2g_forest_device_proof(self.h, ctypes.c_uint64(int(node_id)), ctypes.byref(ptr), ctypes.byref(n)))
        return int(ptr.value), int(n.value)

    def release(self, node_id):
        _ck(load().mp2g_forest_release(self.h, ctypes.c_uint64(int(node_id))))

    @property
    def proved(self):
        load().mp2g_forest_proved.restype = ctypes.c_uint64
        return int(load().mp2g_forest_proved(self.h))

    def free(self):
        if self.h and all(c.h for c in self.ctxs):
            load().mp2g_forest_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- Ecgfp5 multiset digest (mp2-common/src/group_hashing) ---------------------------------
def map_to_curve_batch(ctx, inputs, variant=POSEIDON2, weierstrass=False):
    """map_to_curve_point for each row of `inputs`; returns encodings [count][5] (and the 11-limb
    Weierstrass form when asked)."""
    a = _arr(inputs)
    count, in_len = a.shape
    w = np.empty((count, 5), dtype=np.uint64)
    wei = np.empty((count, 11), dtype=np.uint64) if weierstrass else None
    _ck(load().mp2g_map_to_curve_batch(ctx.h, variant, _p(a), in_len, count, _p(w), _p(wei) if weierstrass else None))
    return (w, wei) if weierstrass else w


def curve_sum(ctx, pts_w, weierstrass=False):
    a = _arr(pts_w).reshape(-1, 5)
    w = np.empty(5, dtype=np.uint64)
    wei = np.empty(11, dtype=np.uint64)
    _ck(load().mp2g_curve_sum(ctx.h, _p(a), a.shape[0], _p(w), _p(wei)))
    return (w, wei) if weierstrass else w


def curve_sum_ranges(ctx, pts_w, ranges):
    """the sum of pts_w[start:end] for every (start, end) of `ranges` in one call: the accumulated digest of every node of a
    tree laid out in order. Returns (encodings [n][5], Weierstrass forms [n][11])."""
    a = _arr(pts_w).reshape(-1, 5)
    r = _arr(ranges, np.uint32).reshape(-1, 2)
    w = np.empty((r.shape[0], 5), dtype=np.uint64)
    wei = np.empty((r.shape[0], 11), dtype=np.uint64)
    _ck(load().mp2g_curve_sum_ranges(ctx.h, _p(a), a.shape[0], _p(r), r.shape[0], _p(w), _p(wei)))
    return w, wei


def scalar_mul_batch(ctx, pts_w, scalars, weierstrass=False):
    """scalars: python ints < 2^128 (hash_to_int_value range). Returns the encodings [count][5] (and the 11-limb Weierstrass
    forms = Point::to_fields when asked)."""
    a = _arr(pts_w).reshape(-1, 5)
    k = _arr([[(int(s) >> (32 * i)) & 0xFFFFFFFF for i in range(4)] for s in scalars], np.uint32).reshape(-1, 4)
    out = np.empty((a.shape[0], 5), dtype=np.uint64)
    wei = np.empty((a.shape[0], 11), dtype=np.uint64) if weierstrass else None
    _ck(load().mp2g_scalar_mul_batch(ctx.h, _p(a), _p(k), a.shape[0], _p(out), _p(wei) if weierstrass else None))
    return (out, wei) if weierstrass else out


def field_hashed_scalar_mul(ctx, inputs, base_w, variant=POSEIDON2):
    a, b = _arr(inputs), _arr(base_w)
    w = np.empty(5, dtype=np.uint64)
    wei = np.empty(11, dtype=np.uint64)
    _ck(load().mp2g_field_hashed_scalar_mul(ctx.h, variant, _p(a), a.size, _p(b), _p(w), _p(wei)))
    return w, wei


def u256_to_limbs(values):
    """U256 -> 8 big-endian u32 words, most significant first (mp2-common/src/u256.rs:870-877)."""
    out = np.empty((len(values), 8), dtype=np.uint32)
    for i, v in enumerate(values):
        v = int(v)
        for j in range(8):
            out[i, j] = (v >> (32 * (7 - j))) & 0xFFFFFFFF
    return out


def compute_table_row_digest(ctx, col_ids, values, unique, variant=POSEIDON2):
    """compute_table_row_digest (mp2-v1/src/values_extraction/mod.rs:527-571).
    values: uint32 [rows][n_cols][8]; unique: uint32 [rows][n_unique][8]."""
    ids = _arr(col_ids)
    v = _arr(values, np.uint32)
    u = _arr(unique, np.uint32)
    rows, n_cols = v.shape[0], ids.size
    n_unique = u.shape[1] if u.ndim == 3 else 0
    w = np.empty(5, dtype=np.uint64)
    wei = np.empty(11, dtype=np.uint64)
    _ck(load().mp2g_row_digest_batch(ctx.h, variant, _p(ids), n_cols, _p(v), _p(u), n_unique, rows, _p(w), _p(wei)))
    return w, wei


def row_digests(ctx, col_ids, values, unique, variant=POSEIDON2):
    """the per-row terms of compute_table_row_digest, row_id * sum_c D(id_c || value_c): (encodings [rows][5], Weierstrass [rows][11])"""
    ids = _arr(col_ids)
    v = _arr(values, np.uint32)
    u = _arr(unique, np.uint32)
    rows, n_cols = v.shape[0], ids.size
    n_unique = u.shape[1] if u.ndim == 3 else 0
    w = np.empty((rows, 5), dtype=np.uint64)
    wei = np.empty((rows, 11), dtype=np.uint64)
    _ck(load().mp2g_row_digests(ctx.h, variant, _p(ids), n_cols, _p(v), _p(u), n_unique, rows, _p(w), _p(wei)))
    return w, wei


def leaf_permutations_queued():
    """permutations the Merkle leaf sponge has queued since the library was loaded (mp2g_stat_leaf_permutations)"""
    return int(load().mp2g_stat_leaf_permutations())


# ---- proof wire format (mp2-common/src/proof.rs) ---------------------------------------------
def serialize_proof(fp, num_constants, caps, openings, fri_proof, public_inputs):
    """bincode bytes of ProofWithPublicInputs (mp2-common/src/proof.rs:84-98 serialize_proof)."""
    caps, openings, fri_proof = _arr(caps), _arr(openings), _arr(fri_proof)
    pis = _arr(public_inputs)
    n = ctypes.c_size_t()
    _ck(load().mp2g_proof_serialize(ctypes.byref(fp), num_constants, _p(caps), _p(openings), _p(fri_proof), _p(pis), pis.size, None, ctypes.byref(n)))
    out = np.empty(n.value, dtype=np.uint8)
    _ck(load().mp2g_proof_serialize(ctypes.byref(fp), num_constants, _p(caps), _p(openings), _p(fri_proof), _p(pis), pis.size, _p(out), ctypes.byref(n)))
    return out.tobytes()


def deserialize_proof(fp, num_constants, data, n_public_inputs):
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    caps = np.zeros((fp.n_oracles, fp.cap_words), dtype=np.uint64)
    openings = np.zeros((fp.n_openings, 2), dtype=np.uint64)
    fri = np.zeros(fp.proof_words, dtype=np.uint64)
    pis = np.zeros(n_public_inputs, dtype=np.uint64)
    _ck(load().mp2g_proof_deserialize(ctypes.byref(fp), num_constants, _p(buf), ctypes.c_size_t(buf.size), _p(caps), _p(openings),
                                      _p(fri), _p(pis), n_public_inputs))
    return caps, openings, fri, pis


def serialize_proof_with_vk(proof_bytes, vk_cap, vk_circuit_digest):
    """ProofWithVK::serialize (mp2-common/src/proof.rs:42-52)."""
    pb = np.frombuffer(proof_bytes, dtype=np.uint8).copy()
    cap, dig = _arr(vk_cap).reshape(-1, 4), _arr(vk_circuit_digest)
    n = ctypes.c_size_t()
    _ck(load().mp2g_proof_with_vk_serialize(_p(pb), ctypes.c_size_t(pb.size), _p(cap), cap.shape[0], _p(dig), None, ctypes.byref(n)))
    out = np.empty(n.value, dtype=np.uint8)
    _ck(load().mp2g_proof_with_vk_serialize(_p(pb), ctypes.c_size_t(pb.size), _p(cap), cap.shape[0], _p(dig), _p(out), ctypes.byref(n)))
    return out.tobytes()


def deserialize_proof_with_vk(fp, num_constants, data, n_public_inputs, vk_cap_len=None):
    """ProofWithVK::deserialize (mp2-common/src/proof.rs:54-57): (caps, openings, fri words, public inputs) with caps[0] = the
    verifier key's constants_sigmas cap (what the prover hands out for a proof it made), and the key's circuit digest."""
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    n_cap = (1 << fp.cap_height) if vk_cap_len is None else int(vk_cap_len)
    caps = np.zeros((fp.n_oracles, fp.cap_words), dtype=np.uint64)
    openings = np.zeros((fp.n_openings, 2), dtype=np.uint64)
    fri = np.zeros(fp.proof_words, dtype=np.uint64)
    pis = np.zeros(n_public_inputs, dtype=np.uint64)
    vk_cap = np.zeros((n_cap, 4), dtype=np.uint64)
    dig = np.zeros(4, dtype=np.uint64)
    _ck(load().mp2g_proof_with_vk_deserialize(ctypes.byref(fp), num_constants, _p(buf), ctypes.c_size_t(buf.size), _p(caps), _p(openings),
                                              _p(fri), _p(pis), n_public_inputs, _p(vk_cap), n_cap, _p(dig)))
    if vk_cap.size == caps[0].size:
        caps[0] = vk_cap.ravel()
    return (caps, openings, fri, pis), vk_cap, dig


# ---- recursion-framework pieces that sit on the path ------------------------------------------
# recursion-framework/src/universal_verifier_gadget/mod.rs:27-40, circuit_builder.rs:26
CIRCUIT_SET_CAP_HEIGHT = 0
RECURSION_THRESHOLD = 12
SHRINK_LIMIT = 15
MIN_CIRCUIT_SIZE = 64
NUM_HASH_OUT_ELTS = 4


def hash_pad_input(values):
    """plonky2 Hasher::hash_pad padding: append 1, zeros to rate-1 (mod 8), append 1."""
    v = [int(x) for x in values] + [1]
    while (len(v) + 1) % 8:
        v.append(0)
    return v + [1]


def circuit_digest(ctx, constants_sigmas_cap, degree_bits, variant=POSEIDON2, domain_separator=()):
    """recursion-framework/src/universal_verifier_gadget/circuit_set.rs:136-158:
    H(flatten(constants_sigmas_cap) || H_pad(domain separator) || degree_bits); the framework's circuits have no domain
    separator, a base circuit may (CircuitBuilder::set_domain_separator, wrap_circuit.rs:327-358)."""
    cap = [int(x) for x in _arr(constants_sigmas_cap).reshape(-1)]
    domain_sep = [int(x) for x in ctx.hash_no_pad(hash_pad_input(domain_separator), variant)]
    return ctx.hash_no_pad(cap + domain_sep + [degree_bits], variant)


class CircuitSet:
    """Merkle set of circuit digests (circuit_set.rs:173-237): leaves in insertion order, padded
    to a power of two with [0] leaves, cap height 0; the root is the 4 extra public inputs of
    every framework proof (circuit_builder.rs:169-171)."""

    def __init__(self, ctx, digests, variant=POSEIDON2):
        d = _arr(digests).reshape(-1, 4)
        self.digests = d
        n = max(1, d.shape[0])
        size = 1 << (n - 1).bit_length()
        leaves = np.zeros((size, 4), dtype=np.uint64)
        leaves[:d.shape[0]] = d
        self.tree = MerkleTree(ctx, leaves, CIRCUIT_SET_CAP_HEIGHT, variant)

    def circuit_set_digest(self):
        return self.tree.cap[0]

    def membership_proof(self, digest):
        """(leaf index little-endian bits, siblings) as set_circuit_membership_target assigns them
        (circuit_set.rs:205-237)."""
        digest = _arr(digest)
        hits = [i for i in range(self.digests.shape[0]) if np.array_equal(self.digests[i], digest)]
        if not hits:
            raise KeyError("circuit digest not found")  # circuit_set.rs:212 "circuit digest not found"
        idx = hits[0]
        _, sib = self.tree.prove([idx])
        bits = [(idx >> i) & 1 for i in range(self.tree.log_leaves)]
        return bits, sib[0]


def compute_table_row_digest_dev(ctx, d_col_ids, n_cols, d_values, d_unique, n_unique, rows, variant=POSEIDON2):
    """Device-resident form of compute_table_row_digest; returns the encoded digest."""
    w = np.empty(5, dtype=np.uint64)
    _ck(load().mp2g_row_digest_batch_dev(ctx.h, variant, d_col_ids.ptr, n_cols, d_values.ptr, d_unique.ptr, n_unique, rows,
                                         None, _p(w), None))
    return w


def partial_products_and_zs(ctx, wires, sigmas, betas, gammas, degree=8):
    """plonk/prover.rs all_wires_permutation_partial_products in prove()'s commit order."""
    wv, sg, b, g = _arr(wires), _arr(sigmas), _arr(betas), _arr(gammas)
    num_routed, n = sg.shape
    out = np.empty((b.size * (num_routed // degree), n), dtype=np.uint64)
    _ck(load().mp2g_partial_products_and_zs(ctx.h, _p(wv), wv.shape[0], _p(sg), int(n).bit_length() - 1, num_routed, degree,
                                            _p(b), _p(g), b.size, _p(out)))
    return out


def fri_prove(ctx, fp, oracles, zeta, challenger):
    """PolynomialBatch::prove_openings over committed batches (granular path for hosts that compute
    the quotient polynomials themselves). Returns the flat FriProof."""
    hs = (ctypes.c_void_p * len(oracles))(*[o.h.value for o in oracles])
    z = _arr(zeta)
    proof = np.empty(fp.proof_words, dtype=np.uint64)
    _ck(load().mp2g_fri_prove(ctx.h, ctypes.byref(fp), hs, _p(z), challenger.h, _p(proof)))
    return proof
