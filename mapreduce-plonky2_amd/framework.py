"""Host-side mirror of the recursion framework's proving calls over libmp2gpu.

`CircuitProver` is `CircuitData::prove` for a batch of witnesses of one circuit (everything prove()
does after witness generation runs on the device, see csrc/prover.hip). `FrameworkProver` is
`CircuitWithUniversalVerifier::generate_proof` (recursion-framework/src/circuit_builder.rs:286-311):
a base proof followed by the wrap proof down to 2^12 rows (wrap_circuit.rs:122-148), both carrying
the same public inputs -- the circuit's own NUM_PUBLIC_INPUTS followed by the 4 limbs of the
circuit-set digest (circuit_builder.rs:169-171). `MapReduce` is the reference's own integration
workload (recursion-framework/tests/integration.rs:138-261: a map circuit over chunks of a dataset,
a 2-to-1 reduce circuit above it) with the public-input chain computed as that test computes it.

The circuits proved here are the synthetic gate-level circuits of circuits.py (the gate sets of a
leaf and of plonky2's recursive verifier, random satisfied rows): the proving work per node is that
of the reference's node, the circuit logic is not (a parent's witness does not contain its
children's proofs; recursion.py builds the real verifier circuit for small trees).
"""
import numpy as np

from . import (BatchedProver, CircuitSet, Gate, POSEIDON2, PolynomialBatch, circuit_digest, standard_recursion_params)
from . import circuits as C

NUM_ROUTED = C.NUM_ROUTED
# recursion-framework/tests/integration.rs:55: the sum of the even elements and the hash of the elements
NUM_PUBLIC_INPUTS = 1 + 4
INPUT_CHUNK_SIZE = 4


def circuit_fri_params(ckt, variant=POSEIDON2, **kw):
    """FRI / oracle shape of a built circuit under standard_recursion_config: constants + sigmas, 135 wires,
    2 x (1 + 9) Z / partial products (+ 2 x 7 lookup polynomials when the circuit has lookup tables), 2 x 8 quotient
    chunks."""
    nlp = getattr(ckt, "num_lookup_polys", 0)
    return standard_recursion_params(ckt.log_n, (int(ckt.pre.shape[0]), C.NUM_WIRES, 2 * (NUM_ROUTED // 8 + nlp), 16), variant=variant,
                                     num_lookup_polys=nlp, **kw)


class CircuitProver:
    """prove() of `batch` witnesses of one circuit per call, device resident (mp2g_prover with the permutation
    argument, the quotient and the gate table switched on)."""

    def __init__(self, ctx, ckt, batch, variant=POSEIDON2, witness_check=False, bind_public_inputs=False, **fri_kw):
        self.ctx, self.ckt, self.batch = ctx, ckt, batch
        self.fp = circuit_fri_params(ckt, variant, **fri_kw)
        pr = BatchedProver(ctx, self.fp, batch)
        self.d_pre = ctx.to_device(ckt.pre)
        pr.set_preprocessed(self.d_pre)
        pr.enable_permutation(NUM_ROUTED, 8)
        pr.enable_quotient()
        pr.set_gates([Gate(g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end) for g in ckt.gates],
                     ckt.num_selectors)
        if getattr(ckt, "luts", None):
            pr.set_lookups(ckt.luts)
        if witness_check:
            pr.enable_witness_check()
        if bind_public_inputs:
            assert ckt.pi_row is not None, "the circuit has no PublicInputGate"
            pr.bind_public_inputs(ckt.pi_row)
        self.pr = pr
        # VerifierOnlyCircuitData: the constants_sigmas cap and the circuit digest
        # H(cap || H_pad([]) || degree_bits) (circuit_set.rs:136-158), the first thing every transcript absorbs
        pre = PolynomialBatch.from_values(ctx, ckt.pre, self.fp.rate_bits, self.fp.cap_height, variant)
        self.constants_sigmas_cap = pre.cap
        pre.free()
        self.circuit_digest = circuit_digest(ctx, self.constants_sigmas_cap, ckt.log_n, variant, getattr(ckt, "domain_separator", ()))
        self.d_circuit_digest = ctx.to_device(self.circuit_digest)

    def prove(self, d_wires, d_pi_hash, d_circuit_digest=None):
        self.pr.prove([d_wires, None, None], d_circuit_digest or self.d_circuit_digest, d_pi_hash)

    def results(self):
        return self.pr.results()

    def free(self):
        self.pr.free()


def tile_witness(ctx, ckt, batch, seed, rand_row=True):
    """[batch][135][n] wire matrices on the device from the circuit's one witness. With rand_row the free
    (unrouted) cells of one Noop row are re-drawn per proof so that the commitments, transcripts and proofs of
    the batch all differ; witness_of(ckt, seed, b) gives proof b's matrix back."""
    w, n = ckt.wires.shape
    buf = ctx.alloc(batch * w * n * 8)
    one = ckt.wires.copy()
    row = noop_row(ckt) if rand_row else None
    for b in range(batch):
        if row is not None:
            one[NUM_ROUTED:, row] = C.rand_field(w - NUM_ROUTED, seed + b)
        buf.upload_at(one, b * w * n * 8)
    return buf


def noop_row(ckt):
    return ckt.instances.index(next(i for i, g in enumerate(ckt.gates) if g.kind == C.NOOP))


def witness_of(ckt, seed, b, pi_hash=None):
    """the wire matrix of proof b of tile_witness(ctx, ckt, ., seed) (host copy, for the checker)"""
    one = ckt.wires.copy()
    one[NUM_ROUTED:, noop_row(ckt)] = C.rand_field(one.shape[0] - NUM_ROUTED, seed + b)
    if pi_hash is not None:
        one[0:4, ckt.pi_row] = pi_hash
    return one


class FrameworkProver:
    """generate_proof for `batch` nodes at a time: base prove() (2^base_bits rows, the leaf gate set) on one
    context / stream, wrap prove() (2^12 rows, the recursive-verifier gate set) on another, both bound to the
    node's public-inputs hash."""

    def __init__(self, ctx_base, ctx_wrap, batch, base_bits=13, wrap_bits=12, variant=POSEIDON2, seed=0xC0FFEE03,
                 base_kinds=None, wrap_kinds=None, witness_check=False, circuits=None):
        self.batch, self.variant, self.seed = batch, variant, seed
        # circuits = (base, wrap) built earlier: provers of several batch sizes share one pair of circuits
        self.base_ckt, self.wrap_ckt = circuits or (C.build(base_bits, base_kinds or C.LEAF_KINDS, seed + base_bits),
                                                    C.build(wrap_bits, wrap_kinds or C.VERIFIER_KINDS, seed + wrap_bits + 100))
        self.base = CircuitProver(ctx_base, self.base_ckt, batch, variant, witness_check, bind_public_inputs=True)
        self.wrap = CircuitProver(ctx_wrap, self.wrap_ckt, batch, variant, witness_check, bind_public_inputs=True)
        self.d_base_w = tile_witness(ctx_base, self.base_ckt, batch, seed)
        self.d_wrap_w = tile_witness(ctx_wrap, self.wrap_ckt, batch, seed + 7)
        self.d_ph = [ctx_base.alloc(batch * 32), ctx_wrap.alloc(batch * 32)]
        self.digests = [self.base.circuit_digest, self.wrap.circuit_digest]

    def generate_proofs(self, pi_hashes):
        """pi_hashes [m <= batch][4]: launches the base and the wrap proofs of m nodes (asynchronous)."""
        m = len(pi_hashes)
        buf = np.zeros((self.batch, 4), dtype=np.uint64)
        buf[:m] = pi_hashes
        for ph in self.d_ph:
            ph.upload(buf)
        self.base.prove(self.d_base_w, self.d_ph[0])
        self.wrap.prove(self.d_wrap_w, self.d_ph[1])

    def results(self):
        return self.base.results(), self.wrap.results()

    def free(self):
        self.base.free()
        self.wrap.free()


class MapReduce:
    """recursion-framework/tests/integration.rs:138-261 over `n_leaves` map proofs: leaf i covers
    dataset[4i .. 4i+4) and exposes (sum of its even elements, H(chunk)); a reduce node exposes
    (sum of its children's sums, H(children's hashes)); every proof's public inputs end with the circuit-set
    digest. Levels are proved bottom-up in chunks of `chunk` nodes (children before parents, as ryhope's work
    plan orders them)."""

    def __init__(self, ctx_base, ctx_wrap, n_leaves, chunk=128, variant=POSEIDON2, seed=0xC0FFEE03, base_bits=13, wrap_bits=12, data_seed=None):
        assert n_leaves >= 1 and n_leaves & (n_leaves - 1) == 0
        self.ctx, self.n_leaves, self.variant = ctx_base, n_leaves, variant
        top = min(chunk, n_leaves)
        self.fw = FrameworkProver(ctx_base, ctx_wrap, top, base_bits=base_bits, wrap_bits=wrap_bits, variant=variant, seed=seed)
        # the narrow levels near the root go through smaller provers of the same circuits (a level of 4 nodes must not
        # cost a batch of 128)
        self.fws = [self.fw] + [FrameworkProver(ctx_base, ctx_wrap, b, variant=variant, seed=seed, circuits=(self.fw.base_ckt, self.fw.wrap_ckt))
                                for b in (64, 32, 16, 8, 4, 2, 1) if b < top]
        # `seed` fixes the circuits (every rank of a job builds the same ones), data_seed this instance's share of the dataset
        self.dataset = C.rand_field(n_leaves * INPUT_CHUNK_SIZE, seed if data_seed is None else data_seed)
        self.levels = []

    def prover_for(self, n_nodes):
        return min((f for f in self.fws if f.batch >= n_nodes), key=lambda f: f.batch, default=self.fw)

    def leaf_public_inputs(self):
        chunks = self.dataset.reshape(self.n_leaves, INPUT_CHUNK_SIZE)
        even = (chunks % np.uint64(2)) == 0
        sums = np.array([sum(int(x) for x, e in zip(row, ev) if e) % C.P for row, ev in zip(chunks, even)], dtype=np.uint64)
        hashes = self.ctx.hash_no_pad_batch(chunks, 4, self.variant)
        return np.concatenate([sums[:, None], hashes], axis=1)

    @staticmethod
    def reduce_public_inputs(ctx, child_pis, variant=POSEIDON2):
        """ReduceCircuitWires::circuit_logic (integration.rs:108-127), ARITY = 2"""
        m = child_pis.shape[0] // 2
        pairs = child_pis.reshape(m, 2, NUM_PUBLIC_INPUTS)
        sums = np.array([(int(a) + int(b)) % C.P for a, b in pairs[:, :, 0]], dtype=np.uint64)
        hashes = ctx.hash_no_pad_batch(pairs[:, :, 1:].reshape(m, 8), 4, variant)
        return np.concatenate([sums[:, None], hashes], axis=1)

    def run(self, keep=lambda level, index: False):
        """prove the whole tree; returns the root's public inputs (NUM_PUBLIC_INPUTS + 4). keep(level, index)
        selects the nodes whose proofs are retained in self.kept[(level, index)] = (pi, pi_hash, base, wrap) with
        base / wrap = (caps, openings, proof)."""
        fw = self.fw
        # the set of circuits of the framework: the map and the reduce circuit, both wrapped to the same size
        self.circuit_set = CircuitSet(self.ctx, np.stack([fw.digests[1], fw.digests[1]]), self.variant)
        set_digest = self.circuit_set.circuit_set_digest()
        self.kept, self.n_proofs = {}, 0
        pis = self.leaf_public_inputs()
        level = 0
        while True:
            full = np.concatenate([pis, np.tile(set_digest, (pis.shape[0], 1))], axis=1)
            hashes = self.ctx.hash_no_pad_batch(full, 4, self.variant)  # public_inputs_hash of prove()
            for lo in range(0, len(hashes), fw.batch):
                part = hashes[lo:lo + fw.batch]
                pv = self.prover_for(len(part))
                pv.generate_proofs(part)
                want = [i for i in range(len(part)) if keep(level, lo + i)]
                if want:
                    (bc, bo, bp), (wc, wo, wp) = pv.results()
                    for i in want:
                        self.kept[(level, lo + i)] = (full[lo + i], part[i], (bc[i], bo[i], bp[i]), (wc[i], wo[i], wp[i]))
                self.n_proofs += len(part)
            self.levels.append(full)
            if pis.shape[0] == 1:
                for c in {fw.base.ctx, fw.wrap.ctx}:
                    c.sync()
                return full[0]
            pis = self.reduce_public_inputs(self.ctx, pis, self.variant)
            level += 1

    def free(self):
        for f in self.fws:
            f.free()


class GpuProver:
    """The proving back end recursion.RecursiveCircuits drives: prove() of one circuit built by recursion.Builder on
    the device, the circuit's verifier data from the preprocessed commitment, and the Merkle node hash of the circuit
    set. One CircuitProver (preprocessed oracle, gate table) per distinct circuit, kept for reuse."""

    def __init__(self, ctx, variant=POSEIDON2, witness_check=True, capacity=0, device_witness=True):
        """capacity > 0: one prover per circuit, created for `capacity` proofs and used for every batch width up to that
        (mp2g_prover_set_active) -- the narrow levels of a tree then cost no device memory of their own; 0: one prover per
        (circuit, batch width)."""
        self.ctx, self.variant, self.witness_check, self.capacity = ctx, variant, witness_check, capacity
        # device_witness: generate_proofs_batch replays the witness programs on the device (prove_chain: mp2g_witness_program_run_dev,
        # the whole base + wrap chain queued on the stream, proofs handed from step to step by device copies); off: host threads
        # (mp2g_witness_program_run_rows) and an upload per step -- kept for A/B runs and as the second opinion of the parity tests
        self.device_witness = device_witness
        self.provers = {}
        self.chains = {}
        self.pinned = {}  # data address of a pinned wire matrix -> its host pointer

    rows_layout = True  # generate_proofs_batch fills [B][n][135] (rows) for this prover; prove_batch_launch transposes on the device

    def pinned_wires(self, shape):
        """a wire matrix ([B][135][n], or [B][n][135] in the row layout) in pinned host memory: prove_batch_launch sends it up with
        an asynchronous copy on the context's stream instead of a blocking pageable one"""
        nbytes = int(np.prod(shape)) * 8
        view, hptr = self.ctx.host_alloc(nbytes)
        a = view.view(np.uint64).reshape(shape)
        self.pinned[a.ctypes.data] = hptr
        return a

    @staticmethod
    def circuit_key(ckt):
        """identity of a circuit for the prover cache: a 128-bit digest of everything a CircuitProver is built from (preprocessed
        polynomials, gate table with selector groups, lookup tables, public-input row, domain separator)"""
        key = getattr(ckt, "_prover_key", None)
        if key is None:
            import hashlib
            h = hashlib.blake2b(digest_size=16)
            h.update(np.ascontiguousarray(ckt.pre).tobytes())
            h.update(repr((ckt.log_n, ckt.num_selectors, [(g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end) for g in ckt.gates],
                           getattr(ckt, "pi_row", None), list(getattr(ckt, "domain_separator", ())))).encode())
            for lut in (getattr(ckt, "luts", None) or []):
                h.update(repr((lut["last_lu_row"], lut["last_lut_row"], lut["first_lut_row"])).encode())
                h.update(np.ascontiguousarray(lut["table"], dtype=np.uint16).tobytes())
            key = ckt._prover_key = h.hexdigest()
        return key

    def _prover(self, ckt):
        self.ctx.make_current()
        key = (ckt.log_n, self.circuit_key(ckt))
        cp = self.provers.get(key)
        if cp is None:
            cp = self.provers[key] = CircuitProver(self.ctx, ckt, 1, self.variant, witness_check=self.witness_check)
        return cp

    def verifier_data(self, ckt):
        cp = self._prover(ckt)
        return cp.constants_sigmas_cap, cp.circuit_digest

    def prove(self, ckt):
        cp = self._prover(ckt)
        cp.prove(self.ctx.to_device(ckt.wires[None]), self.ctx.to_device(ckt.pi_hash[None]))
        if self.witness_check:
            cp.pr.witness_status()  # raises like plonky2's prove() on an unsatisfied witness
        caps, openings, proofs = cp.results()
        return caps[0], openings[0], proofs[0]

    def prove_batch_launch(self, ckt, wires, pi_hash):
        """upload B witnesses of one circuit (wires [B][135][n] or, in the witness executor's row layout, [B][n][135]; pi_hash
        [B][4], host) and queue their prove() on the context's stream; returns the handle prove_batch_finish waits on. One
        batch per (circuit, B) in flight at a time."""
        B, n = wires.shape[0], 1 << ckt.log_n
        self.ctx.make_current()
        rows = wires.shape[1:] == (n, 135) and n != 135
        cap = max(B, self.capacity) if self.capacity else B
        key = (ckt.log_n, self.circuit_key(ckt), cap)
        cp = self.provers.get(key)
        per = wires.nbytes // B
        if cp is None:
            cp = self.provers[key] = CircuitProver(self.ctx, ckt, cap, self.variant, witness_check=self.witness_check)
            cp.d_w, cp.d_ph, cp.d_rows = self.ctx.alloc(cap * per), self.ctx.alloc(cap * 32), None
        if rows and cp.d_rows is None:
            cp.d_rows = self.ctx.alloc(cap * per)
        if cap != B or getattr(cp.pr, "active", cap) != B:
            cp.pr.set_active(B)
        cp.d_ph.upload_at(np.ascontiguousarray(pi_hash, dtype=np.uint64), 0)
        dst = cp.d_rows if rows else cp.d_w
        hptr = self.pinned.get(wires.ctypes.data) if wires.flags["C_CONTIGUOUS"] else None
        if hptr is not None:
            self.ctx.h2d_async(dst, hptr, wires.nbytes)  # ordered before the kernels below on the same stream
        else:
            dst.upload_at(wires, 0)
        if rows:
            self.ctx.wires_from_rows_dev(cp.d_rows, cp.d_w, ckt.log_n, B)
        cp.prove(cp.d_w, cp.d_ph)
        return cp, B

    def _chain(self, ckts, progs, B):
        """the mp2g_chain of a framework circuit (base + wraps) on this context, created for `capacity` proofs (or B when no capacity
        is set) and kept: every narrower batch runs in the same device buffers"""
        from . import ProofChain
        cap = max(B, self.capacity) if self.capacity else B
        key = ("chain", tuple(self.circuit_key(c) for c in ckts), cap)
        ch = self.chains.get(key)
        if ch is None:
            cps = [CircuitProver(self.ctx, ckt, cap, self.variant, witness_check=self.witness_check) for ckt in ckts]
            ch = self.chains[key] = ProofChain(self.ctx, [cp.pr for cp in cps], progs, [cp.d_circuit_digest for cp in cps], cap)
            ch.cps = cps
        return ch

    def prove_chain(self, ckts, progs, cur, capture=None, name="", patches=()):
        """generate_proof's chain for B nodes without the host in the loop (csrc/chain.hip, mp2g_chain_run): `cur` [B][n_inputs] (host) are
        the base circuit's witness inputs, `patches` [(job, word offset, recursion.DeviceProof)] the child proofs that stay on the
        device. Per step the witness program runs on the device into the step's wire matrix, prove() follows on the same stream,
        and the next step's inputs are gathered from the prover's outputs by device copies. One synchronisation at the end. Returns
        [(caps, openings, proof, public_inputs)] of the last step; raises like plonky2's prove() on an unsatisfied witness."""
        B = cur.shape[0]
        self.ctx.make_current()  # worker threads of a rank whose GPU is not device 0
        ch = self._chain(ckts, progs, B)
        flat = []
        for j, off, dp in patches:
            for ptr, nw in dp.parts:
                flat.append((j, off, ptr, nw))
                off += nw
        caps, openings, proofs, pis = ch.run(np.ascontiguousarray(cur, dtype=np.uint64), flat)
        self.last_chain = ch
        if capture is not None:
            for step, (cp, ckt) in enumerate(zip(ch.cps, ckts)):
                n, fp, n_pr = 1 << ckt.log_n, cp.fp, int(progs[step].probe.size)
                d_w, d_probe, d_caps, d_open, d_proof = ch.step_buffers(step)
                w = self.ctx.d2h_raw(d_w, (B, 135, n))
                ph = self.ctx.d2h_raw(d_probe, (B, n_pr))[:, :4]
                c, o, p = (self.ctx.d2h_raw(d_caps, (B, fp.n_oracles, fp.cap_words)), self.ctx.d2h_raw(d_open, (B, fp.n_openings, 2)),
                           self.ctx.d2h_raw(d_proof, (B, fp.proof_words)))
                for b in range(B):
                    capture.append((name, step, ckt, cp.circuit_digest, w[b].copy(), ph[b].copy(), c[b], o[b], p[b]))
        return [(caps[b], openings[b], proofs[b], pis[b]) for b in range(B)]

    def last_device_proof(self, b=0):
        """proof b of the last prove_chain as a recursion.DeviceProof over the chain's output buffers: valid until the next run of
        that chain (hand it on -- to a parent's generate_proofs_batch or to another rank -- before that)"""
        from .recursion import DeviceProof
        return DeviceProof(self.last_chain.device_proof(b), keep=self.last_chain)

    def prove_batch_finish(self, handle):
        cp, B = handle
        if self.witness_check:
            cp.pr.witness_status()  # raises like plonky2's prove() on an unsatisfied witness
        caps, openings, proofs = cp.results()
        return [(caps[b], openings[b], proofs[b]) for b in range(B)]

    def prove_batch(self, ckt, wires, pi_hash):
        """wires [B][135][n] and pi_hash [B][4] (host) of B witnesses of one circuit -> [(caps, openings, proof)]"""
        return self.prove_batch_finish(self.prove_batch_launch(ckt, wires, pi_hash))

    def two_to_one(self, left, right):
        """Hasher::two_to_one = permute([l || r || 0000])[0..4] = hash_no_pad of the 8 limbs (one absorb)"""
        return [int(x) for x in self.ctx.hash_no_pad(list(left) + list(right), self.variant)]

    def free(self):
        for ch in self.chains.values():
            ch.free()
            for cp in ch.cps:
                cp.free()
        self.chains = {}
        for cp in self.provers.values():
            cp.free()
        self.provers = {}
        for hptr in self.pinned.values():
            self.ctx.host_free(hptr)
        self.pinned = {}
