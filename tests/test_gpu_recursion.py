"""Real recursion through the HIP prover: the base proof (map circuit, integration.rs:65-93) and the wrap proof that
verifies it in-circuit (recursion.py: plonky2's verify_proof gadget, wrap_circuit.rs:58-99) are both produced by
libmp2gpu at standard_recursion_config, equal the oracle's proofs of the same witnesses bit for bit, and pass the
oracle's verifier; the wrap proof carries the base proof's public inputs."""
import importlib

import numpy as np
import pytest

import circuits as C
import oracle as O

pytestmark = pytest.mark.gpu
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
FW = importlib.import_module("mapreduce-plonky2_amd.framework")


def gpu_prove(ctx, ckt, witness_check=True):
    cp = FW.CircuitProver(ctx, ckt, 1, witness_check=witness_check)
    cp.prove(ctx.to_device(ckt.wires[None]), ctx.to_device(ckt.pi_hash[None]))
    flags = cp.pr.witness_status() if witness_check else None
    caps, openings, proofs = cp.results()
    out = (cp, caps[0], openings[0], proofs[0], flags)
    return out


def test_base_wrap_verify_on_gpu(ctx, mp2):
    data = O.rand_field(4, 77)
    base = R.map_circuit(data)
    cp, caps, openings, proof, flags = gpu_prove(ctx, base)
    assert flags.tolist() == [0]
    fp = C.oracle_params(base)
    assert bytes(fp) == bytes(cp.fp)
    assert C.verify(base, fp, cp.circuit_digest, base.pi_hash, caps, openings, proof) == 0
    inner = R.InnerCircuit(base, fp, cp.constants_sigmas_cap, cp.circuit_digest, len(base.public_inputs))
    wrap = R.wrap_circuit(inner, caps, openings, proof, base.public_inputs)
    assert wrap.log_n == 12
    wcp, wc, wo, wp, wflags = gpu_prove(ctx, wrap)
    assert wflags.tolist() == [0]
    wfp = C.oracle_params(wrap)
    oc, oo, op_, _ = C.prove(wrap, wfp, wcp.circuit_digest)
    assert np.array_equal(wc, oc) and np.array_equal(wo, oo) and np.array_equal(wp, op_)
    assert C.verify(wrap, wfp, wcp.circuit_digest, wrap.pi_hash, wc, wo, wp) == 0
    assert np.array_equal(wrap.public_inputs, base.public_inputs)
    assert int(wrap.public_inputs[0]) == sum(int(x) for x in data if int(x) % 2 == 0) % O.P
    # a second wrap step (wrap_circuit.rs: the loop keeps wrapping until the size stops shrinking): the wrap proof itself
    # is verified by a circuit whose gate evaluators now cover the recursive verifier's own gate set
    inner2 = R.InnerCircuit(wrap, wfp, wcp.constants_sigmas_cap, wcp.circuit_digest, len(wrap.public_inputs))
    wrap2 = R.wrap_circuit(inner2, wc, wo, wp, wrap.public_inputs)
    w2cp, w2c, w2o, w2p, w2flags = gpu_prove(ctx, wrap2)
    assert w2flags.tolist() == [0]
    w2fp = C.oracle_params(wrap2)
    assert C.verify(wrap2, w2fp, w2cp.circuit_digest, wrap2.pi_hash, w2c, w2o, w2p) == 0
    assert np.array_equal(wrap2.public_inputs, base.public_inputs)
    for p in (cp, wcp, w2cp):
        p.free()


def test_dishonest_wrap_witness_is_flagged(ctx, mp2):
    data = O.rand_field(4, 3)
    base = R.map_circuit(data)
    cp, caps, openings, proof, _ = gpu_prove(ctx, base)
    inner = R.InnerCircuit(base, C.oracle_params(base), cp.constants_sigmas_cap, cp.circuit_digest, len(base.public_inputs))
    bad = proof.copy()
    bad[-1] = (int(bad[-1]) + 1) % O.P  # another proof-of-work witness: the response loses its leading zeros
    wrap = R.wrap_circuit(inner, caps, openings, bad, base.public_inputs, strict=False)
    wcp = FW.CircuitProver(ctx, wrap, 1, witness_check=True)
    wcp.prove(ctx.to_device(wrap.wires[None]), ctx.to_device(wrap.pi_hash[None]))
    with pytest.raises(mp2.Mp2gError):
        wcp.pr.witness_status()
    cp.free()
    wcp.free()


def test_wrapping_base_circuit_with_domain_separator(ctx, mp2):
    """recursion-framework/src/universal_verifier_gadget/wrap_circuit.rs:327-358: a base circuit of more than 2^12 Noop rows
    (degree 13) with a domain separator and one public input; its circuit digest -- H(cap || H_pad(separator) || 13) -- is
    what the proof's transcript starts from and what the wrap circuit holds as a constant. The wrapped proof carries the
    public input and passes the oracle's verifier."""
    public_input, separator = int(O.rand_field(1, 5)[0]), int(O.rand_field(1, 6)[0])
    b = R.Builder()
    b.register_public_inputs([b.add_virtual(public_input)])
    b.set_domain_separator([separator])
    base = b.build(min_log_n=13)
    assert base.log_n == 13 and base.domain_separator == [separator]
    cp, caps, openings, proof, flags = gpu_prove(ctx, base)
    assert flags.tolist() == [0]
    plain = mp2.circuit_digest(ctx, cp.constants_sigmas_cap, 13)
    assert not np.array_equal(cp.circuit_digest, plain)  # the separator is part of the digest
    assert np.array_equal(cp.circuit_digest, O.hash_n_to_m_no_pad(
        np.concatenate([np.asarray(cp.constants_sigmas_cap, dtype=np.uint64).ravel(), O.hash_n_to_m_no_pad(np.array(mp2.hash_pad_input([separator]), dtype=np.uint64), 4),
                        np.array([13], dtype=np.uint64)]), 4))
    fp = C.oracle_params(base)
    assert C.verify(base, fp, cp.circuit_digest, base.pi_hash, caps, openings, proof) == 0
    assert C.verify(base, fp, plain, base.pi_hash, caps, openings, proof) != 0  # a verifier that ignores the separator rejects
    inner = R.InnerCircuit(base, fp, cp.constants_sigmas_cap, cp.circuit_digest, len(base.public_inputs))
    wrap = R.wrap_circuit(inner, caps, openings, proof, base.public_inputs)
    assert wrap.log_n == R.RECURSION_THRESHOLD
    wcp, wcaps, wopen, wproof, wflags = gpu_prove(ctx, wrap)
    assert wflags.tolist() == [0]
    assert int(wrap.public_inputs[0]) == public_input
    assert C.verify(wrap, C.oracle_params(wrap), wcp.circuit_digest, wrap.pi_hash, wcaps, wopen, wproof) == 0


def test_verify_proof_with_fixed_circuit(ctx, mp2):
    """mp2-common/src/proof.rs:171-262: two circuits with the same common data and different verifier data (a constant, 42 or
    24, tied to the first of four public inputs); a verifier circuit built for one of them (verify_proof_fixed_circuit: its
    verifier data as constants) verifies that circuit's proofs and refuses the other's, both ways."""
    def test_circuit(value):
        b = R.Builder()
        pis = [b.add_virtual(value) for _ in range(4)]
        b.connect(pis[0], b.constant(value))
        b.register_public_inputs(pis)
        return b.build()

    circuits = [test_circuit(42), test_circuit(24)]
    assert circuits[0].log_n == circuits[1].log_n and [g.kind for g in circuits[0].gates] == [g.kind for g in circuits[1].gates]
    assert not np.array_equal(circuits[0].pre, circuits[1].pre)
    proved = []
    for ckt in circuits:
        cp, caps, openings, proof, flags = gpu_prove(ctx, ckt)
        assert flags.tolist() == [0]
        inner = R.InnerCircuit(ckt, C.oracle_params(ckt), cp.constants_sigmas_cap, cp.circuit_digest, 4)
        proved.append((ckt, inner, caps, openings, proof))
    assert not np.array_equal(proved[0][1].circuit_digest, proved[1][1].circuit_digest)
    for own, other in ((0, 1), (1, 0)):
        ckt, inner, caps, openings, proof = proved[own]
        verifier = R.wrap_circuit(inner, caps, openings, proof, ckt.public_inputs)
        wcp, wcaps, wopen, wproof, wflags = gpu_prove(ctx, verifier)
        assert wflags.tolist() == [0]
        assert C.verify(verifier, C.oracle_params(verifier), wcp.circuit_digest, verifier.pi_hash, wcaps, wopen, wproof) == 0
        o_ckt, _, o_caps, o_open, o_proof = proved[other]
        with pytest.raises(AssertionError):
            R.wrap_circuit(inner, o_caps, o_open, o_proof, o_ckt.public_inputs)
        # the same through prove(): the witness of the foreign proof does not satisfy the verifier circuit
        bad = R.wrap_circuit(inner, o_caps, o_open, o_proof, o_ckt.public_inputs, strict=False)
        assert np.array_equal(bad.pre, verifier.pre)
        bcp = FW.CircuitProver(ctx, bad, 1, witness_check=True)
        bcp.prove(ctx.to_device(bad.wires[None]), ctx.to_device(bad.pi_hash[None]))
        with pytest.raises(Exception, match="witness"):
            bcp.pr.witness_status()


def test_wires_from_rows_on_the_device(ctx, mp2):
    """mp2g_wires_from_rows_dev: the witness executor's row layout [batch][n][135] becomes the prover's [batch][135][n]"""
    for log_n, batch in ((6, 3), (12, 2), (13, 1)):
        n = 1 << log_n
        rows = O.rand_field((batch, n, 135), 700 + log_n)
        d_rows, d_w = ctx.to_device(rows), ctx.alloc(rows.nbytes)
        ctx.wires_from_rows_dev(d_rows, d_w, log_n, batch)
        assert np.array_equal(d_w.download((batch, 135, n)), rows.transpose(0, 2, 1))
        d_rows.free(); d_w.free()
