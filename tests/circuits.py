"""Oracle-side companions of the synthetic circuits (mapreduce-plonky2_amd/circuits.py): the CPU oracle's
prove(), PLONK identity check and constraint evaluation for a built circuit. Test infrastructure.
"""
import ctypes
import importlib

import numpy as np

import oracle as O

_pc = importlib.import_module("mapreduce-plonky2_amd.circuits")
globals().update({k: v for k, v in vars(_pc).items() if not k.startswith("__")})


def eval_on_points(ckt, consts, wires):
    """C_j at the given points: consts [num_constants][npts], wires [135][npts] -> [maxc][npts]"""
    consts, wires = O.arr(consts), O.arr(wires)
    npts = wires.shape[1]
    maxc = max(gate_num_constraints(g) for g in ckt.gates)
    out = np.zeros((maxc, npts), dtype=np.uint64)
    ph = O.arr(ckt.pi_hash)
    O.lib().orc_gates_eval_points(ckt.gate_array, len(ckt.gates), ckt.num_selectors, consts.shape[0], O.p(consts), wires.shape[0],
                                  O.p(wires), O.sz(npts), O.p(ph), O.p(out))
    return out


def prove_witness(ckt, fp, circuit_digest, wires, pi_hash):
    """oracle prove() of the circuit for the given witness and public-inputs hash; returns (caps, openings, proof, bgao)"""
    n = 1 << ckt.log_n
    vals = [O.arr(ckt.pre), O.arr(wires), np.zeros((fp.oracle_w[2], n), dtype=np.uint64), np.zeros((fp.oracle_w[3], n), dtype=np.uint64)]
    ptrs = (ctypes.c_void_p * 4)(*[v.ctypes.data for v in vals])
    capw = 4 << fp.cap_height
    caps = np.zeros((fp.n_oracles, capw), dtype=np.uint64)
    openings = np.zeros((O.lib().orc_n_openings(ctypes.byref(fp)), 2), dtype=np.uint64)
    proof = np.zeros(O.lib().orc_fri_proof_words(ctypes.byref(fp)), dtype=np.uint64)
    cd, ph, bgao = O.arr(circuit_digest), O.arr(pi_hash), np.zeros(8, dtype=np.uint64)
    O.lib().orc_pcs_prove_gates(ctypes.byref(fp), ptrs, O.p(cd), O.p(ph), NUM_ROUTED, 8, ckt.gate_array, len(ckt.gates),
                                ckt.num_selectors, O.p(bgao), O.p(caps), O.p(openings), O.p(proof))
    return caps, openings, proof, bgao


def prove(ckt, fp, circuit_digest):
    """oracle prove() of the circuit with its own witness; returns (caps, openings, proof, bgao)"""
    return prove_witness(ckt, fp, circuit_digest, ckt.wires, ckt.pi_hash)


def verify(ckt, fp, circuit_digest, pi_hash, caps, openings, proof):
    """oracle verify(): challenges from the transcript, PLONK identity with the gate terms, FRI. 0 = accept."""
    cd, ph = O.arr(circuit_digest), O.arr(pi_hash)
    caps, openings, proof = O.arr(caps), O.arr(openings), O.arr(proof)
    return O.lib().orc_verify_gates(ctypes.byref(fp), O.p(cd), O.p(ph), NUM_ROUTED, 8, ckt.gate_array, len(ckt.gates), ckt.num_selectors,
                                    O.p(caps), O.p(openings), O.p(proof))


def identity_check(ckt, fp, openings, bgao):
    class G2(ctypes.Structure):
        _fields_ = [("c", ctypes.c_uint64 * 2)]
    z = G2()
    z.c[0], z.c[1] = int(bgao[6]), int(bgao[7])
    o, b, g, a, ph = O.arr(openings), O.arr(bgao[0:2]), O.arr(bgao[2:4]), O.arr(bgao[4:6]), O.arr(ckt.pi_hash)
    return O.lib().orc_plonk_identity_check_gates(ctypes.byref(fp), NUM_ROUTED, 8, O.p(o), z, O.p(b), O.p(g), O.p(a), ckt.gate_array,
                                                  len(ckt.gates), ckt.num_selectors, O.p(ph))
