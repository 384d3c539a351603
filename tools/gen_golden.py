#!/usr/bin/env python3
"""Generate tests/golden/oracle_vectors.json from the CPU oracle (fixed SplitMix64 seeds).

The reference cannot run here (Rust, no toolchain), so these are *oracle* outputs: they freeze the
oracle's behaviour (regression protection) and give the GPU tests committed data to hit besides the
live oracle. Vectors that come from the reference tree itself live in sswu_kat.json / hash_kat.json.
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402


def fnv(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def main():
    out = {"_generator": "tools/gen_golden.py (oracle outputs, SplitMix64 seeds as listed)"}
    x = O.rand_field((4, 9), 0xC0FFEE04)
    out["hash_no_pad_9_to_4"] = {"seed": "0xC0FFEE04", "poseidon2": O.hash_no_pad_batch(x, 4, 0).tolist(),
                                 "poseidon": O.hash_no_pad_batch(x, 4, 1).tolist()}
    a = O.rand_field((1, 16), 0xC0FFEE02)
    out["ntt_16"] = {"seed": "0xC0FFEE02", "forward": O.fft(a)[0].tolist(), "coset_g": O.fft(a, coset_shift=O.MULT_GEN)[0].tolist()}
    vals = O.rand_field((5, 64), 0xC0FFEE01)
    coeffs = O.fft(vals, inverse=True)
    levels = O.merkle_build(O.lde_leaves(coeffs, 3), 4, 0)
    out["commit_5x64"] = {"seed": "0xC0FFEE01", "cap_fnv1a": fnv(O.merkle_cap(levels, 4)), "cap0": O.merkle_cap(levels, 4)[0].tolist()}
    ws = (5, 9, 4, 3)
    ofp = O.standard_params(6, ws, pow_bits=6, num_queries=4)
    pv = [O.rand_field((w, 64), 100 + i) for i, w in enumerate(ws)]
    caps, openings, proof = O.pcs_prove(ofp, pv, O.rand_field(4, 1), O.rand_field(4, 2))
    out["pcs_prove_2p6"] = {"oracle_w": list(ws), "pow_bits": 6, "num_queries": 4, "value_seeds": [100, 101, 102, 103],
                            "digest_seed": 1, "pi_seed": 2, "proof_fnv1a": fnv(proof), "openings_fnv1a": fnv(openings),
                            "pow_witness": int(proof[-1])}
    ins = O.rand_field((3, 9), 7)
    w = np.zeros((3, 5), dtype=np.uint64)
    wei = np.zeros((3, 11), dtype=np.uint64)
    O.lib().orc_map_to_curve_batch(0, O.p(ins), O.sz(9), O.sz(3), O.p(w), O.p(wei))
    out["map_to_curve_9"] = {"seed": 7, "encodings": w.tolist(), "weierstrass": wei.tolist()}
    rng = np.random.default_rng(20)
    col_ids = O.rand_field(4, 0xC0FFEE04)
    values = rng.integers(0, 1 << 32, size=(10, 4, 8), dtype=np.uint32)
    unique = values[:, :1, :].copy()
    dw = np.zeros(5, dtype=np.uint64)
    O.lib().orc_row_digest_batch(0, O.p(col_ids), O.sz(4), O.p(O.arr(values, np.uint32)), O.p(O.arr(unique, np.uint32)), O.sz(1), O.sz(10), O.p(dw), None)
    out["row_digest_10x4"] = {"numpy_default_rng": 20, "col_id_seed": "0xC0FFEE04", "unique": "first value column", "encoding": dw.tolist()}
    # gate constraint evaluators: every kind at 5 fixed random points (first filtered constraint values + a
    # checksum of all), and the fingerprint of one complete gate-level proof
    import circuits as C
    ckt = C.build(5, C.ALL_KINDS, 77)
    consts, wires = O.rand_field((ckt.num_constants, 5), 11), O.rand_field((C.NUM_WIRES, 5), 12)
    consts[:ckt.num_selectors, :] = np.arange(5, dtype=np.uint64)[None, :] + np.arange(ckt.num_selectors, dtype=np.uint64)[:, None] * np.uint64(3)
    ev = C.eval_on_points(ckt, consts, wires)
    gates = [[g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end] for g in ckt.gates]
    out["gate_constraints_5pts"] = {"circuit_seed": 77, "const_seed": 11, "wire_seed": 12, "gates": gates, "num_selectors": ckt.num_selectors,
                                    "pi_hash": ckt.pi_hash.tolist(), "c0": ev[0].tolist(), "c1": ev[1].tolist(), "all_fnv1a": fnv(ev)}
    ofp = O.standard_params(5, (ckt.num_constants + C.NUM_ROUTED, C.NUM_WIRES, 20, 16), pow_bits=5, num_queries=3)
    gc, go, gp, _ = C.prove(ckt, ofp, O.rand_field(4, 3))
    out["gate_level_proof_2p5"] = {"circuit_seed": 77, "digest_seed": 3, "pow_bits": 5, "num_queries": 3, "wires_fnv1a": fnv(ckt.wires),
                                   "pre_fnv1a": fnv(ckt.pre), "caps_fnv1a": fnv(gc), "openings_fnv1a": fnv(go), "proof_fnv1a": fnv(gp)}
    path = os.path.join(ROOT, "tests", "golden", "oracle_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
