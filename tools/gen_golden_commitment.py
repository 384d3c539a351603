#!/usr/bin/env python3
"""tests/golden/commitment_vectors.json: outputs of the CPU oracle's restatement of update_off_chain_data_commitment
(mp2-v1/src/api.rs:556-603) on fixed inputs -- freezes the oracle (regression protection) and gives the GPU test committed data."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402


def inputs():
    rng = np.random.default_rng(0xC0FFEE)
    rows, n_cols = 9, 3
    col_ids = O.rand_field(n_cols, 0xC0FFEE04)
    values = rng.integers(0, 1 << 32, size=(rows, n_cols, 8), dtype=np.uint32)
    groups = rng.integers(0, 1 << 32, size=(3, 8), dtype=np.uint32)
    primary = groups[np.array([2, 0, 1, 1, 0, 2, 2, 0, 1])]
    return 0xB10C, primary, col_ids, values, [0, 2], bytes(range(100, 132))


def main():
    pid, primary, col_ids, values, uniq, old = inputs()
    unique = np.ascontiguousarray(values[:, uniq, :])
    out = {"_generator": "tools/gen_golden_commitment.py (oracle outputs; numpy default_rng(0xC0FFEE), col ids rand_field(3, 0xC0FFEE04), primary id 0xB10C, "
                         "unique columns 0 and 2, old commitment bytes 100..131)"}
    for name, v in (("poseidon2", 0), ("poseidon", 1)):
        res = {}
        for key, oc in (("fresh", None), ("update", old)):
            o = np.zeros(32, dtype=np.uint8)
            ob = np.frombuffer(oc, dtype=np.uint8).copy() if oc is not None else None
            O.lib().orc_update_off_chain_data_commitment(v, ctypes.c_uint64(pid), O.p(O.arr(primary, np.uint32)), O.p(col_ids), O.sz(3), O.p(O.arr(values, np.uint32)),
                                                         O.p(O.arr(unique, np.uint32)), O.sz(2), O.sz(9), O.p(ob) if ob is not None else None, O.p(o))
            res[key] = o.tobytes().hex()
        out[name] = res
    with open(os.path.join(ROOT, "tests", "golden", "commitment_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)


if __name__ == "__main__":
    main()
