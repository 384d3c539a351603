"""Checker-side restatement of mapreduce-plonky2_amd/table.py's off-circuit data (TableWitness) from the CPU oracle's Ecgfp5 and
hashing primitives, one call per point, and a small harness that runs a table build on a given prover. Test infrastructure."""
import importlib

import numpy as np

import oracle as O

T = importlib.import_module("mapreduce-plonky2_amd.table")


def o_map(ins, variant=0):
    ins = O.arr(ins)
    w = np.zeros((ins.shape[0], 5), dtype=np.uint64)
    O.lib().orc_map_to_curve_batch(variant, O.p(ins), O.sz(ins.shape[1]), O.sz(ins.shape[0]), O.p(w), None)
    return w


def o_sum(ws):
    ws = O.arr(np.asarray(ws, dtype=np.uint64).reshape(-1, 5))
    w, wei = np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
    assert O.lib().orc_curve_sum(O.p(ws), O.sz(ws.shape[0]), O.p(w), O.p(wei))
    return w, wei


class OracleTableWitness:
    """the fields of table.TableWitness by the oracle (SplitDigestPoint accumulation node by node)"""

    def __init__(self, table, row_spans, variant=0):
        rows, C = table.rows, table.n_cols
        limbs = table.values.astype(np.uint64)
        self.cell_digest = np.zeros((rows, C, 11), dtype=np.uint64)
        for r in range(rows):
            cw = o_map(np.concatenate([table.col_ids[1:, None], limbs[r, 1:, :]], axis=1), variant)
            for k in range(1, C + 1):
                lo, hi = T.sbbst_span(C, k)
                self.cell_digest[r, k - 1] = o_sum(cw[lo - 1:hi])[1]
        self.unique = O.hash_no_pad_batch(limbs[:, 0, :], 4, variant)
        self.row_w = np.zeros((rows, 5), dtype=np.uint64)
        self.row_own = np.zeros((rows, 11), dtype=np.uint64)
        for r in range(rows):
            v = O.arr(table.values[r:r + 1], np.uint32)
            u = O.arr(table.values[r:r + 1, 0:1], np.uint32)
            O.lib().orc_row_digest_batch(variant, O.p(O.arr(table.col_ids)), O.sz(C + 1), O.p(v), O.p(u), O.sz(1), O.sz(1), O.p(self.row_w[r]), O.p(self.row_own[r]))
        self.row_digest, self.root_digest_w = {}, {}
        for k, (lo, hi) in row_spans.items():
            self.root_digest_w[k], self.row_digest[k] = o_sum(self.row_w[lo:hi])
