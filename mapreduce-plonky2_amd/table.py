"""Table creation (BASELINE configs[3]): the cells-tree and row-tree proving flow of verifiable-db on the recursion
framework, driven the way mp2-v1's harness drives it.

What the reference does per table row (mp2-v1/tests/common/celltree.rs:54-189, rowtree.rs:78-337): the row's value
columns form a cells tree (ryhope sbbst over the column positions 1..C: every node is a cell) whose nodes are proved
bottom-up by the cells-tree circuit set (verifiable-db/src/cells_tree/api.rs:100-250: leaf / full node / partial node /
empty node = 0 / 2 / 1 / 0 universal verifiers, 28 public inputs); then the row itself is a node of the row tree (a BST
over the secondary-index values: every node is a row) proved by the row-tree circuit set (row_tree/api.rs:22-160: leaf /
full / partial = 0 / 2 / 1 universal verifiers over row proofs PLUS one proof of the cells set through
RecursiveCircuitsVerifierGadget::verify_proof_in_circuit_set, 43 public inputs). The order of the row proofs is an
UpdateTree work plan (ryhope/src/storage/updatetree.rs:154-163,449-531).

The circuits here are those framework circuits with the tree logic of the reference:
  * node hashes in-circuit exactly as the reference's circuits compute them (cells_tree/{leaf,full_node,partial_node}.rs,
    row_tree/{leaf,full_node,partial_node}.rs; mp2-common/src/hash.rs:16-46 hash_maybe_first), so a root proof's hash is
    the off-circuit tree hash of indexing.py (mp2-v1/src/indexing/cell.rs:120-157, row.rs:257-317);
  * counters, min / max propagation, the multiplier checks between a node and its children, the row-id hash and its
    split into the 128-bit scalar (secondary_index_cell.rs:99-126, mp2-common/src/poseidon.rs:105-117), u32 range checks
    of the U256 limbs;
  * NOT in-circuit: the Ecgfp5 gadgets (map-to-curve, curve addition, scalar multiplication) and the u256 comparisons of
    the BST checks -- those circuit libraries (plonky2_ecgfp5 gadgets, u256 / u32 comparison gadgets) are host-side
    circuit definitions outside the hot path. The digest public inputs carry the values the
    reference's circuits would expose, computed off-circuit on the GPU (mp2g_map_to_curve_batch, mp2g_row_digests,
    mp2g_curve_sum_ranges), as unconstrained witnesses.
So a table build here proves, per row, the same NUMBER and SHAPE of framework proofs the reference proves (per row
C cells-tree proofs + 1 row-tree proof, same verifier counts, same public-input layout), each through witness generation,
base prove() and the wrap chain; the proving work per proof is a lower bound of the reference's (its leaf logic has the
curve gadgets on top).
"""
import os
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import curve_sum_ranges, map_to_curve_batch, row_digests, u256_to_limbs
from . import recursion as R
from . import workplan as W

CELLS_IO = 4 + 11 + 11 + 1 + 1   # cells_tree/public_inputs.rs:52-63: h, individual_vd, multiplier_vd, individual_cnt, multiplier_cnt
ROWS_IO = 4 + 11 + 11 + 8 + 8 + 1  # row_tree/public_inputs.rs:56-69: h, individual_vd, multiplier_vd, min, max, multiplier_cnt
NEUTRAL_FIELDS = [0] * 10 + [1]  # Point::NEUTRAL.to_fields(): x = y = 0, is_inf = 1
CELL_LEN = 1 + 8 + 1            # identifier, value (8 big-endian u32 limbs), is_multiplier


def _u64cat(*parts):
    """concatenate as u64 words (numpy would promote a mix of uint64 arrays and Python ints to float64)"""
    return np.concatenate([np.asarray(x, dtype=np.uint64).ravel() for x in parts])


# ---- tree shapes --------------------------------------------------------------------------------------------------------------------
def sbbst_root(n):
    """ryhope/src/tree/sbbst.rs:251-257"""
    return 1 << (n.bit_length() - 1) if n > 0 else 0


def _sat_children(k):
    """sbbst.rs:487-503 children_inner_in_saturated"""
    layer = (k & -k).bit_length() - 1
    if layer == 0:
        return None
    rank = (k - (1 << layer)) >> (layer + 1)
    cl = layer - 1
    return (2 * rank) * (1 << (cl + 1)) + (1 << cl), (2 * rank + 1) * (1 << (cl + 1)) + (1 << cl)


def sbbst_children(n, k):
    """sbbst.rs:301-333 children_inner: (left or None, right or None) of node k in the tree over 1..n"""
    ch = _sat_children(k)
    if ch is None:
        return None, None
    left = ch[0] if ch[0] <= n else None
    right = ch[1]
    if right <= n:
        return left, right
    if left is None:
        return None, None
    while right is not None and right > n:
        c = _sat_children(right)
        right = c[0] if c is not None else None
    return left, right


def sbbst_span(n, k):
    """the in-order range [lo, hi] of positions under node k (a subtree of a BST laid out in order is contiguous)"""
    layer = (k & -k).bit_length() - 1
    return k - (1 << layer) + 1, min(n, k + (1 << layer) - 1)


def cells_tree_live_peak(n_cols):
    """the most proofs of ONE row's cells tree alive at a time when the tree is proved level by level (levels = heights) and a child
    is dropped once its parent is proved: what a unit of the native build holds per row in the proof pool while it proves the
    cells trees (the root stays until the row node is proved)"""
    height = {}

    def h(k):
        if k not in height:
            height[k] = 1 + max([h(c) for c in sbbst_children(n_cols, k) if c is not None], default=-1)
        return height[k]

    for k in range(1, n_cols + 1):
        h(k)
    alive, peak = 0, 0
    for lvl in range(max(height.values()) + 1):
        here = [k for k in height if height[k] == lvl]
        alive += len(here)
        peak = max(peak, alive)
        alive -= sum(c is not None for k in here for c in sbbst_children(n_cols, k))
    return peak


def balanced_bst(n):
    """the BST a full rebuild gives over n keys in sorted order (node = midpoint of its range): root, {key: (left, right)},
    {key: (lo, hi)} with the subtree of `key` = keys [lo, hi)"""
    nodes, spans = {}, {}

    def rec(lo, hi):
        if lo >= hi:
            return None
        mid = (lo + hi) // 2
        nodes[mid] = (rec(lo, mid), rec(mid + 1, hi))
        spans[mid] = (lo, hi)
        return mid

    return rec(0, n), nodes, spans


# ---- circuit logic ------------------------------------------------------------------------------------------------------------------
def _cell_wire(b, v):
    """cells_tree/mod.rs:88-96 CellWire::new: identifier, U256 value (8 u32 limbs, range-checked), is_multiplier (a safe bool)"""
    ident = b.add_virtual(int(v[0]))
    value = [b.add_virtual(int(x)) for x in v[1:9]]
    for t in value:
        b.range_check(t, 32)
    m = b.add_virtual(int(v[9]))
    b.assert_bool(m)
    return ident, value, m


def _point(b, v):
    return [b.add_virtual(int(x)) for x in v]


def cells_logic(kind, empty_hash):
    """circuit_logic of cells_tree/{leaf.rs:24-50, full_node.rs:24-66, partial_node.rs:24-62, empty_node.rs:21-36}; inputs (flat):
    the cell (CELL_LEN) || individual digest (11) || multiplier digest (11) of the node's subtree"""
    def logic(b, child_pis, inputs):
        if kind == "empty":
            return [b.constant(int(x)) for x in empty_hash] + [b.constant(x) for x in NEUTRAL_FIELDS] * 2 + [b.zero(), b.zero()]
        v = inputs if inputs is not None else [0] * (CELL_LEN + 22)
        ident, value, m = _cell_wire(b, v[:CELL_LEN])
        ind_vd, mul_vd = _point(b, v[CELL_LEN:CELL_LEN + 11]), _point(b, v[CELL_LEN + 11:CELL_LEN + 22])
        empty = [b.constant(int(x)) for x in empty_hash]
        hs = [list(pis[0:4]) for pis in child_pis] + [empty, empty]
        h = b.hash_n_to_m_no_pad(hs[0] + hs[1] + [ident] + value, 4)   # H(left.h || right.h || identifier || value), missing child = H("")
        ind_cnt, mul_cnt = b.not_(m), m
        for pis in child_pis:
            ind_cnt, mul_cnt = b.add(ind_cnt, pis[26]), b.add(mul_cnt, pis[27])
        return h + ind_vd + mul_vd + [ind_cnt, mul_cnt]
    return logic


def hash_maybe_first(b, should_swap, elem1, elem2, rest):
    """mp2-common/src/hash.rs:16-46: H(elem1 || elem2 || rest), the first two exchanged when should_swap -- the swap rides on
    the first Poseidon2 gate's swap wire"""
    z = b.zero()
    state = b.permute_swapped(list(elem1) + list(elem2) + [z] * 4, should_swap)
    for i in range(0, len(rest), 8):
        chunk = list(rest[i:i + 8])
        state = b.permute(chunk + state[len(chunk):])
    return state[:4]


ROW_LEN = CELL_LEN + 4 + 22  # the secondary-index cell, row_unique_data, individual digest, multiplier digest


def rows_logic(kind, gadget, empty_hash):
    """circuit_logic of row_tree/{leaf.rs:33-72 + 96-118, full_node.rs:32-104, partial_node.rs:52-134}: verify the row's cells-tree
    root proof against the cells circuit set, then the node logic over its public inputs and the children's. inputs:
    ((cells proof, verifier data, membership), flat) with flat = cell || row_unique_data (4) || individual digest (11) ||
    multiplier digest (11) [|| is_child_at_left]"""
    def logic(b, child_pis, inputs):
        if inputs is None:
            cells_proof, v = gadget.dummy_inputs(), [0] * (ROW_LEN + 1)
        else:
            cells_proof, v = inputs
        cpis = gadget.verify_proof_in_circuit_set(b, *cells_proof)
        ident, value, m = _cell_wire(b, v[:CELL_LEN])
        unique = [b.add_virtual(int(x)) for x in v[CELL_LEN:CELL_LEN + 4]]
        ind_vd, mul_vd = _point(b, v[CELL_LEN + 4:CELL_LEN + 15]), _point(b, v[CELL_LEN + 15:CELL_LEN + 26])
        # SecondaryIndexCellWire::digest (secondary_index_cell.rs:99-139): counters, the row id H(row_unique_data || individual_cnt)
        # and its 128-bit scalar; the scalar multiplication itself is off-circuit (module docstring)
        ind_cnt = b.add(cpis[26], b.not_(m))
        mul_cnt = b.add(cpis[27], m)
        R.hash_to_int_target(b, b.hash_n_to_m_no_pad(unique + [ind_cnt], 4))
        for pis in child_pis:  # multiplier_vd and multiplier_cnt equal the children's (full_node.rs:47-52, partial_node.rs:65-68)
            for x, y in zip(mul_vd, pis[15:26]):
                b.connect(x, y)
            b.connect(mul_cnt, pis[42])
        empty = [b.constant(int(x)) for x in empty_hash]
        cells_h = list(cpis[0:4])
        if kind == "leaf":
            node_min = node_max = value
            h = b.hash_n_to_m_no_pad(empty + empty + value + value + [ident] + value + cells_h, 4)
        elif kind == "full":
            node_min, node_max = list(child_pis[0][26:34]), list(child_pis[1][34:42])
            h = b.hash_n_to_m_no_pad(list(child_pis[0][0:4]) + list(child_pis[1][0:4]) + node_min + node_max + [ident] + value + cells_h, 4)
        else:
            child = child_pis[0]
            is_left = b.add_virtual(int(v[ROW_LEN]))  # add_virtual_bool_target_unsafe: range-checked by the Poseidon gate's swap wire
            node_min = [b.select(is_left, c, x) for c, x in zip(child[26:34], value)]
            node_max = [b.select(is_left, x, c) for c, x in zip(child[34:42], value)]
            h = hash_maybe_first(b, is_left, empty, list(child[0:4]), node_min + node_max + [ident] + value + cells_h)
        return h + ind_vd + mul_vd + list(node_min) + list(node_max) + [mul_cnt]
    return logic


class TableParams:
    """verifiable-db's cells_tree::PublicParameters + row_tree::PublicParameters (cells_tree/api.rs:100-137, row_tree/api.rs:22-63):
    the two circuit sets, each circuit with its wrap chain, built once. prover / fri_params as for recursion.RecursiveCircuits;
    empty_hash = H::hash_no_pad(&[]) by the same hasher."""
    CELL_KINDS = (("cells_leaf", 0, "leaf"), ("cells_full", 2, "full"), ("cells_partial", 1, "partial"), ("cells_empty", 0, "empty"))
    ROW_KINDS = (("row_leaf", 0, "leaf"), ("row_full", 2, "full"), ("row_partial", 1, "partial"))

    def __init__(self, prover, fri_params, empty_hash, pad_base_bits=0, extra_gates=()):
        """pad_base_bits = k > 0: every base circuit of both sets is padded with no-op rows to at least 2^k rows and carries one row
        of every gate of `extra_gates` it lacks (FrameworkCircuit(min_log_n, extra_gates)): the base-degree sweep of SURVEY 8(d). The
        reference's cells / rows circuits hold curve and u256 gadgets on top of the tree logic built here, so their base degrees lie
        in 12..15 (circuit_builder.rs:323-325 is where the real number would come from); k brackets them."""
        self.empty_hash = [int(x) for x in empty_hash]
        self.pad_base_bits = int(pad_base_bits)
        kw = dict(min_log_n=max(6, self.pad_base_bits), extra_gates=tuple(extra_gates))
        self.cells = R.RecursiveCircuits([R.FrameworkCircuit(n, k, cells_logic(kind, self.empty_hash), CELLS_IO, **kw) for n, k, kind in self.CELL_KINDS],
                                         prover, fri_params)
        self.gadget = R.RecursiveCircuitsVerifierGadget(self.cells)
        self.rows = R.RecursiveCircuits([R.FrameworkCircuit(n, k, rows_logic(kind, self.gadget, self.empty_hash), ROWS_IO, **kw) for n, k, kind in self.ROW_KINDS],
                                        prover, fri_params)
        for fw in (self.cells, self.rows):
            for name in fw.circuits:
                fw.witness_programs(name)  # recorded once; read-only and shared by every session from here on

    def shapes(self):
        return {name: [c[0].log_n for c in chain] for fw in (self.cells, self.rows) for name, chain in fw.chains.items()}

    def cells_proof_inputs(self, proof, name):
        """the witness-program inputs of RecursiveCircuitsVerifierGadget::verify_proof_in_circuit_set for a cells-set proof"""
        vd = self.cells.vds[name]
        return R.universal_inputs(proof, vd, self.cells.membership(vd[1]))


# ---- a synthetic table and its off-circuit side ----------------------------------------------------------------------------------------
class SyntheticTable:
    """`rows` rows of one secondary-index column + `n_cols` value columns of uniform U256 values, sorted by the
    secondary index (row i of the arrays = in-order position i of the row tree). `block` prefixes the secondary values' most
    significant limb so that the blocks of different ranks do not interleave (rank r holds block 2 r, the separator row that joins
    two blocks holds an odd prefix)."""

    def __init__(self, rows, n_cols=4, seed=0xC0FFEE04, block=0):
        from . import circuits as C
        self.rows, self.n_cols = rows, n_cols
        self.col_ids = C.rand_field(n_cols + 1, seed)  # [secondary, value columns]: the same identifiers on every rank
        rng = np.random.default_rng([seed, block])
        v = rng.integers(0, 1 << 32, size=(rows, n_cols + 1, 8), dtype=np.uint32)
        v[:, 0, 0] = (np.uint32(block) << np.uint32(16)) | (v[:, 0, 0] & np.uint32(0xFFFF))
        order = np.lexsort(tuple(v[:, 0, j] for j in range(7, -1, -1)))
        self.values = np.ascontiguousarray(v[order])

    def secondary_int(self, i):
        return sum(int(x) << (32 * (7 - j)) for j, x in enumerate(self.values[i, 0]))


class TableWitness:
    """everything a table's proofs take from outside the circuits, computed on the GPU in a handful of batched calls: per-cell value
    digests D(id || value) (cells_tree/mod.rs:63-71), their accumulation up each row's cells tree (SplitDigestPoint::accumulate), the
    row's unique data and individual digest row_id * (D(secondary cell) + cells digest) (secondary_index_cell.rs:99-139), the
    accumulation up the row tree (row_tree/full_node.rs:78-82)."""

    def __init__(self, ctx, table, row_spans, variant=0):
        rows, C = table.rows, table.n_cols
        limbs = table.values.astype(np.uint64)
        ids = np.broadcast_to(table.col_ids[None, 1:, None], (rows, C, 1))
        cell_in = np.concatenate([ids, limbs[:, 1:, :]], axis=2).reshape(rows * C, 9)
        cell_w = map_to_curve_batch(ctx, cell_in, variant)
        # cells-tree node k (1..C) of row r accumulates the cells of its in-order span
        spans = [sbbst_span(C, k) for k in range(1, C + 1)]
        rg = np.array([[r * C + lo - 1, r * C + hi] for r in range(rows) for lo, hi in spans], dtype=np.uint32)
        self.cell_digest = curve_sum_ranges(ctx, cell_w, rg)[1].reshape(rows, C, 11)
        self.unique = ctx.hash_no_pad_batch(limbs[:, 0, :], 4, variant)       # row_unique_data = H(the secondary-index value)
        row_w, self.row_own = row_digests(ctx, table.col_ids, table.values, table.values[:, 0:1, :], variant)
        self.row_w = row_w
        keys = sorted(row_spans)
        acc = curve_sum_ranges(ctx, row_w, np.array([row_spans[k] for k in keys], dtype=np.uint32))
        self.row_digest = {k: acc[1][i] for i, k in enumerate(keys)}
        self.root_digest_w = {k: acc[0][i] for i, k in enumerate(keys)}


# ---- the build ------------------------------------------------------------------------------------------------------------------------
class TableBuild:
    """One table (or one rank's block of it): C cells-tree proofs and one row-tree proof per row, scheduled by ryhope's batched work
    plan. An item of the plan is a spun-off subtree of the row tree (updatetree.rs:372-385,479-515): the unit handed to one worker
    (= one GPU stream with its provers and pinned wire matrices), which proves the cells trees of the subtree's rows, then the
    subtree's row nodes bottom-up, level by level in batches, and returns. Workers run concurrently; the plan hands out an item once
    the subtrees below it are done."""

    def __init__(self, params, sessions, batch=32, subtree_size=64, host_threads=0, keep_proofs=True, keep_nodes=(), group_rows=None):
        """keep_proofs: retain every row proof and cells-tree root after the run (what a checker re-proves sampled nodes from: ~130 KB
        per row); off, a proof is dropped as soon as its parent is proved -- the live set is the frontier of the tree, which is
        what a 2^17-row block needs -- except for the row-tree nodes listed in keep_nodes (their row proofs and cells roots stay:
        the nodes a checker samples from a large block, and their children)"""
        self.p, self.sessions, self.batch, self.subtree_size, self.keep_proofs = params, sessions, batch, subtree_size, keep_proofs
        self.keep_nodes = set(keep_nodes)
        # Ready items of one wave are disjoint subtrees; a worker takes SEVERAL of them at a time (about group_rows rows) and proves
        # them as one unit: the cells trees of all their rows in full batches, then their row nodes level by level with the levels
        # of the different subtrees merged. (A subtree of a balanced tree over n rows has what n leaves it -- 40 rows for 20480
        # rows cut at 64 -- and on its own fills a batch of 32 once and leaves 8 over at every tree position and level.)
        self.group_rows = 32 * batch if group_rows is None else max(1, int(group_rows))  # measured: tools/dbg/table_sweep.sh (1 -> 4 -> 16 batches of rows: 735 -> 825 -> 868 proofs/s)
        self.host_threads = host_threads
        self.pool = queue.Queue()
        for s in sessions:
            self.pool.put(s)
        self.n_proofs = 0
        self.seconds_in_units = 0.0
        self.lock = threading.Lock()

    def _batched(self, fw, name, jobs, sess):
        out = []
        for lo in range(0, len(jobs), self.batch):
            out += fw.generate_proofs_batch(name, jobs[lo:lo + self.batch], threads=self.host_threads, session=sess)
        with self.lock:
            self.n_proofs += len(jobs)
            self.last_session = sess  # whose prover holds the most recent final proofs on the device (the root's, after run())
        return out

    def cells_proofs(self, table, wit, rows, sess):
        """the cells trees of `rows` (celltree.rs:54-189): per tree position (children before parents) one batch over the rows.
        Returns {row: (root proof, circuit name)}."""
        C = table.n_cols
        root = sbbst_root(C)
        order = sorted(range(1, C + 1), key=lambda k: ((k & -k).bit_length(), k))  # by layer: children first
        proofs = {}
        for k in order:
            left, right = sbbst_children(C, k)
            kids = [c for c in (left, right) if c is not None]
            name = ("cells_leaf", "cells_partial", "cells_full")[len(kids)]
            jobs = []
            for r in rows:
                flat = _u64cat([table.col_ids[k]], table.values[r, k], [0], wit.cell_digest[r, k - 1], NEUTRAL_FIELDS)
                jobs.append(([proofs[(r, c)][0] for c in kids], [proofs[(r, c)][1] for c in kids], flat))
            for r, pr in zip(rows, self._batched(self.p.cells, name, jobs, sess)):
                proofs[(r, k)] = (pr, name)
            for r in rows:
                for c in kids:
                    del proofs[(r, c)]
        return {r: proofs[(r, root)] for r in rows}

    def row_job(self, table, wit, nodes, k, cells_root, row_proofs):
        """CircuitInput::{leaf, partial, full} of row_tree/api.rs:166-230 for row-tree node k"""
        left, right = nodes[k]
        kids = [c for c in (left, right) if c is not None]
        name = ("row_leaf", "row_partial", "row_full")[len(kids)]
        proof, cname = cells_root
        flat = [[table.col_ids[0]], table.values[k, 0], [0], wit.unique[k], wit.row_digest[k], NEUTRAL_FIELDS]
        if len(kids) == 1:
            flat.append([1 if left is not None else 0])
        inputs = _u64cat(self.p.cells_proof_inputs(proof, cname), *flat)
        return name, ([row_proofs[c][0] for c in kids], [row_proofs[c][1] for c in kids], inputs)

    def prove_item(self, table, wit, nodes, keys, row_proofs):
        """one unit of work: `keys` = the row-tree nodes of one spun-off subtree -- or of several disjoint ones of the same wave (any
        order; levels are counted inside `keys`, so the subtrees' levels merge). Children of bottom nodes that lie outside were
        proved by earlier items (row_proofs)."""
        sess = self.pool.get()
        t_unit = time.perf_counter()
        try:
            getattr(getattr(sess.prover, "ctx", None), "make_current", lambda: None)()  # this worker thread drives the session's GPU
            keyset = set(keys)
            cells = self.cells_proofs(table, wit, sorted(keys), sess)
            if self.keep_proofs:
                self.cells_roots.update(cells)
            else:
                self.cells_roots.update({k: v for k, v in cells.items() if k in self.keep_nodes})
            height = {}

            def h(k):
                if k not in height:
                    height[k] = 1 + max([h(c) for c in nodes[k] if c is not None and c in keyset], default=-1)
                return height[k]

            for lvl in range(max(h(k) for k in keys) + 1):
                by_name = {}
                for k in sorted(keys):
                    if height[k] == lvl:
                        name, job = self.row_job(table, wit, nodes, k, cells[k], row_proofs)
                        by_name.setdefault(name, []).append((k, job))
                for name, kj in by_name.items():
                    for (k, _), pr in zip(kj, self._batched(self.p.rows, name, [j for _, j in kj], sess)):
                        row_proofs[k] = (pr, name)
                if not self.keep_proofs:  # this level's nodes have consumed their children and their cells roots
                    for k in keys:
                        if height[k] == lvl:
                            cells.pop(k, None)
                            for c in nodes[k]:
                                if c is not None and c not in self.keep_nodes:
                                    row_proofs.pop(c, None)
        finally:
            with self.lock:
                self.seconds_in_units += time.perf_counter() - t_unit
            self.pool.put(sess)

    def seconds_in_library(self):
        """seconds the workers' chains spent inside mp2g_chain_run (GIL released) since they were created: against seconds_in_units
        (the workers' busy time) it gives the share of host glue -- job assembly, numpy, the GIL -- in a worker's time"""
        return sum(ch.seconds_in_run for s in self.sessions for ch in getattr(s.prover, "chains", {}).values())

    def run(self, table, wit, root, nodes):
        """drain the batched work plan of the row tree (rowtree.rs:78-337 with into_batched_workplan): returns (root proof, name)"""
        ut = W.UpdateTree.from_map(0, root, nodes)
        plan = ut.into_batched_workplan(self.subtree_size) if self.subtree_size > 1 else ut.into_workplan()
        row_proofs = {}
        self.row_proofs, self.cells_roots = row_proofs, {}  # kept after the run: what a checker re-proves sampled nodes from
        self.wave_log = []
        with ThreadPoolExecutor(max_workers=len(self.sessions)) as ex:
            while True:
                # one wave = every item that is Ready now; its items are disjoint subtrees and run concurrently. The plan is only
                # polled again once the whole wave is done: polled with items outstanding it re-cuts the subtrees around them
                # and hands out nodes a second time (updatetree.rs:479-515 builds an item from whatever is ready at the moment)
                t_wave = time.perf_counter()
                wave = W.drain_wave(plan)
                if not wave:
                    break
                futures, n_before = [], self.n_proofs
                item_keys = []
                for it in wave:
                    if it.subtree is not None:
                        item_keys.append([int(k) for k in it.subtree.nodes()])
                        it.subtree.free()
                    else:
                        item_keys.append([int(it.k)])
                # groups of whole items, about group_rows rows each, but no fewer groups than workers while the wave has the items
                total = sum(len(k) for k in item_keys)
                target = max(1, min(self.group_rows, -(-total // len(self.sessions))))
                group = []
                for keys in item_keys:
                    group += keys
                    if len(group) >= target:
                        futures.append(ex.submit(self.prove_item, table, wit, nodes, group, row_proofs))
                        group = []
                if group:
                    futures.append(ex.submit(self.prove_item, table, wit, nodes, group, row_proofs))
                for f in futures:
                    f.result()  # re-raises a worker's failure (an unsatisfied witness makes prove() refuse, as the reference panics)
                for it in wave:
                    plan.done(it.k)
                self.wave_log.append((len(wave), self.n_proofs - n_before, time.perf_counter() - t_wave))  # (items, proofs, seconds)
        assert plan.completed()
        plan.free()
        return row_proofs[root]


def sample_nodes(nodes, spans, row0=0):
    """the row-tree nodes a checker re-proves after a build: row0, then the node with the widest span of every kind (leaf / partial /
    full) not seen yet. Returns (samples in that order, the set to keep in a lean build = the samples and their children)."""
    seen, samples = set(), []
    for k in [row0] + sorted(nodes, key=lambda k: (-(spans[k][1] - spans[k][0]), k)):
        kind = sum(c is not None for c in nodes[k])
        if kind in seen:
            continue
        seen.add(kind)
        samples.append(k)
    keep = set(samples)
    for k in samples:
        keep.update(c for c in nodes[k] if c is not None)
    return samples, keep


def join_blocks(build, ctx, left, right, sep_block, n_cols=4, seed=0xC0FFEE04, variant=0):
    """the row-tree node above two ranks' blocks: the separator row between them (a one-row table with the odd prefix sep_block)
    with the blocks' root proofs as its children. left / right = (root proof, circuit name, accumulated digest encoding [5]).
    Proves the separator's cells tree and its full node; returns (proof, name, digest encoding) of the joined tree."""
    from . import curve_sum
    table = SyntheticTable(1, n_cols, seed, sep_block)
    wit = TableWitness(ctx, table, {0: (0, 1)}, variant)
    w, wei = curve_sum(ctx, np.stack([wit.row_w[0], np.asarray(left[2], dtype=np.uint64), np.asarray(right[2], dtype=np.uint64)]), weierstrass=True)
    wit.row_digest[0] = wei
    sess = build.pool.get()
    try:
        cells = build.cells_proofs(table, wit, [0], sess)
        name, job = build.row_job(table, wit, {0: ("L", "R")}, 0, cells[0], {"L": left[:2], "R": right[:2]})
        (proof,) = build._batched(build.p.rows, name, [job], sess)
    finally:
        build.pool.put(sess)
    return proof, name, w


def expected_root_public_inputs(ctx, table, wit, root, nodes, spans, variant=0):
    """the row-tree root's public inputs computed OFF-circuit (indexing.py: MerkleCell::aggregate, RowPayload::aggregate batched per
    level; the multiset digest; min / max of the block): what the root proof must expose"""
    from . import indexing as IX
    rows, C = table.rows, table.n_cols
    empty = IX.empty_poseidon_hash(ctx, variant)
    ints = lambda a: [sum(int(x) << (32 * (7 - j)) for j, x in enumerate(v)) for v in a]
    cell_h = {}
    for k in sorted(range(1, C + 1), key=lambda k: ((k & -k).bit_length(), k)):
        left, right = sbbst_children(C, k)
        lh = cell_h[left] if left is not None else np.tile(empty, (rows, 1))
        rh = cell_h[right] if right is not None else np.tile(empty, (rows, 1))
        cell_h[k] = IX.cell_node_hashes(ctx, lh, rh, np.full(rows, table.col_ids[k]), ints(table.values[:, k]), variant)
    cells_root = cell_h[sbbst_root(C)]
    sec = ints(table.values[:, 0])
    height, row_h = {}, {}

    def h(k):
        if k not in height:
            height[k] = 1 + max([h(c) for c in nodes[k] if c is not None], default=-1)
        return height[k]

    for k in nodes:
        h(k)
    for lvl in range(max(height.values()) + 1):
        ks = [k for k in sorted(nodes) if height[k] == lvl]
        lh = np.stack([row_h[nodes[k][0]] if nodes[k][0] is not None else empty for k in ks])
        rh = np.stack([row_h[nodes[k][1]] if nodes[k][1] is not None else empty for k in ks])
        mins, maxs = [sec[spans[k][0]] for k in ks], [sec[spans[k][1] - 1] for k in ks]
        hs = IX.row_node_hashes(ctx, lh, rh, mins, maxs, np.full(len(ks), table.col_ids[0]), [sec[k] for k in ks], cells_root[ks], variant)
        for k, x in zip(ks, hs):
            row_h[k] = x
    lo, hi = spans[root]
    return _u64cat(row_h[root], wit.row_digest[root], NEUTRAL_FIELDS, u256_to_limbs([sec[lo]])[0], u256_to_limbs([sec[hi - 1]])[0], [0])


# ---- the same build with the scheduler in C++ (csrc/forest.hip) -------------------------------------------------------------------------
class NativeTableBuild:
    """TableBuild with the unit loop in the library: the block's nodes (4 cells-tree nodes and 1 row-tree node per row) are registered
    with mp2g_forest once -- circuit, children, and the words of their witness inputs that are not child proofs, assembled here with
    numpy for the whole block at a time --, then every wave of the update plan goes down as units (groups of Ready items) in one
    mp2g_forest_prove call: worker threads, level batching, job assembly, the hand-over of child proofs (device pool, gather kernels)
    and the witness check all run in C++; no proof visits the host except the root and the nodes a checker asked to keep.
    Same proofs as TableBuild, word for word (tests/test_gpu_table.py::test_native_build_equals_the_python_build)."""

    CIRCUITS = ("cells_leaf", "cells_full", "cells_partial", "row_leaf", "row_full", "row_partial")
    row_job = TableBuild.row_job  # CircuitInput of a row node from host proofs: what a checker re-proves kept nodes with

    def __init__(self, params, provers, batch=32, subtree_size=64, group_rows=None, pool_slots=None):
        self.p, self.provers, self.batch, self.subtree_size = params, provers, batch, subtree_size
        self.group_rows = 32 * batch if group_rows is None else max(1, int(group_rows))
        self.n_proofs = 0
        self.wave_log = []
        fws = {n: (params.cells if n.startswith("cells") else params.rows) for n in self.CIRCUITS}
        self.chains = []
        for pv in provers:
            pv.ctx.make_current()
            self.chains.append([pv._chain([c[0] for c in fws[n].chains[n]], fws[n].witness_programs(n), batch) for n in self.CIRCUITS])
        # proof words of a final proof of either set, in a parent's input order (recursion.proof_inputs): public inputs, 3 caps, openings, FRI
        def pw(fw):
            ch = self.chains[0][self.CIRCUITS.index(next(n for n in self.CIRCUITS if fws[n] is fw))]
            fp, prog = ch.fps[-1], ch.programs[-1]
            return int(prog.n_public_inputs) + 3 * int(fp.cap_words) + 2 * int(fp.n_openings) + int(fp.proof_words), fp, int(prog.n_public_inputs)
        (self.pw_cells, self.fp_cells, self.npi_cells), (self.pw_rows, self.fp_rows, self.npi_rows) = pw(params.cells), pw(params.rows)
        mem = lambda fw: 5 * max(0, (fw.set_size - 1).bit_length())  # membership proof: index bits + 4-limb siblings
        self.head = {"cells": 68, "rows": 68}                         # verifier data in front of a child proof: cap (16 x 4) + digest (4)
        self.mem = {"cells": mem(params.cells), "rows": mem(params.rows)}
        # children of every circuit as (set of the child, ...) in input order; the row circuits' last child is the row's cells root
        self.kids = {"cells_leaf": (), "cells_full": ("cells", "cells"), "cells_partial": ("cells",), "row_leaf": ("cells",), "row_full": ("rows", "rows", "cells"),
                     "row_partial": ("rows", "cells")}
        own = {"cells_leaf": CELL_LEN + 22, "cells_full": CELL_LEN + 22, "cells_partial": CELL_LEN + 22, "row_leaf": ROW_LEN, "row_full": ROW_LEN, "row_partial": ROW_LEN + 1}
        self.desc, self.offsets = [], {}
        for i, n in enumerate(self.CIRCUITS):
            at, offs = 4, []
            for k in self.kids[n]:
                at += self.head[k]
                offs.append(at)
                at += (self.pw_cells if k == "cells" else self.pw_rows) + self.mem[k]
            n_in = at + own[n]
            assert n_in == self.chains[0][i].programs[0].n_inputs, f"{n}: layout {n_in} != the witness program's {self.chains[0][i].programs[0].n_inputs} inputs"
            n_const = n_in - sum(self.pw_cells if k == "cells" else self.pw_rows for k in self.kids[n])
            self.desc.append((n_in, offs, n_const))
            self.offsets[n] = offs
        self.pool_slots = pool_slots
        self.forest = None
        self.last_session = None

    # node ids: row-tree node of row k = k; cells-tree node (row k, position c) = (c << 40) | k
    @staticmethod
    def cell_id(k, c):
        return (int(c) << 40) | int(k)

    def _child_block(self, fw, name):
        """verifier data and membership proof around a child proof of circuit `name` of set fw"""
        vd = fw.vds[name]
        bits, sib = fw.membership(vd[1])
        return (_u64cat(vd[0], vd[1]), _u64cat(bits, sib))

    def register(self, table, wit, root, nodes, keep=()):
        """every node of the block, circuit by circuit (numpy over all rows at once)"""
        p, C, rows = self.p, table.n_cols, table.rows
        keep = set(keep)
        set_c, set_r = np.asarray(p.cells.set_digest, dtype=np.uint64), np.asarray(p.rows.set_digest, dtype=np.uint64)
        neutral = np.asarray(NEUTRAL_FIELDS, dtype=np.uint64)
        ks = np.arange(rows, dtype=np.uint64)
        cell_kind = {}
        for c in range(1, C + 1):
            kids = [x for x in sbbst_children(C, c) if x is not None]
            cell_kind[c] = (("cells_leaf", "cells_partial", "cells_full")[len(kids)], kids)
        blocks_c = {n: self._child_block(p.cells, n) for n in ("cells_leaf", "cells_full", "cells_partial")}
        blocks_r = {n: self._child_block(p.rows, n) for n in ("row_leaf", "row_full", "row_partial")}
        tile = lambda v: np.broadcast_to(np.asarray(v, dtype=np.uint64)[None, :], (rows, len(v)))
        keep_cells = np.array([1 if k in keep else 0 for k in range(rows)], dtype=np.uint8)
        for c in range(1, C + 1):
            name, kids = cell_kind[c]
            flat = np.concatenate([np.full((rows, 1), table.col_ids[c], dtype=np.uint64), table.values[:, c].astype(np.uint64), np.zeros((rows, 1), dtype=np.uint64),
                                   wit.cell_digest[:, c - 1], tile(neutral)], axis=1)
            parts = [tile(set_c)]
            for kc in kids:
                head, tail = blocks_c[cell_kind[kc][0]]
                parts += [tile(head), tile(tail)]
            consts = np.ascontiguousarray(np.concatenate(parts + [flat], axis=1))
            child_ids = np.stack([(np.uint64(kc) << np.uint64(40)) | ks for kc in kids], axis=1) if kids else None
            is_root = c == sbbst_root(C)
            self.forest.add_nodes(self.CIRCUITS.index(name), (np.uint64(c) << np.uint64(40)) | ks, child_ids, consts, keep_cells if is_root else None)
        root_c = sbbst_root(C)
        root_name = cell_kind[root_c][0]
        head_c, tail_c = blocks_c[root_name]
        self.cells_root_name = root_name
        # row nodes by kind; a child's verifier data and membership proof depend on the CHILD's circuit
        self.row_name = {k: ("row_leaf", "row_partial", "row_full")[(l is not None) + (r is not None)] for k, (l, r) in nodes.items()}
        for n in ("row_leaf", "row_partial", "row_full"):
            ka = np.array(sorted(k for k, nm in self.row_name.items() if nm == n), dtype=np.int64)
            m = len(ka)
            if not m:
                continue
            cols, child_cols = [np.broadcast_to(set_r[None, :], (m, 4))], []
            for side in ((), (None,), (0, 1))[("row_leaf", "row_partial", "row_full").index(n)]:
                kid = np.array([(nodes[int(k)][side] if side is not None else next(x for x in nodes[int(k)] if x is not None)) for k in ka], dtype=np.int64)
                child_cols.append(kid.astype(np.uint64))
                cols.append(np.stack([blocks_r[self.row_name[int(x)]][0] for x in kid]))
                cols.append(np.stack([blocks_r[self.row_name[int(x)]][1] for x in kid]))
            cols += [np.broadcast_to(head_c[None, :], (m, head_c.size)), np.broadcast_to(tail_c[None, :], (m, tail_c.size))]
            cols += [np.full((m, 1), table.col_ids[0], dtype=np.uint64), table.values[ka, 0].astype(np.uint64), np.zeros((m, 1), dtype=np.uint64),
                     np.asarray(wit.unique[ka], dtype=np.uint64), np.stack([np.asarray(wit.row_digest[int(k)], dtype=np.uint64) for k in ka]),
                     np.broadcast_to(neutral[None, :], (m, 11))]
            if n == "row_partial":
                cols.append(np.array([[1 if nodes[int(k)][0] is not None else 0] for k in ka], dtype=np.uint64))
            consts = np.ascontiguousarray(np.concatenate(cols, axis=1))
            child_ids = np.stack(child_cols + [(np.uint64(root_c) << np.uint64(40)) | ka.astype(np.uint64)], axis=1)
            kp = np.array([1 if int(k) in keep else 0 for k in ka], dtype=np.uint8)
            self.forest.add_nodes(self.CIRCUITS.index(n), ka.astype(np.uint64), child_ids, consts, kp)

    def _host_proof(self, words, rows_set, name):
        """(caps, openings, FRI words, public inputs) of a pooled proof, as the prover hands them out: the cap of oracle 0 (verifier
        data, not part of a proof's words) is the circuit's constants_sigmas cap"""
        fp, n_pi = (self.fp_rows, self.npi_rows) if rows_set else (self.fp_cells, self.npi_cells)
        cw, no = int(fp.cap_words), int(fp.n_openings)
        caps = np.zeros((4, cw), dtype=np.uint64)
        caps[0] = np.asarray((self.p.rows if rows_set else self.p.cells).vds[name][0], dtype=np.uint64).ravel()
        caps[1:4] = words[n_pi:n_pi + 3 * cw].reshape(3, cw)
        at = n_pi + 3 * cw
        return caps, words[at:at + 2 * no].reshape(no, 2).copy(), words[at + 2 * no:].copy(), words[:n_pi].copy()

    def run(self, table, wit, root, nodes, keep=()):
        """register the block, then drain the batched work plan wave by wave, every wave as one mp2g_forest_prove over groups of its
        items; returns (root proof, name). Nodes in `keep` (and their children and cells roots) stay downloadable afterwards:
        self.row_proofs / self.cells_roots hold them as host tuples, like TableBuild's."""
        from . import Forest
        C = table.n_cols
        keep_rows = set(keep)
        for k in list(keep_rows):
            keep_rows.update(c for c in nodes[k] if c is not None)
        keep_rows.add(root)
        if self.forest is not None:
            self.forest.free()
        # the pool holds the frontier: per worker a unit's cells leaves (2 per row) while its full nodes are proved, plus the roots of
        # the items of earlier waves and what the caller keeps
        # (per row of a unit: the widest live set of its cells tree, level by level -- 3 proofs for 4 value columns, 9 for 20 --, and
        # never less than the cells root + the row node beside it)
        unit = max(1, min(self.group_rows, -(-table.rows // len(self.provers))))
        per_row = max(2, cells_tree_live_peak(C))
        slots = self.pool_slots or (len(self.provers) * (per_row * unit + 4 * self.batch) + table.rows // max(1, self.subtree_size // 2) + 2 * len(keep_rows) + 256)
        self.forest = Forest([pv.ctx for pv in self.provers], self.desc, self.chains, max(self.pw_cells, self.pw_rows), slots)
        self.register(table, wit, root, nodes, keep_rows)
        ut = W.UpdateTree.from_map(0, root, nodes)
        plan = ut.into_batched_workplan(self.subtree_size) if self.subtree_size > 1 else ut.into_workplan()
        # the harness loop (drain the Ready items, prove, mark done) runs inside the library: a row key k stands for the row node k and
        # its C cells-tree nodes (c << 40) | k
        t0 = time.perf_counter()
        stop = self._progress_writer(t0, (C + 1) * table.rows)
        try:
            items = self.forest.prove_plan(plan, self.group_rows, n_satellites=C, satellite_shift=40)
        finally:
            stop()
        self.wave_log = [(n_items, None, None) for n_items in items]
        self.seconds_in_prove = time.perf_counter() - t0
        assert plan.completed()
        plan.free()
        self.n_proofs += self.forest.proved
        self.row_proofs = {k: (self._host_proof(self.forest.proof_words(k), True, self.row_name[k]), self.row_name[k]) for k in keep_rows}
        self.cells_roots = {k: (self._host_proof(self.forest.proof_words(self.cell_id(k, sbbst_root(C))), False, self.cells_root_name), self.cells_root_name) for k in keep_rows}
        return self.row_proofs[root]

    def _progress_writer(self, t0, total):
        """MP2G_PROGRESS_FILE=path: a line (seconds, proofs so far, of how many) every 30 s while a long block is proved (the forest's
        counter is an atomic the workers bump per batch), so that a run cut off by a time limit still says how far it came"""
        path = os.environ.get("MP2G_PROGRESS_FILE")
        if not path:
            return lambda: None
        import threading
        done, forest = threading.Event(), self.forest

        def loop():
            while not done.wait(30.0):
                with open(path, "a") as f:
                    f.write(f"{time.perf_counter() - t0:.1f} s  {forest.proved} / {total} proofs\n")
        th = threading.Thread(target=loop, daemon=True)
        th.start()

        def stop():
            done.set()
            th.join()
            with open(path, "a") as f:
                f.write(f"{time.perf_counter() - t0:.1f} s  {forest.proved} / {total} proofs (plan done)\n")
        return stop

    def free(self):
        if self.forest is not None:
            self.forest.free()
            self.forest = None
