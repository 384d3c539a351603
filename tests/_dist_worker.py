"""Worker for tests/test_distributed_gloo.py: world_size ranks over gloo on CPU. The local compute
step is played by the CPU oracle (tests may use it); what is under test is the product's sharding /
collective plumbing in mapreduce-plonky2_amd/sharding.py, which bench.py uses unchanged over RCCL."""
import hashlib
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    import oracle as O
    sh = importlib.import_module("mapreduce-plonky2_amd.sharding")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    # ---- multiset digest: rows sharded, one all_gather of one point per rank, local sum
    rows, n_cols, n_unique = 24, 3, 1
    rng = np.random.default_rng(1234)
    col_ids = O.rand_field(n_cols, 0xC0FFEE04)
    values = rng.integers(0, 1 << 32, size=(rows, n_cols, 8), dtype=np.uint32)
    unique = rng.integers(0, 1 << 32, size=(rows, n_unique, 8), dtype=np.uint32)

    def digest(lo, hi):
        w = np.zeros(5, dtype=np.uint64)
        O.lib().orc_row_digest_batch(0, O.p(col_ids), O.sz(n_cols), O.p(O.arr(values[lo:hi], np.uint32)),
                                     O.p(O.arr(unique[lo:hi], np.uint32)), O.sz(n_unique), O.sz(hi - lo), O.p(w), None)
        return w

    lo, hi = sh.shard_range(rows, rank, world)
    gathered = sh.all_gather_words(dist, digest(lo, hi))
    assert gathered.shape == (world, 5)
    total = np.zeros(5, dtype=np.uint64)
    assert O.lib().orc_curve_sum(O.p(O.arr(gathered)), O.sz(world), O.p(total), None)
    assert np.array_equal(total, digest(0, rows)), "sharded digest != whole-table digest"

    # ---- aggregation tree hand-off: 8 leaves, binary, payload = fake serialized proofs
    n_leaves = 8
    store = {}
    llo, lhi = sh.shard_range(n_leaves, rank, world)
    for i in range(llo, lhi):
        store[(0, i)] = bytes([i]) * (100 + i)
    plan = sh.subtree_plan(n_leaves, 2, world)
    moves = sh.tree_handoff_plan(n_leaves, 2, world)
    for lvl_idx, lvl in enumerate(plan):
        for (l, src, dst, child) in [m for m in moves if m[0] == lvl_idx]:
            got = sh.exchange_bytes(dist, store.get((lvl_idx, child), b""), src, dst)
            if rank == dst:
                store[(lvl_idx, child)] = got
        for node, owner, children in lvl:
            if owner == rank:
                parts = [store[(lvl_idx, c)] for c in children]  # children must be local by now
                store[(lvl_idx + 1, node)] = b"".join(parts)
    if rank == 0:
        root = store[(len(plan), 0)]
        want = b"".join(bytes([i]) * (100 + i) for i in range(n_leaves))
        assert root == want, "aggregation order broken"
    # ---- the proof hand-off of bench.py --workload tree: fixed-size word tensors, pairwise along the levels above the
    # shard boundary (rank r with bit l set sends its subtree root to r - 2^l); payload = the sender's rank pattern
    import torch
    sizes = [7, 11, 5, 9]
    mine = [torch.full((n,), 1000 * rank + i, dtype=torch.int64) for i, n in enumerate(sizes)]
    acc = [rank]
    for lvl in range(world.bit_length() - 1):
        bit = 1 << lvl
        if rank & (bit - 1):
            break
        if rank & bit:
            sh.send_proof_words(dist, mine + [torch.tensor(acc + [-1] * (world - len(acc)), dtype=torch.int64)], rank - bit)
            break
        got = sh.recv_proof_words(dist, sizes + [world], rank + bit)
        assert all(int(t[0]) == 1000 * (rank + bit) + i for i, t in enumerate(got[:4]))
        acc += [int(x) for x in got[4] if int(x) >= 0]
    if rank == 0:
        assert sorted(acc) == list(range(world)), "every rank's subtree reaches the root exactly once"
    # ---- the pre-timing hand-off probe of bench.py (sharding.probe_handoff): every join pair moves a small buffer the way its root
    # proof will move; all pairs direct -> "device"; ONE sender whose direct tensor cannot be made -> its peer's pending recv is
    # answered by the staged tensor and EVERY rank chooses "staged"; no direct path at all (gloo in bench.py) -> "host"
    pairs = sh.join_pairs(world)
    assert len(pairs) == world - 1 and sorted(src for _, src, _ in pairs) == list(range(1, world)), "every rank but 0 hands its tree over exactly once"
    assert all(src - dst == 1 << lvl and dst % (2 << lvl) == 0 for lvl, src, dst in pairs)
    host = lambda w: torch.from_numpy(np.ascontiguousarray(w, dtype=np.uint64).view(np.int64).copy())
    assert sh.probe_handoff(dist, host, host)["mode"] == "device" and sh.HANDOFF["mode"] == "device"
    failing = world - 1  # a sender at level 0
    def direct(w):
        if rank == failing:
            raise RuntimeError("no tensor view of this allocation")
        return host(w)
    got = sh.probe_handoff(dist, direct, host)
    assert got["mode"] == "staged" and got["reason"], got
    if rank in (failing, failing - 1):
        assert f"rank {failing} -> {failing - 1}" in got["reason"], got
    def corrupt(w):  # a direct path that delivers other words than it was given
        w = w.copy()
        if rank == 1:
            w[5] ^= np.uint64(1)
        return host(w)
    assert sh.probe_handoff(dist, corrupt, host)["mode"] == "staged"
    assert sh.probe_handoff(dist, None, host)["mode"] == "host"
    # ---- work-plan driven proving: same UpdateTree on every rank, subtrees dealt per wave, root results
    # published by one all_gather per wave; the root must equal the sequential bottom-up result
    import hashlib
    wp = importlib.import_module("mapreduce-plonky2_amd.workplan")
    prng = np.random.default_rng(7)
    parent = {0: None}
    for k in range(1, 60):
        parent[k] = int(prng.integers(0, k))
    def path(k):
        out = []
        while k is not None:
            out.append(k)
            k = parent[k]
        return out[::-1]
    leaves = [k for k in parent if k not in parent.values()]
    kids = {k: sorted(c for c, p in parent.items() if p == k) for k in parent}
    def seq(k):
        return hashlib.sha256(bytes([k]) + b"".join(seq(c) for c in kids[k])).digest()
    for batch in (1, 5):
        tree = wp.UpdateTree.from_paths([path(k) for k in leaves], 1)
        plan = tree.into_workplan() if batch == 1 else tree.into_batched_workplan(batch)
        proved_here = []
        def prove_item(item, done):
            # a Node item is one proof; a Subtree item is proved bottom-up locally. Children outside the
            # item were finished in earlier waves and arrive through `done`.
            memo = {}
            keys = item.subtree.bottom_up() if item.subtree is not None else [item.k]
            for k in keys:
                memo[k] = hashlib.sha256(bytes([k]) + b"".join(memo[c] if c in memo else done[c] for c in kids[k])).digest()
                proved_here.append(k)
            return memo[item.k]
        res = sh.run_workplan(dist, plan, prove_item)
        assert res[0] == seq(0), "work-plan root differs from the sequential result"
        counts = sh.all_gather_words(dist, [len(proved_here)])
        assert int(counts.sum()) == len(parent), "every node is proved exactly once across the ranks"
        assert world == 1 or int(counts.min()) > 0, "no rank may sit idle on a 60-node tree"
    # ---- the proof store shared by the ranks of a host (mapreduce-plonky2_amd/proofstore.py; mp2-v1/tests/common/proof_storage.rs): every
    # rank keeps its block root under its own ProofKey and re-writes one key all ranks share (the latest proof under a key counts, and
    # a reader never sees half a proof: files are renamed into place); rank 0 then takes every rank's proof out, as the call that
    # joins the blocks of `bench.py --resume-dir` does
    import tempfile
    PS = importlib.import_module("mapreduce-plonky2_amd.proofstore")
    path = [tempfile.mkdtemp(prefix="mp2g_store_") if rank == 0 else None]
    dist.broadcast_object_list(path, src=0)
    store = PS.ProofStore(path[0])
    blob = lambda r: hashlib.sha256(bytes([r])).digest() * 4096  # 128 KB: the size of a ProofWithVK blob
    store.store_proof(PS.ProofKey.row("gloo_table", 1, f"{rank:064x}"), blob(rank), {"block": rank})
    shared = PS.ProofKey.row("gloo_table", 1, "shared")
    for _ in range(8):
        store.store_proof(shared, blob(rank))
        got = store.get_proof_exact(shared)
        assert len(got) == 32 * 4096 and got[:32] * 4096 == got, "a reader saw a torn proof"
    dist.barrier()
    if rank == 0:
        for r in range(world):
            assert store.get_proof_exact(PS.ProofKey.row("gloo_table", 1, f"{r:064x}")) == blob(r) and store.note(PS.ProofKey.row("gloo_table", 1, f"{r:064x}")) == {"block": r}
        assert len(store.keys()) == world + 1
        import shutil
        shutil.rmtree(path[0])
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
