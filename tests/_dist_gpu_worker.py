"""Worker for tests/test_gpu_map_reduce.py::test_two_ranks_prove_an_update_tree: two ranks (gloo rendezvous, both on
the box's GPU) drive one work plan (sharding.run_workplan) and PROVE their share of the nodes with real batched
provers; a node's public-input hash is the hash of its children's proof fingerprints, so the root depends on every
proof of the tree. Rank 0 prints the root fingerprint; the test compares it with a single-process run."""
import hashlib
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def tree_paths():
    prng = np.random.default_rng(3)
    parent = {0: None}
    for k in range(1, 14):
        parent[k] = int(prng.integers(0, k))
    kids = {k: sorted(c for c, p in parent.items() if p == k) for k in parent}

    def path(k):
        out = []
        while k is not None:
            out.append(k)
            k = parent[k]
        return out[::-1]
    leaves = [k for k in parent if not kids[k]]
    return [path(k) for k in leaves], kids


def run(dist):
    import circuits as C
    import oracle as O  # rand_field only
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    sh = importlib.import_module("mapreduce-plonky2_amd.sharding")
    wp = importlib.import_module("mapreduce-plonky2_amd.workplan")
    ctx = mp2.Context(0)
    kinds = [k for k in C.VERIFIER_KINDS if k[0] != C.PUBLIC_INPUT]
    ckt = C.build(5, kinds, 9)
    fp = mp2.standard_recursion_params(5, (ckt.num_constants + C.NUM_ROUTED, C.NUM_WIRES, 20, 16), pow_bits=4, num_queries=3)
    pr = mp2.BatchedProver(ctx, fp, 1)
    pr.set_preprocessed(ctx.to_device(ckt.pre))
    pr.enable_permutation(C.NUM_ROUTED, 8)
    pr.enable_quotient()
    pr.set_gates([mp2.Gate(g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end) for g in ckt.gates], ckt.num_selectors)
    pr.enable_witness_check()
    d_w, d_cd, d_ph = ctx.to_device(ckt.wires[None]), ctx.to_device(O.rand_field(4, 1)), ctx.alloc(32)
    paths, kids = tree_paths()
    proved = []

    def prove_node(k, child_fps):
        pi = np.frombuffer(hashlib.sha256(bytes([k]) + b"".join(child_fps)).digest(), dtype=np.uint64) % np.uint64(O.P)
        d_ph.upload(pi.reshape(1, 4))
        pr.prove([d_w, None, None], d_cd, d_ph)
        assert pr.witness_status().tolist() == [0]
        caps, openings, proofs = pr.results()
        proved.append(k)
        return hashlib.sha256(mp2.serialize_proof(fp, ckt.num_constants, caps[0], openings[0], proofs[0], pi)).digest()

    def prove_item(item, done):
        memo = {}
        for k in (item.subtree.bottom_up() if item.subtree is not None else [item.k]):
            memo[k] = prove_node(k, [memo[c] if c in memo else done[c] for c in kids[k]])
        return memo[item.k]

    plan = wp.UpdateTree.from_paths(paths, 1).into_batched_workplan(3)
    res = sh.run_workplan(dist, plan, prove_item)
    n_proved = len(proved)
    if dist is not None:
        n_proved = int(sh.all_gather_words(dist, [n_proved]).sum())
    pr.free()
    ctx.close()
    return res[0].hex(), n_proved, len(kids)


if __name__ == "__main__":
    import torch.distributed as dist
    dist.init_process_group("gloo")
    root, n_proved, n_nodes = run(dist)
    assert n_proved == n_nodes
    if dist.get_rank() == 0:
        print(f"root={root} proved={n_proved}")
    dist.barrier()
    dist.destroy_process_group()
