"""A/B leg of tools/dbg/merkle_fused_ab.sh: MerkleTree::new over 135 x 2^15 resident values (2^18 leaves, cap 4) -- the leaf sponges
and the tree levels by themselves (mp2g_batch_rehash_dev) -- and the whole commitment; medians of 9 launches between HIP events"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mp2 = importlib.import_module("mapreduce-plonky2_amd")
C = importlib.import_module("mapreduce-plonky2_amd.circuits")
ctx = mp2.Context(0)
for lg, w in ((15, 135), (12, 135), (12, 20)):
    d_v = ctx.to_device(C.rand_field((w, 1 << lg), 0xC0FFEE02))
    pb = mp2.PolynomialBatch.from_values_dev(ctx, d_v, lg, w, 3, 4)
    def med(fn, runs=9):
        fn()
        ms = []
        for _ in range(runs):
            ctx.timer_start(); fn(); ms.append(ctx.timer_stop())
        return float(np.median(ms))
    commit, leaf, levels = med(lambda: pb.recommit_from_values_dev(d_v)), med(lambda: pb.rehash_dev(1)), med(lambda: pb.rehash_dev(2))
    L = 1 << (lg + 3)
    perms = L * ((w + 7) // 8) + L - 16
    print(f"{w} x 2^{lg}: commit {commit:.3f} ms, leaf sponges {leaf:.3f} ms, tree levels ({lg + 3 - 4} levels) {levels:.3f} ms; "
          f"(leaf + levels) {perms / ((leaf + levels) / 1e3) / 1e9:.3f} G permutations/s; cap {pb.cap[0][:2]}")
    pb.free(); d_v.free()
