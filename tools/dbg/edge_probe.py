import importlib, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle as O
mp2 = importlib.import_module("mapreduce-plonky2_amd")
ctx = mp2.Context(0)
def attempt(name, f):
    try:
        r = f()
        print(name, "->", r if not isinstance(r, np.ndarray) or r.size < 24 else r.shape)
    except Exception as e:
        print(name, "raised", type(e).__name__, str(e)[:100])
a1 = O.rand_field((3, 1), 1)
attempt("ntt n=1", lambda: ctx.ntt(a1))
attempt("ntt n=1 == input", lambda: bool(np.array_equal(ctx.ntt(a1), a1)))
attempt("ntt inverse n=1", lambda: bool(np.array_equal(ctx.ntt(a1, inverse=True), a1)))
attempt("ntt batch 0", lambda: ctx.ntt(np.zeros((0, 8), dtype=np.uint64)))
attempt("hash count 0", lambda: ctx.hash_no_pad_batch(np.zeros((0, 5), dtype=np.uint64)))
attempt("hash in_len 0", lambda: ctx.hash_no_pad_batch(np.zeros((2, 0), dtype=np.uint64)))
attempt("lde rate 0", lambda: bool(np.array_equal(ctx.lde_leaves(O.rand_field((2, 8), 3), 0), O.lde_leaves(O.rand_field((2, 8), 3), 0))))
attempt("lde n=1", lambda: ctx.lde_leaves(O.rand_field((2, 1), 3), 3))
attempt("curve_sum empty", lambda: mp2.curve_sum(ctx, np.zeros((0, 5), dtype=np.uint64)))
attempt("map_to_curve count 0", lambda: mp2.map_to_curve_batch(ctx, np.zeros((0, 9), dtype=np.uint64)))
attempt("scalar_mul count 0", lambda: mp2.scalar_mul_batch(ctx, np.zeros((0, 5), dtype=np.uint64), []))
attempt("row digest 0 rows", lambda: mp2.compute_table_row_digest(ctx, np.array([1, 2], dtype=np.uint64), np.zeros((0, 2, 8), dtype=np.uint32), np.zeros((0, 1, 8), dtype=np.uint32)))
attempt("merkle 1 leaf cap 0", lambda: mp2.MerkleTree(ctx, O.rand_field((1, 7), 4), 0).cap)
attempt("merkle 1 leaf cap 0 vs oracle", lambda: bool(np.array_equal(mp2.MerkleTree(ctx, O.rand_field((1, 7), 4), 0).cap, O.merkle_cap(O.merkle_build(O.rand_field((1, 7), 4), 0), 0))))
attempt("merkle cap==log", lambda: bool(np.array_equal(mp2.MerkleTree(ctx, O.rand_field((4, 7), 4), 2).cap, O.merkle_cap(O.merkle_build(O.rand_field((4, 7), 4), 2), 2))))
attempt("merkle prove none", lambda: mp2.MerkleTree(ctx, O.rand_field((4, 7), 4), 1).prove([]))
attempt("commit n=1", lambda: mp2.PolynomialBatch.from_values(ctx, O.rand_field((3, 1), 1), 3, 0).cap)
