import importlib, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
mp2 = importlib.import_module("mapreduce-plonky2_amd")
import oracle as O
ctx = mp2.Context(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W = (84, 135, 20, 16)
fp = mp2.standard_recursion_params(k, W)
n = 1 << k
pr = mp2.BatchedProver(ctx, fp, B)
pre = ctx.to_device(O.rand_field((W[0], n), 1))
pr.set_preprocessed(pre)
d_vals = [ctx.to_device(O.rand_field((B, w, n), 2 + i)) for i, w in enumerate(W[1:])]
d_cd = ctx.to_device(O.rand_field(4, 7)); d_ph = ctx.to_device(O.rand_field((B, 4), 8))
for it in range(2):
    ctx.sync(); t = time.perf_counter()
    pr.prove(d_vals, d_cd, d_ph); ctx.sync()
    print("prove", it, time.perf_counter() - t, flush=True)
ctx.close()
