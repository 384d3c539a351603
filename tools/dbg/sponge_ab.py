"""the two sponge kernels on the same amount of work: hash_no_pad_batch (leaf-major input) and leaf_hash_poly_major (inside
commit_from_values_dev: 135 polynomials of 2^17 points -> 2^20 leaves of 135 limbs); run under rocprofv3 --kernel-trace --stats"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
mp2 = importlib.import_module("mapreduce-plonky2_amd")
ctx = mp2.Context(0)
rng = np.random.default_rng(1)
n_hash, limbs = 1 << 20, 136
for fill in ("zeros", "random"):
    a = np.zeros((n_hash, limbs), dtype=np.uint64) if fill == "zeros" else rng.integers(0, 0xFFFFFFFF00000000, size=(n_hash, limbs), dtype=np.uint64)
    d_in, d_out = ctx.to_device(a), ctx.alloc(n_hash * 32)
    for _ in range(3):
        ctx.timer_start()
        mp2._ck(mp2.load().mp2g_hash_no_pad_batch_dev(ctx.h, 0, d_in.ptr, limbs, n_hash, 4, d_out.ptr))
        ms = ctx.timer_stop()
    print(f"hash_no_pad_batch {fill}: {ms:.3f} ms = {n_hash * 17 / ms / 1e6:.3f} G perm/s")
    d_in.free(); d_out.free()
vals = rng.integers(0, 0xFFFFFFFF00000000, size=(135, 1 << 17), dtype=np.uint64)
d_v = ctx.to_device(vals)
for _ in range(3):
    ctx.timer_start()
    b = mp2.PolynomialBatch.from_values_dev(ctx, d_v, 17, 135, 3, 4)
    ms = ctx.timer_stop()
    b.free()
print(f"commit_from_values_dev 135 x 2^17 (iNTT + LDE + 2^20 leaves x 17 perms + levels): {ms:.3f} ms")
ctx.close()
