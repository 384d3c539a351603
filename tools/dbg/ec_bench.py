import importlib, sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
mp2 = importlib.import_module("mapreduce-plonky2_amd")
import oracle as O
ctx = mp2.Context(0)
L = mp2.load()
rows, n_cols, n_unique = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 4, 1
rng = np.random.default_rng(1)
d_ids = ctx.to_device(O.rand_field(n_cols, 3))
d_vals = ctx.to_device(rng.integers(0, 1 << 32, size=(rows, n_cols, 8), dtype=np.uint32))
d_unq = ctx.to_device(rng.integers(0, 1 << 32, size=(rows, n_unique, 8), dtype=np.uint32))
d_frac = ctx.alloc(20 * 8)
w = np.zeros(5, dtype=np.uint64)
for it in range(3):
    ctx.sync(); t = time.perf_counter()
    mp2._ck(L.mp2g_row_digest_batch_dev(ctx.h, 0, d_ids.ptr, n_cols, d_vals.ptr, d_unq.ptr, n_unique, rows, d_frac.ptr, mp2._p(w), None))
    dt = time.perf_counter() - t
    print(f"row_digest {rows} rows x {n_cols} cols: {dt*1e3:.1f} ms  {rows/dt:.0f} rows/s  {rows*n_cols/dt:.0f} map_to_curve/s")
# map_to_curve alone
ins = O.rand_field((1 << 16, 9), 5)
t = time.perf_counter(); mp2.map_to_curve_batch(ctx, ins); print("map_to_curve_batch 65536 (host ptrs):", time.perf_counter() - t)
ctx.close()
