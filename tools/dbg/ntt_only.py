import importlib, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
mp2 = importlib.import_module("mapreduce-plonky2_amd")
import oracle as O
ctx = mp2.Context(0)
for log_n, batch in ((22, 1), (22, 4), (22, 16), (15, 135), (15, 1024), (12, 135 * 8), (12, 16384)):
    n = 1 << log_n
    d_in = ctx.to_device(O.rand_field((batch, n), 1))
    d_out = ctx.alloc(batch * n * 8)
    for br in (True, False):
        ctx.ntt_dev(d_in, d_out, log_n, batch, bitrev_out=br)
        ms = []
        for _ in range(10):
            ctx.timer_start(); ctx.ntt_dev(d_in, d_out, log_n, batch, bitrev_out=br); ms.append(ctx.timer_stop())
        t = float(np.median(ms))
        print(f"ntt log_n={log_n} batch={batch} bitrev={br}: {t*1e3:.1f} us  {16*n*batch/t/1e6:.0f} GB/s", flush=True)
# LDE 135 x 2^15 -> x8
d_c = ctx.to_device(O.rand_field((135, 1 << 15), 2)); d_v = ctx.alloc(135 * (1 << 18) * 8)
ctx.lde_dev(d_c, 15, 135, 3, d_v)
ms = []
for _ in range(10):
    ctx.timer_start(); ctx.lde_dev(d_c, 15, 135, 3, d_v); ms.append(ctx.timer_stop())
t = float(np.median(ms)); print(f"lde 135x2^15 x8: {t*1e3:.1f} us  {72*(1<<15)*135/t/1e6:.0f} GB/s (72 n w)")
ctx.close()
