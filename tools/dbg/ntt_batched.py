"""batched NTT shapes of the prover (8192 x 2^12, 4096 x 2^13 with 8 cosets = LDE) per library given in MP2G_LIB"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
mp2 = importlib.import_module("mapreduce-plonky2_amd")
ctx = mp2.Context(0)
for log_n, nb in ((12, 8192), (13, 4096), (10, 32768)):
    d = ctx.alloc(nb * (1 << log_n) * 8)
    ctx.ntt_dev(d, d, log_n, nb, bitrev_out=True)
    ms = []
    for _ in range(10):
        ctx.timer_start(); ctx.ntt_dev(d, d, log_n, nb, bitrev_out=True); ms.append(ctx.timer_stop())
    t = float(np.median(ms))
    print(f"  {nb} x 2^{log_n}: {t*1e3:.1f} us  {16.0 * nb * (1 << log_n) / t / 1e6:.0f} GB/s")
    d.free()
ctx.close()
