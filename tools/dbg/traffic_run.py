import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
mp2 = importlib.import_module("mapreduce-plonky2_amd")
import oracle as O
ctx = mp2.Context(0)
n = 1 << 22
d_in = ctx.to_device(O.rand_field((1, n), 1)); d_out = ctx.alloc(n * 8)
# calibration: inverse coset NTT ends with scale_powers_kernel over the whole 32 MiB buffer (8 B/lane r+w)
for _ in range(6):
    ctx.ntt_dev(d_in, d_out, 22, 1, inverse=True, coset_shift=mp2.MULT_GEN)
# measured: forward 2^22 NTT, bit-reversed output (the bench's roofline leg)
for _ in range(10):
    ctx.ntt_dev(d_in, d_out, 22, 1, bitrev_out=True)
ctx.close()
