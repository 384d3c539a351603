"""PolynomialBatch::from_values of 135 x 2^17 resident values, 12 times: the process a `rocprofv3 --pmc` pass of the leaf sponge runs
(2^20 leaves x 17 permutations per launch of leaf_hash_poly_major_kernel<0>; tools/dbg/profile_r05.sh)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mp2 = importlib.import_module("mapreduce-plonky2_amd")
C = importlib.import_module("mapreduce-plonky2_amd.circuits")
ctx = mp2.Context(0)
lg, w = 17, 135
d_v = ctx.to_device(C.rand_field((w, 1 << lg), 0xC0FFEE02))
pb = mp2.PolynomialBatch.from_values_dev(ctx, d_v, lg, w, 3, 4)
for _ in range(12):
    ctx.timer_start()
    pb.recommit_from_values_dev(d_v)
    print(f"commit 135 x 2^{lg}: {ctx.timer_stop():.3f} ms")
