"""2^22 NTT under the tuning knobs of ntt.hip (MP2G_NTT_LW11, MP2G_NTT_N1); prints us and GB/s per setting."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    C = importlib.import_module("mapreduce-plonky2_amd.circuits")
    ctx = mp2.Context(0)
    for batch in (1, 8):
        n = 1 << 22
        d_in = ctx.to_device(C.rand_field((batch, n), 1)); d_out = ctx.alloc(batch * n * 8)
        ctx.ntt_dev(d_in, d_out, 22, batch, bitrev_out=True)
        ms = []
        for _ in range(20):
            ctx.timer_start(); ctx.ntt_dev(d_in, d_out, 22, batch, bitrev_out=True); ms.append(ctx.timer_stop())
        t = float(np.median(ms))
        print(f"  batch {batch}: {t*1e3/batch:.1f} us per transform, {16*n*batch/t/1e6:.0f} GB/s", flush=True)
    ctx.close()
else:
    import ast
    envs = ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else ({}, {"MP2G_NTT_LW11": "1"}, {"MP2G_NTT_LW11": "3"}, {"MP2G_NTT_N1": "10"}, {"MP2G_NTT_N1": "12"}, {"MP2G_NTT_N1": "10", "MP2G_NTT_LW11": "1"})
    for env in envs:
        print(env or "default", flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env))
