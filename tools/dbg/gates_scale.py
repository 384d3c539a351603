"""Time one gate-level prove() (debug aid): python tools/dbg/gates_scale.py K B [kind ...]; no args = sweep in subprocesses"""
import faulthandler, importlib, os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) == 1:
    for cfg in (["8", "2"], ["9", "1"], ["9", "2", "3"], ["9", "2", "7"], ["9", "2", "11"], ["9", "2", "1", "2", "3", "4"], ["10", "1"]):
        r = subprocess.run([sys.executable, __file__] + cfg, capture_output=True, text=True, timeout=120)
        print(cfg, r.returncode, r.stdout.strip().replace("\n", " ; ")[-300:], r.stderr.strip()[:120].replace("\n", " | "), flush=True)
    sys.exit(0)
faulthandler.dump_traceback_later(50, exit=True)
import circuits as C
import oracle as O
mp2 = importlib.import_module("mapreduce-plonky2_amd")
ctx = mp2.Context(0)
k, B = int(sys.argv[1]), int(sys.argv[2])
kinds = [kk for kk in C.ALL_KINDS if not sys.argv[3:] or kk[0] == 0 or str(kk[0]) in sys.argv[3:]]
t = time.time(); ckt = C.build(k, kinds, 5); print("build", k, round(time.time() - t, 2), flush=True)
fp = mp2.standard_recursion_params(k, (ckt.num_constants + 80, 135, 20, 16))
gates = [mp2.Gate(g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end) for g in ckt.gates]
d_w = ctx.to_device(np.stack([ckt.wires] * B)); d_cd = ctx.to_device(O.rand_field(4, 1)); d_ph = ctx.to_device(np.stack([ckt.pi_hash] * B))
order = (True, False) if os.environ.get("GATES_FIRST") else (False, True)
if os.environ.get("STD_TEST_PARAMS"):
    import ctypes
    ofp = O.standard_params(k, (ckt.num_constants + 80, 135, 20, 16), pow_bits=4, num_queries=3)
    ctypes.memmove(ctypes.byref(fp), ctypes.byref(ofp), ctypes.sizeof(fp))
pr = mp2.BatchedProver(ctx, fp, B)
pr.set_preprocessed(ctx.to_device(ckt.pre))
pr.enable_permutation(80, 8); pr.enable_quotient()
for with_gates in order:
    pr.set_gates(gates if with_gates else [], ckt.num_selectors)
    for it in range(2):
        t = time.time(); pr.prove([d_w, None, None], d_cd, d_ph); ctx.sync(); print("gates" if with_gates else "perm", round(time.time() - t, 4), flush=True)
pr.free()
