"""Single-proof latency with and without hipGraph replay (debug aid): python tools/dbg/graph_latency.py K B"""
import faulthandler, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
faulthandler.dump_traceback_later(80, exit=True)
import circuits as C
import oracle as O
mp2 = importlib.import_module("mapreduce-plonky2_amd")
ctx = mp2.Context(0)
k, B = int(sys.argv[1]), int(sys.argv[2])
ckt = C.build(k, C.ALL_KINDS, 5)
fp = mp2.standard_recursion_params(k, (ckt.num_constants + 80, 135, 20, 16))
pr = mp2.BatchedProver(ctx, fp, B)
pr.set_preprocessed(ctx.to_device(ckt.pre))
pr.enable_permutation(80, 8); pr.enable_quotient()
pr.set_gates([mp2.Gate(g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end) for g in ckt.gates], ckt.num_selectors)
d_w = ctx.to_device(np.stack([ckt.wires] * B)); d_cd = ctx.to_device(O.rand_field(4, 1)); d_ph = ctx.to_device(np.stack([ckt.pi_hash] * B))
def run(n):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); pr.prove([d_w, None, None], d_cd, d_ph); ctx.sync(); ts.append(time.perf_counter() - t)
    return ts
ts = run(6); ref = pr.results()
print("plain ms", [round(x * 1e3, 2) for x in ts], flush=True)
pr.enable_graph(True)
ts = run(8); got = pr.results()
print("graph ms", [round(x * 1e3, 2) for x in ts], flush=True)
print("identical", all(np.array_equal(a, b) for a, b in zip(ref, got)))
pr.enable_graph(False)
print("plain again ms", [round(x * 1e3, 2) for x in run(3)])
