"""Per-kind gate kernel check at many points (debug aid)."""
import faulthandler, importlib, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) == 1:
    for k in range(1, 14):
        r = subprocess.run([sys.executable, __file__, str(k)], capture_output=True, text=True, timeout=120)
        print(k, r.returncode, r.stdout.strip()[-200:], r.stderr.strip()[-150:].replace("\n", " | "), flush=True)
    sys.exit(0)
import circuits as C
import oracle as O
mp2 = importlib.import_module("mapreduce-plonky2_amd")
ctx = mp2.Context(0)
kind = next(k for k in C.ALL_KINDS if k[0] == int(sys.argv[1]))
ckt = C.build(5, [(C.NOOP, 0, 0, 0), kind], 21)
npts = 20000
consts = O.rand_field((ckt.num_constants, npts), 1)
wires = O.rand_field((C.NUM_WIRES, npts), 2)
gates = [mp2.Gate(g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end) for g in ckt.gates]
got = mp2.eval_gate_constraints(ctx, gates, ckt.num_selectors, consts, wires, ckt.pi_hash)
want = C.eval_on_points(ckt, consts, wires)
print("match", np.array_equal(got, want))
