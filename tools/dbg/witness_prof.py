"""where the device witness replay spends its time, per class of level (needs a library built with -DWIT_PROF: tools/dbg/witness_prof.sh)"""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
mp2 = importlib.import_module("mapreduce-plonky2_amd")
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
FW = importlib.import_module("mapreduce-plonky2_amd.framework")
ctx = mp2.Context(0)
prover = FW.GpuProver(ctx)
fw = R.RecursiveCircuits([R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)], prover, FW.circuit_fri_params)
data = O.rand_field(8, 1)
leaves = fw.generate_proofs_batch("map", [([], [], data[:4]), ([], [], data[4:])])
vd = fw.vds["map"]
row = np.concatenate([np.asarray(fw.set_digest, dtype=np.uint64)] + [R.universal_inputs(leaves[i], vd, fw.membership(vd[1])) for i in range(2)])
prog = fw.witness_programs("reduce")[0]
n = 1 << prog.log_n
names = ["narrow Poseidon2 levels (lane-cooperative)", "wide Poseidon2 levels (one lane per row)", "reducing / interpolation / inverse levels", "other levels"]
for B in (1, 32, 128):
    inp = np.tile(row, (B, 1))
    d_in, d_w, d_pr = ctx.to_device(inp), ctx.alloc(B * 135 * n * 8), ctx.alloc(B * prog.probe.size * 8)
    prog.run_dev(ctx, d_in, B, d_w, d_pr); ctx.sync()
    ctx.timer_start(); prog.run_dev(ctx, d_in, B, d_w, d_pr); ms = ctx.timer_stop()
    prof = (ctypes.c_uint64 * 8)()
    assert mp2.load().mp2g_dbg_witness_prof(prof) == 0
    tot = sum(int(prof[2 * k]) for k in range(4))
    print(f"B={B}: {ms:.2f} ms; block 0: {tot} shader cycles in levels")
    for k in range(4):
        print(f"   {names[k]:48s} {int(prof[2 * k + 1]):4d} levels {int(prof[2 * k]) / 1e3:9.1f} kcycles ({100 * int(prof[2 * k]) / tot:4.1f} %), {int(prof[2 * k]) / max(1, int(prof[2 * k + 1])):8.0f} cycles per level")
