"""time mp2g_witness_program_run_dev against the host replay for the reduce circuit's base and wrap programs"""
import importlib, os, sys, time
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
print("levels", prog.n_levels, "inputs", prog.n_inputs)
for B in (1, 8, 32, 128):
    inp = np.tile(row, (B, 1))
    d_in, d_w, d_pr = ctx.to_device(inp), ctx.alloc(B * 135 * n * 8), ctx.alloc(B * prog.probe.size * 8)
    prog.run_dev(ctx, d_in, B, d_w, d_pr); ctx.sync()
    ctx.timer_start()
    prog.run_dev(ctx, d_in, B, d_w, d_pr)
    ms = ctx.timer_stop()
    t0 = time.perf_counter()
    prog.run(inp, threads=0)
    host = (time.perf_counter() - t0) * 1e3
    print(f"B={B}: device {ms:.2f} ms, host ({os.cpu_count()} threads) {host:.1f} ms")
    d_in.free(); d_w.free(); d_pr.free()
