"""Real recursion at scale: an N-leaf map-reduce tree of recursion-framework/tests/integration.rs's circuits with universal
verifiers, witnesses by the recorded witness programs on host threads, proofs by batched HIP provers. Prints the time per
level and framework proofs/s. Usage: python tools/dbg/real_tree.py [n_leaves=64] [witness_check=0]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
mp2 = importlib.import_module("mapreduce-plonky2_amd")
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
FW = importlib.import_module("mapreduce-plonky2_amd.framework")
C = importlib.import_module("mapreduce-plonky2_amd.circuits")
n_leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = mp2.Context(0)
prover = FW.GpuProver(ctx, witness_check=bool(int(sys.argv[2])) if len(sys.argv) > 2 else False)
t0 = time.perf_counter()
fw = R.RecursiveCircuits([R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)], prover, FW.circuit_fri_params)
print(f"build_circuits_params: {time.perf_counter() - t0:.1f} s; shapes {({k: [c[0].log_n for c in v] for k, v in fw.chains.items()})}", flush=True)
data = C.rand_field(4 * n_leaves, 0xC0FFEE03)
for rep in range(2):  # the first pass creates the provers of every batch size
    t0 = time.perf_counter()
    level = fw.generate_proofs_batch("map", [([], [], data[4 * i:4 * i + 4]) for i in range(n_leaves)])
    t1 = time.perf_counter()
    print(f"pass {rep}: {n_leaves} map proofs {t1 - t0:.2f} s", flush=True)
    names, n_proofs = ["map"] * n_leaves, n_leaves
    while len(level) > 1:
        t2 = time.perf_counter()
        level = fw.generate_proofs_batch("reduce", [([level[2 * i], level[2 * i + 1]], [names[2 * i], names[2 * i + 1]], None) for i in range(len(level) // 2)])
        names = ["reduce"] * len(level)
        n_proofs += len(level)
        print(f"   {len(level)} reduce proofs {time.perf_counter() - t2:.2f} s", flush=True)
    dt = time.perf_counter() - t0
    print(f"pass {rep}: {n_proofs} framework proofs (real circuits, witness generation included) in {dt:.2f} s = {n_proofs / dt:.1f} proofs/s; root sum ok "
          f"{int(level[0][3][0]) == sum(int(x) for x in data if int(x) % 2 == 0) % C.P}", flush=True)
prover.free(); ctx.close()
