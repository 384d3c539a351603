"""Where a step of `bench.py --workload recursion` goes: host witness programs, upload, prove() and the downloads, summed per phase."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
mp2 = importlib.import_module("mapreduce-plonky2_amd")
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
FW = importlib.import_module("mapreduce-plonky2_amd.framework")
C = importlib.import_module("mapreduce-plonky2_amd.circuits")
T = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t; return r
    return w
mp2.WitnessProgram.run = timed("witness programs (host threads)", mp2.WitnessProgram.run)
mp2.DeviceBuffer.upload = timed("upload", mp2.DeviceBuffer.upload)
FW.CircuitProver.prove = timed("prove() launch", FW.CircuitProver.prove)
FW.CircuitProver.results = timed("sync + download", FW.CircuitProver.results)
R.proof_inputs = timed("proof_inputs (numpy)", R.proof_inputs)
R.universal_inputs = timed("universal_inputs (numpy, includes proof_inputs)", R.universal_inputs)
ctx = mp2.Context(0)
prover = FW.GpuProver(ctx, 0, witness_check=True)
_ws = mp2.BatchedProver.witness_status
mp2.BatchedProver.witness_status = timed("witness_status (sync)", _ws)
fw = R.RecursiveCircuits([R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)], prover,
                         lambda ckt: FW.circuit_fri_params(ckt, 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
data = C.rand_field(4 * n, 7)
def step():
    level = fw.generate_proofs_batch("map", [([], [], data[4 * i:4 * i + 4]) for i in range(n)])
    names = ["map"] * n
    while len(level) > 1:
        level = fw.generate_proofs_batch("reduce", [([level[2 * i], level[2 * i + 1]], [names[2 * i], names[2 * i + 1]], None) for i in range(len(level) // 2)])
        names = ["reduce"] * len(level)
step(); T.clear()
t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
print(f"{n} leaves: {dt*1e3:.0f} ms per step, {(2*n-1)/dt:.0f} framework proofs/s")
for k, v in sorted(T.items(), key=lambda x: -x[1]):
    print(f"  {v*1e3:8.1f} ms  {k}")
