"""cProfile of one 256-leaf tree step of the real-recursion driver (host glue hot spots)."""
import cProfile, importlib, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mp2 = importlib.import_module("mapreduce-plonky2_amd")
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
FW = importlib.import_module("mapreduce-plonky2_amd.framework")
C = importlib.import_module("mapreduce-plonky2_amd.circuits")
ctx = mp2.Context(0)
prover = FW.GpuProver(ctx, 0, witness_check=True)
fw = R.RecursiveCircuits([R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)], prover,
                         lambda ckt: FW.circuit_fri_params(ckt, 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
data = C.rand_field(4 * n, 7)
def step():
    level = fw.generate_proofs_batch("map", [([], [], data[4 * i:4 * i + 4]) for i in range(n)])
    names = ["map"] * n
    while len(level) > 1:
        level = fw.generate_proofs_batch("reduce", [([level[2 * i], level[2 * i + 1]], [names[2 * i], names[2 * i + 1]], None) for i in range(len(level) // 2)])
        names = ["reduce"] * len(level)
step()
pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
