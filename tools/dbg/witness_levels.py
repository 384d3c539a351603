"""What the dependency levels of a recursive-verifier witness program hold (no GPU): the wrap circuit of the map circuit, built by
recursion.py; levels as csrc/witness.hip schedules them (level = 1 + max level of the slots read). Answers where a wave-level scan of the
ReducingGate chains could help the device replay (witness_dev.hip): python tools/dbg/witness_levels.py > profiles/r04/witness_levels.txt"""
import collections, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import circuits as C, oracle as O
from test_recursion import verifier_data
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
FW = importlib.import_module("mapreduce-plonky2_amd.framework")
base = R.map_circuit(O.rand_field(4, 77))
cap, cd = verifier_data(base)
inner = R.InnerCircuit(base, FW.circuit_fri_params(base), cap, cd, len(base.public_inputs))
b = R.Builder(strict=False)
b.register_public_inputs(R.verify_proof_circuit(b, inner, *R.dummy_proof(inner)))
w = b.build(min_log_n=12)
names = {1: "ARITH", 2: "ARITH_EXT", 3: "P2", 4: "BASE_SUM", 5: "RA", 6: "REDUCING", 7: "REDUCING_EXT", 8: "COSET", 9: "WIRE", 10: "DIV_EXT", 11: "LO63", 12: "HI", 13: "SPLIT", 15: "POSEIDON"}
lvl, per = {}, collections.defaultdict(collections.Counter)
for pos, op in R.tape_instructions(w.tape):
    if op == R.OP_PAR:
        continue
    rd, wr, _, _ = R.instruction_slots(w.tape, pos)
    l = 1 + max([lvl.get(int(s), 0) for s in rd], default=0)
    for s in wr:
        lvl[int(s)] = l
    per[l][names[op]] += 1
tot = collections.Counter()
for c in per.values():
    tot.update(c)
print(f"wrap circuit of the map circuit: 2^{w.log_n} rows, {w.n_used_rows} used, {sum(tot.values())} instructions in {len(per)} levels: {dict(tot)}")
p2 = [c["P2"] for c in per.values() if c["P2"]]
print(f"levels with Poseidon2 rows: {len(p2)}; of these with <= 2 rows: {sum(1 for x in p2 if x <= 2)}, with > 64 rows: {sum(1 for x in p2 if x > 64)} (widths {sorted(set(x for x in p2 if x > 64))})")
print("levels holding Reducing / ReducingExtension / CosetInterpolation / extension-division instructions:")
for l in sorted(per):
    c = per[l]
    if any(k in c for k in ("REDUCING", "REDUCING_EXT", "COSET", "DIV_EXT")):
        print(f"  level {l}: {dict(c)}")
