"""The recursion framework with real circuits on the HIP prover: four map proofs, two reduce levels above them, every
proof = base prove() + wrap prove() by libmp2gpu with the device-side witness check on (prove()'s panic on a bad
witness), universal verifiers inside the reduce circuit; the root's public inputs are the dataset's (integration.rs:224-228)
and the root proof passes the oracle's verifier."""
import importlib

import numpy as np
import pytest

import circuits as C
import oracle as O

pytestmark = pytest.mark.gpu
R = importlib.import_module("mapreduce-plonky2_amd.recursion")
FW = importlib.import_module("mapreduce-plonky2_amd.framework")


def test_map_reduce_tree_of_real_proofs(ctx, mp2):
    prover = FW.GpuProver(ctx)
    circs = [R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)]
    fw = R.RecursiveCircuits(circs, prover, FW.circuit_fri_params)
    data = O.rand_field(16, 0xC0FFEE03)
    level = [fw.generate_proof("map", [], [], data[4 * i:4 * i + 4]) for i in range(4)]
    names = ["map"] * 4
    while len(level) > 1:
        level = [fw.generate_proof("reduce", [level[2 * i], level[2 * i + 1]], [names[2 * i], names[2 * i + 1]], None) for i in range(len(level) // 2)]
        names = ["reduce"] * len(level)
    pis = level[0][3]
    assert int(pis[0]) == sum(int(x) for x in data if int(x) % 2 == 0) % O.P
    hs = [O.hash_n_to_m_no_pad(data[4 * i:4 * i + 4], 4) for i in range(4)]
    while len(hs) > 1:
        hs = [O.hash_n_to_m_no_pad(np.concatenate([hs[2 * i], hs[2 * i + 1]]), 4) for i in range(len(hs) // 2)]
    assert np.array_equal(pis[1:5], hs[0])
    assert np.array_equal(pis[5:], np.asarray(fw.set_digest, dtype=np.uint64))
    wckt, wcap, wdig = fw.chains["reduce"][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(pis, 4), *level[0][:3]) == 0
    prover.free()


def test_sixteen_leaf_tree_through_the_witness_programs(ctx, mp2):
    """the same framework at 16 leaves, level by level in batches: witnesses by the recorded witness programs
    (mp2g_witness_program_run on host threads), proofs by batched HIP provers with the witness check on. 31 framework
    proofs = 62 prove() calls' worth of real circuits (map 2^6 + wrap 2^12, reduce 2^13 + wrap 2^12)."""
    prover = FW.GpuProver(ctx)
    circs = [R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)]
    fw = R.RecursiveCircuits(circs, prover, FW.circuit_fri_params)
    n_leaves = 16
    data = O.rand_field(4 * n_leaves, 0xC0FFEE03)
    level = fw.generate_proofs_batch("map", [([], [], data[4 * i:4 * i + 4]) for i in range(n_leaves)])
    # the batch path and the Python-builder path agree on a leaf
    one = fw.generate_proof("map", [], [], data[0:4])
    assert all(np.array_equal(x, y) for x, y in zip(level[0], one))
    names = ["map"] * n_leaves
    while len(level) > 1:
        jobs = [([level[2 * i], level[2 * i + 1]], [names[2 * i], names[2 * i + 1]], None) for i in range(len(level) // 2)]
        level = fw.generate_proofs_batch("reduce", jobs)
        names = ["reduce"] * len(level)
    pis = level[0][3]
    assert int(pis[0]) == sum(int(x) for x in data if int(x) % 2 == 0) % O.P
    hs = [O.hash_n_to_m_no_pad(data[4 * i:4 * i + 4], 4) for i in range(n_leaves)]
    while len(hs) > 1:
        hs = [O.hash_n_to_m_no_pad(np.concatenate([hs[2 * i], hs[2 * i + 1]]), 4) for i in range(len(hs) // 2)]
    assert np.array_equal(pis[1:5], hs[0]) and np.array_equal(pis[5:], np.asarray(fw.set_digest, dtype=np.uint64))
    wckt, wcap, wdig = fw.chains["reduce"][-1]
    assert C.verify(wckt, C.oracle_params(wckt), wdig, O.hash_n_to_m_no_pad(pis, 4), *level[0][:3]) == 0
    prover.free()


def test_independent_trees_in_parallel_sessions(ctx, mp2):
    """two independent 4-leaf trees proved at the same time, one thread + GPU context + ProofSession each (the way
    bench.py --workload recursion --trees N fills the GPU while another tree's witnesses are generated): same root
    proofs, word for word, as the two trees proved one after the other on the framework's own session"""
    import threading
    prover = FW.GpuProver(ctx)
    circs = [R.FrameworkCircuit("map", 0, R.map_logic, 5), R.FrameworkCircuit("reduce", 2, R.reduce_logic, 5)]
    fw = R.RecursiveCircuits(circs, prover, FW.circuit_fri_params)
    datas = [O.rand_field(16, 0xC0FFEE03 + t) for t in range(2)]

    def tree(data, session):
        level = fw.generate_proofs_batch("map", [([], [], data[4 * i:4 * i + 4]) for i in range(4)], session=session)
        names = ["map"] * 4
        while len(level) > 1:
            jobs = [([level[2 * i], level[2 * i + 1]], [names[2 * i], names[2 * i + 1]], None) for i in range(len(level) // 2)]
            level = fw.generate_proofs_batch("reduce", jobs, session=session)
            names = ["reduce"] * len(level)
        return level[0]

    sequential = [tree(d, None) for d in datas]
    ctxs = [mp2.Context(0), mp2.Context(0)]
    provers = [FW.GpuProver(c) for c in ctxs]
    sessions = [R.ProofSession(p) for p in provers]
    out = [None, None]
    ths = [threading.Thread(target=lambda t=t: out.__setitem__(t, tree(datas[t], sessions[t]))) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for t in range(2):
        assert out[t] is not None
        assert all(np.array_equal(x, y) for x, y in zip(out[t], sequential[t]))
        assert int(out[t][3][0]) == sum(int(x) for x in datas[t] if int(x) % 2 == 0) % O.P
    for p in provers + [prover]:
        p.free()
    for c in ctxs:
        c.close()


def test_two_ranks_real_recursion_with_proof_handoff():
    """bench.py --workload recursion on two ranks (gloo rendezvous, both on the test box's GPU): each rank proves an 8-leaf
    tree of real framework proofs, then rank 1's root proof travels as bincode bytes (mp2g_proof_serialize ->
    send/recv -> mp2g_proof_deserialize) to rank 0, whose reduce node verifies both roots in-circuit; the run itself
    asserts that the final public input is the sum of the even elements of both ranks' data."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "recursion", "--batch", "8", "--trees", "2", "--steps", "1", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MP2G_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["shapes"] == {"map": [6, 12], "reduce": [13, 12]}
    assert line["framework_proofs_per_s"] > 0 and len(line["config"]["root_public_inputs"]) == 9
