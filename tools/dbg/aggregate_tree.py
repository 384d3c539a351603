"""BASELINE configs[2]: 2-to-1 aggregation of N synthetic leaf proofs on one GPU, every node = base prove()
(2^13 rows) + wrap prove() (2^12 rows) of gate-level circuits; a parent's public-input hash is the hash of
its children's wrap commitments (the data dependency that orders the levels, as the universal verifier's
public inputs do). Usage: python tools/dbg/aggregate_tree.py [n_leaves=1024] [chunk=128]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import circuits as C
import oracle as O  # rand_field only
mp2 = importlib.import_module("mapreduce-plonky2_amd")
n_leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 128
KINDS = [k for k in C.VERIFIER_KINDS if k[0] != C.PUBLIC_INPUT]  # aggregation nodes are verifier circuits; public inputs enter through the hash only
ctxs = {13: mp2.Context(0), 12: mp2.Context(0)}  # base and wrap on their own streams
ckts = {k: C.build(k, KINDS, 0xC0FFEE03 + k) for k in (13, 12)}
fps = {k: mp2.standard_recursion_params(k, (ckts[k].num_constants + 80, 135, 20, 16)) for k in (13, 12)}
d_cd = {k: ctxs[k].to_device(O.rand_field(4, 7 + k)) for k in (13, 12)}
provers = {}
def prover(k, B):
    if (k, B) not in provers:
        cx, ckt = ctxs[k], ckts[k]
        pr = mp2.BatchedProver(cx, fps[k], B)
        pr.set_preprocessed(cx.to_device(ckt.pre))
        pr.enable_permutation(80, 8); pr.enable_quotient()
        pr.set_gates([mp2.Gate(g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end) for g in ckt.gates], ckt.num_selectors)
        provers[(k, B)] = (pr, cx.to_device(np.stack([ckt.wires] * B)), cx.alloc(B * 4 * 8))
    return provers[(k, B)]
def prove(k, pis):
    """prove len(pis) nodes of shape k with the given public-input hashes; returns their wires caps"""
    out = []
    for lo in range(0, len(pis), chunk):
        part = pis[lo:lo + chunk]
        B = 1 << max(0, int(np.ceil(np.log2(len(part)))))
        pr, d_w, d_ph = prover(k, B)
        buf = np.zeros((B, 4), dtype=np.uint64); buf[:len(part)] = part
        d_ph.upload(buf)
        pr.prove([d_w, None, None], d_cd[k], d_ph)
        caps, _, _ = pr.results()
        out.append(caps[:len(part), 1, :])
    return np.concatenate(out)
# warm up every prover size (twiddle tables, allocations) outside the timed region
m = n_leaves
while m >= 1:
    for k in (13, 12):
        prover(k, min(chunk, m))
    m //= 2
prove(13, np.zeros((min(chunk, n_leaves), 4), dtype=np.uint64)); prove(12, np.zeros((min(chunk, n_leaves), 4), dtype=np.uint64))
t0 = time.perf_counter()
pis = np.zeros((n_leaves, 4), dtype=np.uint64); pis[:, 0] = np.arange(n_leaves, dtype=np.uint64)
n_proofs, level = 0, 0
while True:
    base_caps = prove(13, pis)
    wrap_pis = ctxs[12].hash_no_pad_batch(base_caps, 4)
    wrap_caps = prove(12, wrap_pis)
    n_proofs += len(pis)
    print(f"level {level}: {len(pis)} nodes, {time.perf_counter() - t0:.3f} s", flush=True)
    if len(pis) == 1:
        break
    pis = ctxs[13].hash_no_pad_batch(wrap_caps.reshape(len(pis) // 2, -1), 4)
    level += 1
dt = time.perf_counter() - t0
print(f"aggregation tree over {n_leaves} leaves: {n_proofs} framework proofs (base 2^13 + wrap 2^12 each) in {dt:.3f} s = {n_proofs / dt:.1f} proofs/s; root pi {pis[0].tolist()}")
