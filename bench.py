#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric ("leaf proofs/sec (whole node) + NTT GB/s vs HBM peak") on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload. The metric is quoted on the 2^20-row table build (configs[3]); its unit of work is the
framework leaf proof, which the recursion framework always produces as one base `prove()` plus
one wrap `prove()` down to 2^12 rows (recursion-framework/src/circuit_builder.rs:286-311,
wrap_circuit.rs:122-148). One step = one batch of `--batch` such leaf proofs, shaped as SURVEY
8(d) config 3 says (base 2^13 + wrap 2^12, standard_recursion_config: constants + 80 sigmas,
135 wires, 20 Z/partial products, 16 quotient chunks, rate 1/8, cap 16, FRI [4,4], 16-bit PoW, 28
queries), on synthetic witness matrices that are resident in HBM before the timed region. What
runs per proof is everything `prove()` does after witness generation -- wires commitment, Z / partial
products, quotient polynomials (permutation terms and the gate constraints), their commitments,
Fiat-Shamir, openings, FRI (HOT LOOPS 1-3 of SURVEY 3.1) -- for satisfied synthetic circuits composed as the reference composes them
(tests/circuits.py): the wrap circuit has the gate set of plonky2's recursive verifier (Noop, Constant,
PublicInput, Arithmetic, ArithmeticExtension, MulExtension, BaseSum<2>, Exponentiation, Reducing,
ReducingExtension, RandomAccess, CosetInterpolation, Poseidon2: 13 gates), the base (leaf) circuit adds
the user-logic gates (BaseSum<4>, U32Arithmetic, U32RangeCheck, U32Subtraction, U32AddMany, Comparison:
19 gates); as in plonky2 every gate of a circuit is evaluated at every LDE point, so the cost depends on
the gate set, not on the row mix. Rows are dealt over the gates with random copy constraints. Witness
generation stays on the host. The proofs verify (FRI + the PLONK identity at zeta with the gate terms,
tests/test_gpu_gates.py).
Leaf proofs shard across ranks with no data-path collective ("scaling": "weak"); the per-rank
multiset digests meet in one 160-byte all_gather outside the per-proof path.

The NTT half of the metric is the `roofline` object: BASELINE configs[1]'s 2^22-point forward NTT,
timed with HIP events on the context's stream in this same run, algorithmic 16 B/point.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
LOG_NTT = 22
ORACLE_W = (None, 135, 20, 16)  # constants (selectors + 2) + 80 sigmas: from the circuit | wires | Z, partial products | quotient chunks
NUM_ROUTED = 80  # standard_recursion_config: 80 routed wires, quotient_degree_factor 8 => 2 x (1 + 9) Z / partial products
SEED = 0xC0FFEE03


def gpu_clocks(device):
    """rocm-smi's current clocks of the device (state the measured clocks with every report)"""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
        return {k.strip(): v for k, v in card.items() if "clock" in k.lower() and ("sclk" in k.lower() or "mclk" in k.lower() or "fclk" in k.lower())}
    except Exception as e:  # no rocm-smi / no permission: say so instead of guessing
        return {"error": str(e)[:80]}


def build_circuit(C, role, k):
    """the synthetic circuit of a role: base = leaf gate set (verifier gadget + u32 logic), wrap = verifier gate set"""
    return C.build(k, C.LEAF_KINDS if role == "base" else C.VERIFIER_KINDS, SEED + k + (0 if role == "base" else 100))


def cpu_baseline(base_bits, budget_s=15.0, variant=0):
    """The CPU oracle (our restatement of the same pipeline; kind 'port') on all host cores, on a bounded sample
    of the same workload. Leaf proofs are independent, so the cores are used the way a CPU deployment would use
    them: groups of 32 threads (OpenMP inside a proof: polynomials, leaves, quotient points, PoW candidates) prove
    one leaf proof each, all groups at once; rounds of that until ~budget_s seconds are spent."""
    import threading
    import circuits as C
    import oracle as O
    cores = os.cpu_count() or 1
    groups = max(1, cores // 32)
    per_group = max(1, cores // groups)
    shapes = []
    for role, k in (("base", base_bits), ("wrap", 12)):
        ckt = build_circuit(C, role, k)
        shapes.append((O.standard_params(k, (ckt.pre.shape[0],) + ORACLE_W[1:], variant=variant), ckt))
    cd = O.rand_field(4, 1)
    omp = ctypes.CDLL("libgomp.so.1")

    def one_proof():
        omp.omp_set_num_threads(per_group)  # per-thread ICV: the parallel regions this thread opens
        for ofp, ckt in shapes:
            C.prove(ckt, ofp, cd)

    t_total, n_proofs = 0.0, 0
    while n_proofs < 6 * groups and (n_proofs == 0 or t_total * (n_proofs + groups) / n_proofs < budget_s):
        ts = [threading.Thread(target=one_proof) for _ in range(groups)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        t_total += time.perf_counter() - t0
        n_proofs += groups
    return {"value": n_proofs / t_total, "unit": "leaf proofs/s", "cores": cores, "kind": "port",
            "sample": f"{n_proofs} leaf proof(s) = base 2^{base_bits} + wrap 2^12 prove() of the same gate-level circuits by oracle/, "
                      f"{groups} proof(s) at a time with {per_group} OpenMP threads each (polynomials, leaves, quotient points, PoW "
                      "candidates; FRI composition and transcript are single-threaded)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128, help="leaf proofs per step and rank")
    ap.add_argument("--base-bits", type=int, default=13)
    ap.add_argument("--streams", type=int, default=4, help="HIP streams: 1 = both shapes on one; 2 = one per shape; 4 = two half-batches per shape")
    ap.add_argument("--host-inputs", action="store_true",
                    help="PCIe-inclusive variant: every step uploads its wire / quotient matrices from pinned host memory "
                         "on the prover's stream (never the headline value; see DESIGN.md)")
    ap.add_argument("--hasher", choices=("poseidon2", "poseidon"), default="poseidon2",
                    help="poseidon2 = the reference's default config (Poseidon2GoldilocksConfig); poseidon = its "
                         "`original_poseidon` feature (mp2-common/src/lib.rs:37-40), the variant pinned against the reference")
    ap.add_argument("--witness-check", action="store_true",
                    help="also run the device-side witness check (gate + copy constraints on H) inside every prove, "
                         "as plonky2's prove() does before it panics on a bad witness")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        # MP2G_BENCH_BACKEND=gloo runs the multi-rank code path on a box with fewer GPUs than ranks (ranks share
        # devices round-robin; collectives go through host tensors) -- a plumbing check, not a measurement
        backend = os.environ.get("MP2G_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    VARIANT = 0 if args.hasher == "poseidon2" else 1
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    sharding = importlib.import_module("mapreduce-plonky2_amd.sharding")
    import circuits as C  # synthetic circuit + witness generator (pure Python)
    import oracle as O  # rand_field (SplitMix64 stream); the oracle library itself is used by the cpu_baseline leg only
    # two contexts = two HIP streams on the same GPU: the base and the wrap prover run concurrently, so
    # the latency-bound stretches of one (transcript, top Merkle levels) hide under the other's sponges
    ctx = mp2.Context(local_rank)
    n_ctx = max(1, args.streams)
    ctxs = [ctx] + [mp2.Context(local_rank) for _ in range(n_ctx - 1)]
    B = args.batch
    # provers: (shape, context, share of the batch). 1 stream: both shapes on it; 2: one each;
    # 4: every shape split into two half-batches
    if n_ctx >= 4:
        plan = [("base", args.base_bits, ctxs[0], B // 2), ("wrap", 12, ctxs[1], B // 2), ("base", args.base_bits, ctxs[2], B - B // 2),
                ("wrap", 12, ctxs[3], B - B // 2)]
    elif n_ctx >= 2:
        plan = [("base", args.base_bits, ctxs[0], B), ("wrap", 12, ctxs[1], B)]
    else:
        plan = [("base", args.base_bits, ctx, B), ("wrap", 12, ctx, B)]

    # ---- synthetic inputs, resident in HBM before the timed region -----------------------------
    provers = []
    circuits = {}
    for role, k, cx, nb in plan:
        n = 1 << k
        # a satisfied gate-level circuit of the role's gate set: rows dealt over the gates, random copy constraints
        if role not in circuits:
            circuits[role] = build_circuit(C, role, k)
        ckt = circuits[role]
        oracle_w = (ckt.pre.shape[0],) + ORACLE_W[1:]
        fp = mp2.standard_recursion_params(k, oracle_w, variant=VARIANT)
        pr = mp2.BatchedProver(cx, fp, nb)
        pr.set_preprocessed(cx.to_device(ckt.pre))
        pr.enable_permutation(NUM_ROUTED, 8)  # Z / partial products on the device from wires + sigmas
        pr.enable_quotient()                  # quotient chunks on the device
        pr.set_gates([mp2.Gate(g.kind, g.p0, g.p1, g.p2, g.selector_index, g.group_start, g.group_end) for g in ckt.gates],
                     ckt.num_selectors)       # ... including the gate constraint terms
        if args.witness_check:
            pr.enable_witness_check()
        # the witness, tiled over the batch; the unrouted cells of one Noop row are free, so re-drawing
        # them per proof keeps the commitments, transcripts and proofs of the batch distinct
        wires_one = ckt.wires
        noop_row = ckt.instances.index(next(i for i, g in enumerate(ckt.gates) if g.kind == C.NOOP))
        d_vals = []
        for i, w in enumerate(oracle_w[1:]):
            if i >= 1:
                d_vals.append(None)  # oracles 2 and 3: produced by the prover itself
                continue
            one = wires_one.copy()
            buf = cx.alloc(nb * w * n * 8)
            for b in range(nb):
                one[NUM_ROUTED:, noop_row] = O.rand_field(w - NUM_ROUTED, SEED + 1000 * len(provers) + 31 * rank + b)
                mp2._ck(mp2.load().mp2g_h2d(cx.h, ctypes.c_void_p(buf.ptr.value + b * w * n * 8), mp2._p(one), ctypes.c_size_t(one.nbytes)))
            d_vals.append(buf)
        d_cd = cx.to_device(O.rand_field(4, SEED + 7))
        d_ph = cx.to_device(np.stack([ckt.pi_hash] * nb))  # bound to the wires by the PublicInput gate
        staging = []
        if args.host_inputs:
            for buf in d_vals:
                if buf is None:
                    continue
                view, hptr = cx.host_alloc(buf.nbytes)
                view[:] = np.frombuffer(buf.download((buf.nbytes // 8,)).tobytes(), dtype=np.uint8)
                staging.append((buf, hptr, buf.nbytes))
        provers.append((pr, d_vals, d_cd, d_ph, cx, staging))
    n_ntt = 1 << LOG_NTT
    d_poly = ctx.to_device(O.rand_field((1, n_ntt), 0xC0FFEE02 + rank))
    d_out = ctx.alloc(n_ntt * 8)

    def step():
        for pr, d_vals, d_cd, d_ph, cx, staging in provers:
            for buf, hptr, nbytes in staging:
                cx.h2d_async(buf, hptr, nbytes)
            pr.prove(d_vals, d_cd, d_ph)

    def sync_all():
        for c in ctxs:
            c.sync()

    for _ in range(args.warmup):
        step()
    sync_all()

    # roofline leg: the 2^22 NTT, HIP events on the stream the kernels are launched on
    ctx.ntt_dev(d_poly, d_out, LOG_NTT, 1, bitrev_out=True)
    ntt_ms = []
    for _ in range(10):
        ctx.timer_start()
        ctx.ntt_dev(d_poly, d_out, LOG_NTT, 1, bitrev_out=True)
        ntt_ms.append(ctx.timer_stop())

    # the same kernel family on the shape the prover actually runs: 8192 transforms of 2^12 points
    # (LDE-sized batch, no tail effects); reported beside the roofline leg, not as `value`
    nb12 = 8192
    d_b12 = ctx.alloc(nb12 * 4096 * 8)
    ctx.ntt_dev(d_b12, d_b12, 12, nb12, bitrev_out=True)
    ctx.timer_start()
    ctx.ntt_dev(d_b12, d_b12, 12, nb12, bitrev_out=True)
    ntt12_ms = ctx.timer_stop()
    d_b12.free()

    # the sponge, the kernel that takes half of a step: 2^21 leaves of 136 limbs (17 permutations each), the rate
    # of the instruction-bound Poseidon2 permutation; reported beside the roofline leg
    n_hash, limbs = 1 << 21, 136
    d_hin = ctx.alloc(n_hash * limbs * 8)
    d_hout = ctx.alloc(n_hash * 4 * 8)
    hargs = (ctx.h, VARIANT, d_hin.ptr, limbs, n_hash, 4, d_hout.ptr)
    mp2._ck(mp2.load().mp2g_hash_no_pad_batch_dev(*hargs))
    ctx.timer_start()
    mp2._ck(mp2.load().mp2g_hash_no_pad_batch_dev(*hargs))
    hash_ms = ctx.timer_stop()
    d_hin.free(); d_hout.free()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        sync_all()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if args.witness_check:
        for pr, *_ in provers:
            pr.witness_status()  # raises on a violated constraint
    # per-stage split of one batch per shape, each prover alone on the GPU (outside the timed region)
    stages = {}
    if rank == 0:
        seen_shapes = set()
        for (role, k, cx, nb), (pr, d_vals, d_cd, d_ph, _, _) in zip(plan, provers):
            if role in seen_shapes:
                continue
            seen_shapes.add(role)
            sync_all()
            pr.enable_timing(True)
            pr.prove(d_vals, d_cd, d_ph)
            ms = pr.stage_ms()
            pr.enable_timing(False)
            stages[f"{role} 2^{k} x {nb}"] = {s: round(v, 3) for s, v in ms.items()}

    # the per-rank multiset digest (2^16 rows x 4 value columns, device resident) meets in one
    # all_gather of one encoded point per rank, outside the per-proof path
    rows, n_cols = 1 << 16, 4
    rng = np.random.default_rng(0xC0FFEE04 + rank)
    d_ids = ctx.to_device(O.rand_field(n_cols, 0xC0FFEE04))
    d_values = ctx.to_device(rng.integers(0, 1 << 32, size=(rows, n_cols, 8), dtype=np.uint32))
    d_unique = ctx.to_device(rng.integers(0, 1 << 32, size=(rows, 1, 8), dtype=np.uint32))
    mp2.compute_table_row_digest_dev(ctx, d_ids, n_cols, d_values, d_unique, 1, rows)
    t1 = time.perf_counter()
    w = mp2.compute_table_row_digest_dev(ctx, d_ids, n_cols, d_values, d_unique, 1, rows)
    digest_s = time.perf_counter() - t1
    if dist is not None:
        allw = sharding.all_gather_words(dist, w, device=torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else None)
        w = mp2.curve_sum(ctx, allw)

    if rank == 0:
        # HBM-side bytes of the same two launches from the TCC counters (collected in separate
        # --pmc passes and corrected as MI355X_MICROARCH.md prescribes; profiles/r01/ntt_traffic.json)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01", "ntt_traffic.json")) as f:
                traffic = json.load(f)["ntt_2p22_forward_bitrev"]["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            pass
        ntt_s = float(np.median(ntt_ms)) / 1e3
        achieved = 16.0 * n_ntt / ntt_s / 1e9
        out = {
            "metric": "leaf proofs/sec (whole node) + NTT GB/s vs HBM peak, 2^20-row table build, 1/2/4/8 GPU",
            "value": world * args.steps * B / dt,
            "unit": "leaf proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 (Goldilocks field)", "data": "synthetic",
            "config": {"workload": f"configs[3]-shaped leaf proofs: base 2^{args.base_bits} + wrap 2^12 prove() from the wire matrix "
                                   "(commitments, permutation argument, quotient with gate constraints, Fiat-Shamir, openings, FRI) at "
                                   "standard_recursion_config; base circuit = 19 gates (recursive-verifier set + u32 / comparison logic), wrap circuit = the 13 "
                                   "gates of plonky2's recursive verifier; "
                                   "roofline leg = configs[1] 2^22-point NTT",
                       "batch_per_rank": B, "streams": args.streams, "witness_check": bool(args.witness_check), "host_inputs": bool(args.host_inputs), "oracle_polys": {r: [int(c.pre.shape[0])] + list(ORACLE_W[1:]) for r, c in circuits.items()},
                       "gates": {r: len(c.gates) for r, c in circuits.items()}, "hasher": "Poseidon2" if VARIANT == 0 else "Poseidon (original_poseidon feature)",
                       "sharding": f"{world} rank(s), leaf proofs independent, digest all_gather 160 B"},
            "roofline": {"bound": "hbm", "kernel": "ntt (2^22 forward, both launches)",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "launch_ms": ntt_s * 1e3, "algorithmic_bytes": 16 * n_ntt},
            "ntt_batched_2p12": {"transforms": nb12, "GBps": 16.0 * nb12 * 4096 / (ntt12_ms / 1e3) / 1e9,
                                 "frac_of_hbm_peak": 16.0 * nb12 * 4096 / (ntt12_ms / 1e3) / 1e9 / HBM_PEAK_GBPS},
            "sponge": {"hasher": args.hasher, "permutations_per_s": n_hash * (limbs // 8) / (hash_ms / 1e3), "bound": "VALU issue (integer ALU)",
                                 "input": f"{n_hash} x {limbs} limbs, hash_no_pad, resident"},
            "stage_ms": stages,
            "clocks": gpu_clocks(local_rank),
            "digest_rows_per_s": rows / digest_s,
            "digest_check": [int(x) for x in w],
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(args.base_bits, variant=VARIANT)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    for c in reversed(ctxs):
        c.close()


if __name__ == "__main__":
    main()
