#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on the MI355X hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[1], "2^22-point Goldilocks NTT + Poseidon
Merkle-cap on 1 MI355X" with the concrete shapes of SURVEY 8(d) config 2, seed 0xC0FFEE02:
  (i)  one 2^22-point forward NTT (coefficients -> evaluations),
  (ii) PolynomialBatch::from_values of a 135 x 2^15 wire matrix: 135 iNTTs, LDE x8 on the coset
       g<w>, Poseidon2 leaf sponge over 2^18 leaves of 135 limbs, Merkle tree to a 16-hash cap.
(ii) is exactly the wires commitment of one 2^15-row leaf proof at standard_recursion_config.
One step = (i) + (ii), inputs resident in HBM. `value` = commitment-equivalent leaf proofs per
second over all ranks (each rank runs its own independent steps: the map-reduce leaf proofs shard
with no data-path collective, "scaling": "weak").  The NTT GB/s-vs-HBM-peak half of the metric is
the `roofline` object (algorithmic 16 B/point over the measured duration of the NTT launches).
"""
import argparse
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
LOG_N, W, RATE, CAP = 15, 135, 3, 4
SEED = 0xC0FFEE02


def cpu_baseline(log_ntt, reps=3):
    """CPU oracle (our restatement of the plonky2 FFT; 'port'), one thread, bounded sample."""
    import oracle as O
    a = O.rand_field((1, 1 << log_ntt), SEED)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    O.fft(a[:, :1024])
    t0 = time.perf_counter()
    for _ in range(reps):
        O.fft(a)
    dt = (time.perf_counter() - t0) / reps
    return {"value": 16.0 * (1 << log_ntt) / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{reps} x 2^{log_ntt}-point forward NTT by oracle/ntt.c (radix-2, single thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    import oracle as O  # only for the synthetic input stream and the cpu_baseline leg
    ctx = mp2.Context(local_rank)

    # ---- synthetic inputs, resident in HBM before the timed region
    n_ntt = 1 << LOG_NTT
    d_poly = ctx.to_device(O.rand_field((1, n_ntt), SEED + rank))
    d_out = ctx.alloc(n_ntt * 8)
    d_wires = ctx.to_device(O.rand_field((W, 1 << LOG_N), SEED + 1000 + rank))
    batch = mp2.PolynomialBatch.from_values_dev(ctx, d_wires, LOG_N, W, RATE, CAP)

    def step():
        ctx.ntt_dev(d_poly, d_out, LOG_NTT, 1, bitrev_out=True)
        batch.recommit_from_values_dev(d_wires)

    for _ in range(args.warmup):
        step()
    ctx.sync()
    # dominant-kernel timing for the roofline line: HIP events on the context's stream
    ntt_ms = []
    for _ in range(5):
        ctx.timer_start()
        ctx.ntt_dev(d_poly, d_out, LOG_NTT, 1, bitrev_out=True)
        ntt_ms.append(ctx.timer_stop())
    commit_ms = []
    for _ in range(3):
        ctx.timer_start()
        batch.recommit_from_values_dev(d_wires)
        commit_ms.append(ctx.timer_stop())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ntt_s = float(np.median(ntt_ms)) / 1e3
        achieved = 16.0 * n_ntt / ntt_s / 1e9
        out = {
            "metric": "leaf proofs/sec (whole node) + NTT GB/s vs HBM peak",
            "value": world * args.steps / dt,
            "unit": "commitment-equivalent leaf proofs/s (one 2^22 NTT + one 135x2^15 wires commitment each)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 (Goldilocks)", "data": "synthetic",
            "config": {"workload": "configs[1]: 2^22-point Goldilocks NTT + 135x2^15 -> 2^18-leaf Poseidon2 Merkle cap(4)",
                       "log_ntt": LOG_NTT, "commit": f"{W}x2^{LOG_N}, rate_bits {RATE}, cap_height {CAP}",
                       "hasher": "Poseidon2", "sharding": f"{world} independent ranks"},
            "roofline": {"bound": "hbm", "kernel": "ntt_cols_kernel<11>+ntt_rows_kernel<11> (2^22 forward)",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "launch_ms": ntt_s * 1e3, "algorithmic_bytes": 16 * n_ntt},
            "commit_ms": float(np.median(commit_ms)),
            "merkle_perms_per_s": ((1 << (LOG_N + RATE)) * 17 + (2 << (LOG_N + RATE)) - 16) / (float(np.median(commit_ms)) / 1e3),
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(LOG_NTT)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
