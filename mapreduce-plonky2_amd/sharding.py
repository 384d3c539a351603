"""Sharding of the map-reduce over the GPUs of one node (one process per GPU).

The reference has no distributed runtime: its harness walks a ryhope work plan sequentially
(mp2-v1/tests/common/celltree.rs:54-189; `UpdateTree::into_workplan`, ryhope/src/storage/
updatetree.rs:154-163,362-470: a node becomes Ready when all its children are done). Here:

  * leaf proofs / table rows are block-partitioned across ranks and need no communication;
  * the aggregation tree above the shard boundary takes log2(world) pairwise hand-offs of
    serialized child proofs (`tree_handoff_plan`, `exchange_bytes`);
  * the multiset digest is a commutative group sum: every rank contributes one point
    (20 limbs, 160 B) through one all_gather and adds the `world` points locally
    (`all_gather_words`). RCCL has no user-defined reduction, and at 160 B the exchange is
    latency-bound, so an all_gather + local add beats any ring all-reduce formulation.

Everything here is backend-agnostic torch.distributed ("nccl" = RCCL on the GPU box, "gloo" in the
CPU tests); no arithmetic happens in this module.
"""
import re

import numpy as np


# Device bytes per proof in flight (round 5: the provers of a context share one scratch -- csrc/ctx.h `prover_scratch`):
#  * the scratch holds the working set of ONE prove() at a time: per row of the LARGEST circuit step of the build the x8 LDEs of the
#    135 wire, 20 Z / partial-product and 16 quotient polynomials with their coefficients, the Merkle levels of those three oracles,
#    the FRI layers, the quotient values, plus the witness executor's row-major staging matrix;
#  * every chain step keeps its own hand-over buffers: the wire matrix (135 words per row), inputs, probe, caps, openings, proof words
#    -- per row of EVERY step of every circuit.
# With MP2G_SHARE_SCRATCH=0 every prover owns its working set: the first figure is paid per row of every step of every circuit (rounds
# 1-4: 16 KB x the sum of the rows; 160 GB measured at 4 workers x 32 proofs in flight).
SCRATCH_BYTES_PER_ROW = 17 * 1024
CHAIN_BYTES_PER_ROW = 3 * 512
PROOF_BYTES_PER_ROW = 16 * 1024  # unshared provers
HBM_BYTES = 288 * 10**9


def scratch_is_shared():
    import os
    # the same reading as the library's (csrc/prover.hip: atoi(value) != 0; unset = shared): "0", "off", "" all mean "not shared"
    v = os.environ.get("MP2G_SHARE_SCRATCH")
    if v is None:
        return True
    m = re.match(r"\s*[+-]?\d+", v)
    return bool(m) and int(m.group(0)) != 0


def plan_rank_resources(shapes, workers, batch, ranks_on_host, host_cpus, hbm_bytes=HBM_BYTES, shared=None):
    """What ONE rank of a node takes for a table / recursion build: `shapes` = {circuit: [log2 rows of the base circuit and of every
    wrap step]} (TableParams.shapes()), `workers` proving threads (GPU streams with a prover set each) holding `batch` proofs in
    flight per circuit chain. Returns the host threads per worker (the node's hardware threads over the workers of all ranks on the
    host: never more threads than the host has), their total over the node, and the device bytes the rank's provers need -- each
    rank has a GPU of its own, so this is compared with ONE GPU's memory. `fits` false = shrink `batch` or `workers`."""
    shared = scratch_is_shared() if shared is None else shared
    rows = sum(1 << int(k) for chain in shapes.values() for k in chain)
    widest = max(1 << int(k) for chain in shapes.values() for k in chain)
    per_proof = (SCRATCH_BYTES_PER_ROW * widest + CHAIN_BYTES_PER_ROW * rows) if shared else PROOF_BYTES_PER_ROW * rows
    device_bytes = int(workers) * int(batch) * per_proof
    workers_on_host = max(1, int(workers) * max(1, int(ranks_on_host)))
    host_threads = max(1, int(host_cpus) // workers_on_host)
    return {"host_threads_per_worker": host_threads, "host_threads_on_node": host_threads * workers_on_host, "worker_threads_on_node": workers_on_host,
            "device_bytes_per_rank": device_bytes, "shared_scratch": bool(shared), "fits": device_bytes <= 0.9 * hbm_bytes}


def shard_range(n_items, rank, world):
    """Contiguous block partition: rank r owns [lo, hi). Sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def subtree_plan(n_leaves, arity, world):
    """Bottom-up work plan of a complete `arity`-ary aggregation tree over n_leaves leaf proofs,
    sharded over `world` ranks (both powers of arity for the levels that cross ranks).

    Returns a list of levels; level l is a list of (node_index, owner_rank, child_indices).
    A node is owned by the rank that owns its first leaf, so everything below the shard
    boundary is rank-local (ryhope's WorkplanItem::Subtree handed to one GPU); above it the
    children of a node live on `arity` different ranks and must be handed to the owner.
    """
    levels = []
    width = n_leaves
    span = 1  # leaves under one node of the current level
    while width > 1:
        assert width % arity == 0, "complete tree expected"
        width //= arity
        span *= arity
        lvl = []
        for node in range(width):
            first_leaf = node * span
            owner = owner_of(first_leaf, n_leaves, world)
            lvl.append((node, owner, [node * arity + k for k in range(arity)]))
        levels.append(lvl)
    return levels


def owner_of(item, n_items, world):
    base, rem = divmod(n_items, world)
    cut = rem * (base + 1)
    if item < cut:
        return item // (base + 1)
    return rem + (item - cut) // base if base else world - 1


def tree_handoff_plan(n_leaves, arity, world):
    """[(level, src_rank, dst_rank, child_node_index)] for every child proof that has to move
    between ranks, in the order the levels are proved."""
    moves = []
    span = 1
    for lvl_idx, lvl in enumerate(subtree_plan(n_leaves, arity, world)):
        for node, owner, children in lvl:
            for c in children:
                src = owner_of(c * span, n_leaves, world)
                if src != owner:
                    moves.append((lvl_idx, src, owner, c))
        span *= arity
    return moves


def all_gather_words(dist, local_words, device=None):
    """all_gather of a small fixed-size uint64 payload (e.g. one point, 20 limbs).
    Returns an array [world][len(local_words)]. uint64 travels as int64 (same bits)."""
    import torch
    a = np.ascontiguousarray(local_words, dtype=np.uint64)
    t = torch.from_numpy(a.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in out])


def exchange_bytes(dist, payload, src, dst, device=None):
    """Point-to-point hand-off of one serialized child proof (bytes) from src to dst; returns the
    payload on dst, None elsewhere. Two messages: length, then body."""
    import torch
    rank = dist.get_rank()
    if rank == src:
        body = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        n = torch.tensor([body.numel()], dtype=torch.int64)
        if device is not None:
            body, n = body.to(device), n.to(device)
        dist.send(n, dst)
        dist.send(body, dst)
        return None
    if rank == dst:
        n = torch.zeros(1, dtype=torch.int64, device=device)
        dist.recv(n, src)
        body = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
        dist.recv(body, src)
        return bytes(body.cpu().numpy())
    return None


class _DevView:
    """a DeviceBuffer range as a __cuda_array_interface__ object (torch.as_tensor wraps it without a copy)"""

    def __init__(self, buf, offset_bytes, n_words):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i8", "data": (buf.ptr.value + offset_bytes, False), "version": 2}
        self._keep = buf


def device_words(buf, offset_bytes, n_words, device):
    """int64 torch view of n_words u64 words of a libmp2gpu device buffer (same bits; RCCL has no u64)"""
    import torch
    return torch.as_tensor(_DevView(buf, offset_bytes, n_words), device=device)


def send_proof_words(dist, parts, dst, device=None):
    """hand one child proof to the rank that proves its parent: `parts` = the prover's output ranges (caps, openings,
    FRI proof words, public inputs) as int64 tensors, device tensors over RCCL (the proof never visits the host) or
    host tensors over gloo. Sizes are fixed by the circuit shape, so there is no length message."""
    for t in parts:
        dist.send(t, dst)


def recv_proof_words(dist, sizes, src, device=None):
    import torch
    out = []
    for n in sizes:
        t = torch.empty(n, dtype=torch.int64, device=device)
        dist.recv(t, src)
        out.append(t)
    return out


class _RawView:
    def __init__(self, ptr, n_words, keep):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
        self._keep = keep


# How root proofs move between ranks above the shard boundary, decided ONCE per run by `probe_handoff` before anything is timed:
#   "device": RCCL reads the prover's own output ranges (libmp2gpu allocations wrapped as tensor views) -- no copy on the sender;
#   "staged": the sender downloads the proof (DeviceProof.to_host), and a torch-allocated tensor travels (RCCL: uploaded first);
#   "host":   gloo -- host tensors (the CPU tests, and MP2G_BENCH_BACKEND=gloo on a box with fewer GPUs than ranks).
HANDOFF = {"mode": None, "reason": None}
PROBE_WORDS = 64


def join_pairs(world):
    """[(level, src, dst)] of the binary join above the shard boundary: at level l the ranks with the low l bits clear meet in pairs,
    the upper one (bit l set) hands its tree's root proof to the lower one (log2(world) point-to-point hand-offs)."""
    out = []
    for lvl in range(max(0, int(world).bit_length() - 1)):
        bit = 1 << lvl
        for r in range(0, world, 2 * bit):
            if r + bit < world:
                out.append((lvl, r + bit, r))
    return out


def probe_pattern(src, marker):
    w = (np.arange(PROBE_WORDS, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64((src + 1) << 40)
    w[0] = marker
    return w


def probe_handoff(dist, direct_tensor, staged_tensor, device=None):
    """Before anything is timed: every pair of `join_pairs` moves one small buffer exactly the way the root proofs will move.
    `direct_tensor(words)` returns the tensor the direct path would send -- under RCCL a view of a buffer ALLOCATED BY libmp2gpu
    holding `words` (the one thing no single-GPU test can exercise) -- or raises; `staged_tensor(words)` returns a torch-allocated
    tensor of the same words (the other path). A sender whose direct send raises satisfies its peer's pending recv with the staged
    tensor (marker word 2 instead of 1), a receiver that finds other words than the pattern reports a failure, and one all_reduce
    (MAX) makes every rank choose the same mode for every level: "device" only if all pairs moved the pattern directly. With
    direct_tensor = None (gloo) the pairs exchange host tensors and the mode is "host". Sets and returns HANDOFF."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    bad, reason = 0, None
    for lvl, src, dst in join_pairs(world):
        if rank == src:
            if direct_tensor is None:
                dist.send(staged_tensor(probe_pattern(src, 3)), dst)
                continue
            try:
                t = direct_tensor(probe_pattern(src, 1))
                dist.send(t, dst)
                if device is not None:
                    torch.cuda.synchronize()
            except Exception as e:  # the peer is waiting in recv for PROBE_WORDS words: the staged tensor answers it, marked
                bad, reason = 1, f"rank {src} -> {dst}: {type(e).__name__}: {e}"[:300]
                dist.send(staged_tensor(probe_pattern(src, 2)), dst)
                if device is not None:
                    torch.cuda.synchronize()
        elif rank == dst:
            t = torch.empty(PROBE_WORDS, dtype=torch.int64, device=device)
            dist.recv(t, src)
            if device is not None:
                torch.cuda.synchronize()
            got = t.cpu().numpy().view(np.uint64)
            marker = int(got[0])
            if marker not in (1, 2, 3) or not np.array_equal(got[1:], probe_pattern(src, marker)[1:]):
                bad, reason = 1, f"rank {src} -> {dst}: the probe arrived with other words than were sent"
            elif marker == 2:
                bad, reason = 1, f"rank {src} -> {dst}: the sender fell back to a staged tensor"
    flag = torch.tensor([bad], dtype=torch.int64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if direct_tensor is None:
        if int(flag.item()):
            raise RuntimeError(f"hand-off probe failed over {dist.get_backend()}: {reason}")
        HANDOFF.update(mode="host", reason=None)
    elif int(flag.item()):
        HANDOFF.update(mode="staged", reason=reason or "another pair's direct send failed")
    else:
        HANDOFF.update(mode="device", reason=None)
    return dict(HANDOFF)


def send_device_proof(dist, ctx, dp, dst, device=None):
    """hand a final proof (recursion.DeviceProof) to the rank that proves its parent. RCCL (device given) and HANDOFF mode "device"
    (or no probe run): the word ranges of the prover's output buffers are sent as they are, device to device; mode "staged": the
    proof is downloaded and a torch-allocated tensor travels; gloo: through host tensors. The context's stream is drained first:
    the collective runs on torch's stream."""
    import torch
    from .recursion import DeviceProof
    ctx.sync()
    if isinstance(dp, DeviceProof) and device is not None and HANDOFF["mode"] != "staged":
        parts = [torch.as_tensor(_RawView(ptr, n, dp.keep), device=device) for ptr, n in dp.parts]
    else:
        # a host proof tuple (the host-witness back end keeps no chain outputs on the device), the staged mode, or gloo: host tensors;
        # with RCCL they go up first -- the receiver sees device tensors either way
        caps, openings, fri, pis = dp.to_host(ctx) if isinstance(dp, DeviceProof) else dp
        parts = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64).ravel().copy()) for a in (pis, np.asarray(caps)[1:4], openings, fri)]
        if device is not None:
            parts = [t.to(device) for t in parts]
    try:
        for t in parts:
            dist.send(t, dst)
        if device is not None:
            torch.cuda.synchronize()
    except Exception as e:  # a failed point-to-point transfer must end the job with the reason, not leave the peers waiting
        raise RuntimeError(f"rank {dist.get_rank()}: sending a root proof to rank {dst} failed ({dist.get_backend()}): {e}") from e


def recv_device_proof(dist, sizes, src, device=None):
    """the receiving side: sizes = words of (public inputs, three caps, openings, FRI proof). Returns a recursion.DeviceProof over
    the received device tensors (RCCL) or the host proof tuple (gloo)."""
    import torch
    from .recursion import DeviceProof
    try:
        ts = recv_proof_words(dist, sizes, src, device)
        if device is not None:
            torch.cuda.synchronize()
    except Exception as e:
        raise RuntimeError(f"rank {dist.get_rank()}: receiving a root proof from rank {src} failed ({dist.get_backend()}): {e}") from e
    if device is not None:
        return DeviceProof([(t.data_ptr(), t.numel()) for t in ts], keep=ts)
    pis, caps3, openings, fri = [t.numpy().view(np.uint64) for t in ts]
    caps = np.concatenate([np.zeros(caps3.size // 3, dtype=np.uint64), caps3]).reshape(4, -1)
    return caps, openings.reshape(-1, 2), fri, pis


def all_gather_bytes(dist, payload, device=None):
    """all_gather of one variable-length byte string per rank (serialized root proofs of a wave):
    lengths first, then bodies padded to the longest. Returns [bytes per rank]."""
    import torch
    world = dist.get_world_size()
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(x.item()) for x in lens]
    cap = max(max(lens), 1)
    body = torch.zeros(cap, dtype=torch.uint8)
    body[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8) if payload else body[:0]
    if device is not None:
        body = body.to(device)
    out = [torch.empty_like(body) for _ in range(world)]
    dist.all_gather(out, body)
    return [bytes(o.cpu().numpy()[:ln]) for o, ln in zip(out, lens)]


def run_workplan(dist, plan, prove_item, device=None):
    """Drive an UpdatePlan (workplan.py) over the ranks of `dist` (None = single process).

    Every rank holds the same plan. Per wave: drain the Ready items, deal them to ranks
    (`workplan.assign_subtrees`, deterministic), hand every item's owner the results it needs from other ranks --
    the roots of the spun-off subtrees (or single nodes) below it, each sent point to point from the rank that
    proved it (`exchange_bytes`: one send/recv pair per child proof, the survey's binary-tree pattern; nothing is
    broadcast) -- then each rank proves its items with `prove_item(item, child_results) -> bytes`: an item is one
    node or one spun-off subtree proved bottom-up locally, only its root result ever leaves the rank. Every rank
    walks the same list of transfers in the same order, so a rank is in at most one transfer at a time and the
    blocking pairs cannot deadlock. Children-before-parents is the plan's own guarantee
    (ryhope/src/storage/updatetree.rs:449-531). Returns {key: result} of the items this rank holds (proved or
    received); the tree root's result is published to every rank at the end (one small all_gather)."""
    from . import workplan as wp
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    parents = plan.tree().parents()
    kids = {}
    for k, p in parents.items():
        kids.setdefault(p, []).append(k)
    root_key = next(k for k, p in parents.items() if p is None)
    results, producer = {}, {}
    while True:
        wave = wp.drain_wave(plan)
        if not wave:
            if not plan.completed():
                raise RuntimeError("work plan stalled: items not marked done")
            break
        owners = wp.assign_subtrees(wave, world)
        for it, o in zip(wave, owners):
            inside = set(it.subtree.nodes()) if it.subtree is not None else {it.k}
            needed = sorted(c for n in inside for c in kids.get(n, []) if c not in inside)
            for c in needed:
                src = producer[c]
                if src != o and dist is not None:
                    got = exchange_bytes(dist, results.get(c, b""), src, o, device)
                    if rank == o:
                        results[c] = got
        for it, o in zip(wave, owners):
            if o == rank:
                results[it.k] = prove_item(it, results)
            producer[it.k] = o
        for it in wave:
            plan.done(it.k)
    if dist is not None:
        parts = all_gather_bytes(dist, results.get(root_key, b"") if producer.get(root_key) == rank else b"", device)
        results[root_key] = parts[producer[root_key]]
    return results
