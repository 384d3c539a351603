"""What eight ranks on ONE host take (the driver's 8-GPU scaling run): sharding.plan_rank_resources sizes the host threads and the
provers' device memory per rank; bench.py's TableRig uses it and refuses a configuration that cannot fit. No GPU needed."""
import importlib

S = importlib.import_module("mapreduce-plonky2_amd.sharding")

# table.TableParams.shapes() of the default build (tests/test_table.py pins them on the oracle prover)
SHAPES = {"cells_leaf": [6, 12], "cells_full": [13, 12], "cells_partial": [12, 12], "cells_empty": [6, 12], "row_leaf": [12, 12], "row_full": [14, 13, 12],
          "row_partial": [13, 12]}


def test_eight_ranks_on_one_host_fit():
    """bench.py's defaults (4 workers x 32 proofs in flight) with 8 ranks on a 256-thread host: no more host threads than the host
    has, every rank's provers inside its own GPU's 288 GB"""
    for cpus in (256, 192, 64, 8):
        p = S.plan_rank_resources(SHAPES, workers=4, batch=32, ranks_on_host=8, host_cpus=cpus, shared=False)
        assert p["worker_threads_on_node"] == 32
        assert p["host_threads_per_worker"] == max(1, cpus // 32)
        assert p["host_threads_on_node"] <= max(cpus, 32)
        assert p["fits"] and p["device_bytes_per_rank"] < 0.9 * S.HBM_BYTES
    one = S.plan_rank_resources(SHAPES, 4, 32, 1, 256, shared=False)
    assert one["host_threads_per_worker"] == 64 and one["device_bytes_per_rank"] == S.plan_rank_resources(SHAPES, 4, 32, 8, 256, shared=False)["device_bytes_per_rank"]


def test_shared_scratch_is_sized_by_the_widest_step():
    """with the provers of a context sharing one scratch (csrc/ctx.h, the default) a rank pays the working set of ONE prove() per proof
    in flight -- the widest circuit step -- plus every step's hand-over buffers; without it (MP2G_SHARE_SCRATCH=0) the working set of
    every step of every circuit: 4 x 48 proofs in flight take a quarter of the GPU instead of most of it"""
    own = S.plan_rank_resources(SHAPES, 4, 48, 1, 256, shared=False)
    shared = S.plan_rank_resources(SHAPES, 4, 48, 1, 256, shared=True)
    assert own["fits"] and 0.8 * S.HBM_BYTES < own["device_bytes_per_rank"] < 0.9 * S.HBM_BYTES
    assert shared["fits"] and shared["device_bytes_per_rank"] < 0.35 * own["device_bytes_per_rank"]
    assert S.plan_rank_resources(SHAPES, 4, 128, 1, 256, shared=True)["fits"] and not S.plan_rank_resources(SHAPES, 4, 128, 1, 256, shared=False)["fits"]


def test_the_degree_sweep_keeps_its_memory_constant():
    """by_base_degree halves the proofs in flight per degree step (bench.py: batch >> (k - 12)): the device bytes stay within a factor
    of two of the k = 12 build and fit; the full batch at k = 15 would not"""
    def shapes(k):
        wraps = {12: [12], 13: [12], 14: [13, 12], 15: [13, 12]}[k]
        return {name: [max(k, ch[0])] + (wraps if max(k, ch[0]) == k else ch[1:]) for name, ch in SHAPES.items()}
    base = S.plan_rank_resources(shapes(12), 4, 32, 1, 256, shared=False)
    for k in (12, 13, 14, 15):
        p = S.plan_rank_resources(shapes(k), 4, max(4, 32 >> (k - 12)), 1, 256, shared=False)
        assert p["fits"] and p["device_bytes_per_rank"] <= 2 * base["device_bytes_per_rank"]
    assert not S.plan_rank_resources(shapes(15), 4, 32, 1, 256, shared=False)["fits"]
    # with the shared scratch the full batch fits at every degree
    assert all(S.plan_rank_resources(shapes(k), 4, 48, 1, 256, shared=True)["fits"] for k in (12, 13, 14, 15))
