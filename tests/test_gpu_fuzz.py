"""Seeded random-shape parity sweeps (HIP vs oracle): NTT sizes/batches/modes, Merkle shapes, hash
lengths, whole PCS proofs with random oracle widths. Shapes are drawn from a fixed seed so failures
reproduce; MP2G_FUZZ_CASES scales the sweep (default keeps the GPU suite short)."""
import ctypes
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("MP2G_FUZZ_CASES", "12"))


def test_fuzz_ntt(ctx):
    rng = np.random.default_rng(101)
    for case in range(N_CASES * 2):
        log_n = int(rng.integers(1, 17))
        batch = int(rng.integers(1, 6)) if log_n > 12 else int(rng.integers(1, 40))
        inverse = bool(rng.integers(0, 2))
        coset = int(rng.integers(0, 2)) * O.MULT_GEN
        bitrev = bool(rng.integers(0, 2)) and not (inverse and coset)
        a = O.rand_field((batch, 1 << log_n), 1000 + case)
        want = O.fft(a, inverse=inverse, coset_shift=coset)
        if bitrev:
            want = want[:, O.bitrev_perm(1 << log_n)]
        got = ctx.ntt(a, inverse=inverse, coset_shift=coset, bitrev_out=bitrev)
        assert np.array_equal(got, want), (log_n, batch, inverse, coset, bitrev)


def test_fuzz_merkle_and_hash(ctx, mp2):
    rng = np.random.default_rng(202)
    for case in range(N_CASES):
        log_l = int(rng.integers(0, 11))
        leaf_len = int(rng.integers(1, 140))
        cap_h = int(rng.integers(0, min(log_l, 5) + 1))
        variant = int(rng.integers(0, 2))
        leaves = O.rand_field((1 << log_l, leaf_len), 2000 + case)
        t = mp2.MerkleTree(ctx, leaves, cap_h, variant)
        lv = O.merkle_build(leaves, cap_h, variant)
        assert np.array_equal(t.cap, O.merkle_cap(lv, cap_h)), (log_l, leaf_len, cap_h, variant)
        t.free()
        in_len, out_len = int(rng.integers(0, 60)), int(rng.integers(1, 13))
        x = O.rand_field((17, in_len), 3000 + case) if in_len else np.zeros((17, 0), dtype=np.uint64)
        assert np.array_equal(ctx.hash_no_pad_batch(x, out_len, variant), O.hash_no_pad_batch(x, out_len, variant))


def test_fuzz_pcs_prove(ctx, mp2):
    rng = np.random.default_rng(303)
    for case in range(max(3, N_CASES // 3)):
        log_n = int(rng.integers(2, 10))
        ws = tuple(int(x) for x in rng.integers(1, 12, size=4))
        zs_count = int(rng.integers(0, ws[2] + 1))
        variant = int(rng.integers(0, 2))
        ofp = O.standard_params(log_n, ws, variant=variant, pow_bits=int(rng.integers(0, 9)), num_queries=int(rng.integers(1, 6)),
                                zs_count=zs_count, cap_height=int(rng.integers(0, 4)))
        fp = mp2.FriParams()
        ctypes.memmove(ctypes.byref(fp), ctypes.byref(ofp), ctypes.sizeof(fp))
        vals = [O.rand_field((w, 1 << log_n), 4000 + 10 * case + i) for i, w in enumerate(ws)]
        cd, ph = O.rand_field(4, 5000 + case), O.rand_field(4, 6000 + case)
        caps, openings, proof = mp2.pcs_prove(ctx, fp, vals, cd, ph)
        oc, oo, op = O.pcs_prove(ofp, vals, cd, ph)
        assert np.array_equal(caps, oc) and np.array_equal(openings, oo) and np.array_equal(proof, op), (log_n, ws, zs_count, variant)
        assert O.pcs_verify(ofp, cd, ph, caps, openings, proof) == 0


def test_fuzz_curve(ctx, mp2):
    rng = np.random.default_rng(404)
    for case in range(max(3, N_CASES // 3)):
        rows, n_cols, n_unique = int(rng.integers(0, 70)), int(rng.integers(1, 6)), int(rng.integers(0, 3))
        col_ids = O.rand_field(n_cols, 7000 + case)
        values = rng.integers(0, 1 << 32, size=(rows, n_cols, 8), dtype=np.uint32)
        unique = rng.integers(0, 1 << 32, size=(rows, n_unique, 8), dtype=np.uint32)
        w, wei = mp2.compute_table_row_digest(ctx, col_ids, values, unique)
        ow, owei = np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
        O.lib().orc_row_digest_batch(0, O.p(col_ids), O.sz(n_cols), O.p(O.arr(values, np.uint32)), O.p(O.arr(unique, np.uint32)),
                                     O.sz(n_unique), O.sz(rows), O.p(ow), O.p(owei))
        assert np.array_equal(w, ow) and np.array_equal(wei, owei), (rows, n_cols, n_unique)
