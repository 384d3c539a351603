"""HIP Ecgfp5 batch kernels vs the CPU oracle and the reference's SSWU known-answer tests
(SURVEY 8 rows a8-a11)."""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def o_map(ins, variant=0):
    ins = O.arr(ins)
    w = np.zeros((ins.shape[0], 5), dtype=np.uint64)
    wei = np.zeros((ins.shape[0], 11), dtype=np.uint64)
    O.lib().orc_map_to_curve_batch(variant, O.p(ins), O.sz(ins.shape[1]), O.sz(ins.shape[0]), O.p(w), O.p(wei))
    return w, wei


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("in_len", [1, 5, 9, 17])
def test_map_to_curve_batch(ctx, mp2, variant, in_len):
    ins = O.rand_field((200, in_len), 0xC0FFEE04 + in_len)
    w, wei = mp2.map_to_curve_batch(ctx, ins, variant, weierstrass=True)
    ow, owei = o_map(ins, variant)
    assert np.array_equal(w, ow)
    assert np.array_equal(wei, owei)


def test_swu_kats_through_hip_decode(ctx, mp2):
    """sswu_value.rs:88-118: the KAT outputs are valid encodings; summing a single decoded point
    returns the same encoding (decode -> encode round trip on the device)."""
    kat = json.load(open(os.path.join(G, "sswu_kat.json")))
    for v in kat["vectors"]:
        w = mp2.curve_sum(ctx, [v["output"]])
        assert [int(x) for x in w] == v["output"]


def test_curve_sum_and_edge_cases(ctx, mp2):
    ins = O.rand_field((300, 9), 7)
    w = mp2.map_to_curve_batch(ctx, ins)
    for count in (1, 2, 3, 127, 128, 129, 300):
        want = np.zeros(5, dtype=np.uint64)
        want_wei = np.zeros(11, dtype=np.uint64)
        assert O.lib().orc_curve_sum(O.p(O.arr(w[:count])), O.sz(count), O.p(want), O.p(want_wei))
        got, got_wei = mp2.curve_sum(ctx, w[:count], weierstrass=True)
        assert np.array_equal(got, want) and np.array_equal(got_wei, want_wei)
    # empty sum and the neutral element (encoding 0, Weierstrass is_inf = 1)
    got, got_wei = mp2.curve_sum(ctx, np.zeros((0, 5), dtype=np.uint64), weierstrass=True)
    assert not got.any() and int(got_wei[10]) == 1 and not got_wei[:10].any()
    got = mp2.curve_sum(ctx, np.zeros((3, 5), dtype=np.uint64))
    assert not got.any()
    # P + (-P) = neutral: -P has encoding -w
    neg = (np.uint64(O.P) - w[0]) % np.uint64(O.P)
    assert not mp2.curve_sum(ctx, np.stack([w[0], neg])).any()
    # invalid encoding is rejected (w = 1: delta is not a square for this curve unless decode says so)
    bad = None
    for cand in range(1, 50):
        e = O.arr([cand, 0, 0, 0, 0])
        if not O.lib().orc_decode_check(O.p(e)):
            bad = e
            break
    assert bad is not None
    with pytest.raises(mp2.Mp2gError):
        mp2.curve_sum(ctx, bad.reshape(1, 5))


def test_scalar_mul_batch(ctx, mp2):
    ins = O.rand_field((40, 9), 11)
    w = mp2.map_to_curve_batch(ctx, ins)
    rng = np.random.default_rng(3)
    scalars = [int.from_bytes(rng.bytes(16), "little") for _ in range(40)]
    scalars[0], scalars[1], scalars[2] = 0, 1, (1 << 128) - 1
    # the digit boundaries of the signed 4-bit windows (csrc/ecgfp5.hip pt_mul128): all digits 8 (the largest positive one, no carry),
    # all 9 (every digit negative, a carry into every next one and into the 33rd), a carry chain through 7s that ends in an 8,
    # single digits at either end, 2^127 (top digit 8), and the neutral point as the base
    edge = [int("8" * 32, 16), int("9" * 32, 16), int("7" * 31 + "9", 16), int("f" * 31 + "8", 16), 8, 9, 15, 16, 1 << 127, (1 << 127) + 8, 0xF << 124, 0x8 << 124 | 0x8]
    scalars[3:3 + len(edge)] = edge
    w[20] = 0  # Point::NEUTRAL encodes to zero
    got = mp2.scalar_mul_batch(ctx, w, scalars)
    assert not got[20].any()
    for i in range(40):
        kl = O.arr([(scalars[i] >> (32 * j)) & 0xFFFFFFFF for j in range(4)], np.uint32)
        want = np.zeros(5, dtype=np.uint64)
        assert O.lib().orc_scalar_mul(O.p(w[i]), O.p(kl), 4, O.p(want), None)
        assert np.array_equal(got[i], want), i
    assert not got[0].any() and np.array_equal(got[1], w[1])


@pytest.mark.parametrize("variant", [0, 1])
def test_field_hashed_scalar_mul(ctx, mp2, variant):
    base = mp2.map_to_curve_batch(ctx, O.rand_field((1, 9), 5), variant)[0]
    inputs = O.rand_field(11, 6)
    w, wei = mp2.field_hashed_scalar_mul(ctx, inputs, base, variant)
    ow, owei = np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
    assert O.lib().orc_field_hashed_scalar_mul(variant, O.p(inputs), O.sz(11), O.p(base), O.p(ow), O.p(owei))
    assert np.array_equal(w, ow) and np.array_equal(wei, owei)


@pytest.mark.parametrize("rows,n_cols,n_unique", [(1, 1, 1), (7, 4, 1), (130, 4, 2), (300, 5, 0), (0, 3, 1)])
def test_compute_table_row_digest(ctx, mp2, rows, n_cols, n_unique):
    rng = np.random.default_rng(rows + n_cols)
    col_ids = O.rand_field(n_cols, 0xC0FFEE04)
    values = rng.integers(0, 1 << 32, size=(rows, n_cols, 8), dtype=np.uint32)
    unique = rng.integers(0, 1 << 32, size=(rows, n_unique, 8), dtype=np.uint32)
    if rows > 3:
        values[1] = 0
        values[2] = 0xFFFFFFFF
    w, wei = mp2.compute_table_row_digest(ctx, col_ids, values, unique)
    ow, owei = np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
    O.lib().orc_row_digest_batch(0, O.p(col_ids), O.sz(n_cols), O.p(O.arr(values, np.uint32)), O.p(O.arr(unique, np.uint32)),
                                 O.sz(n_unique), O.sz(rows), O.p(ow), O.p(owei))
    assert np.array_equal(w, ow) and np.array_equal(wei, owei)


def test_cell_values_digest_is_additive(ctx, mp2):
    """verifiable-db/src/cells_tree/mod.rs:65-72: D(id || value limbs); the multiset digest of a
    row is the sum of its cells' digests in any order."""
    ids = O.rand_field(4, 1)
    vals = mp2.u256_to_limbs([1, 2 ** 255 + 12345, 2 ** 256 - 1, 0])
    ins = np.concatenate([ids.reshape(4, 1), vals.astype(np.uint64)], axis=1)
    w = mp2.map_to_curve_batch(ctx, ins)
    assert np.array_equal(mp2.curve_sum(ctx, w), mp2.curve_sum(ctx, w[::-1]))


def test_split_digest_point(ctx, mp2):
    """mp2-common/src/digest.rs SplitDigestPoint over the C ABI vs the same composition of oracle primitives."""
    import importlib
    dg = importlib.import_module("mapreduce-plonky2_amd.digest")
    L = O.lib()
    pts = mp2.map_to_curve_batch(ctx, O.rand_field((4, 9), 321))

    def o_sum(ws):
        w, wei = np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
        assert L.orc_curve_sum(O.p(O.arr(np.stack(ws))), O.sz(len(ws)), O.p(w), O.p(wei))
        return w, wei

    def o_map(fields):
        w = np.zeros((1, 5), dtype=np.uint64)
        L.orc_map_to_curve_batch(0, O.p(O.arr(fields).reshape(1, -1)), O.sz(len(fields)), O.sz(1), O.p(w), None)
        return w[0]

    def o_hashed_mul(inputs, base):
        w, wei = np.zeros(5, dtype=np.uint64), np.zeros(11, dtype=np.uint64)
        assert L.orc_field_hashed_scalar_mul(0, O.p(O.arr(inputs)), O.sz(len(inputs)), O.p(O.arr(base)), O.p(w), O.p(wei))
        return w

    a = dg.SplitDigestPoint.from_single_digest_point(ctx, pts[0], False)
    b = dg.SplitDigestPoint.from_single_digest_point(ctx, pts[1], True)
    c = dg.SplitDigestPoint.from_single_digest_point(ctx, pts[2], False)
    d = dg.SplitDigestPoint.from_single_digest_point(ctx, pts[3], True)
    assert not a.is_merge_case() and b.is_merge_case()
    # simple case: no multiplier -> the row digest is map_to_curve(individual.to_fields())
    ac = a.accumulate(c)
    assert np.array_equal(ac.individual, o_sum([pts[0], pts[2]])[0]) and not ac.multiplier.any()
    assert np.array_equal(ac.cond_combine_to_row_digest(), o_map(o_sum([pts[0], pts[2]])[1]))
    # merge case: HashToInt(map(multiplier).to_fields()) * map(individual)
    s = a.accumulate(b).accumulate(c).accumulate(d)
    ind, ind_wei = o_sum([pts[0], pts[2]])
    mul, mul_wei = o_sum([pts[1], pts[3]])
    assert np.array_equal(s.individual, ind) and np.array_equal(s.multiplier, mul) and s.is_merge_case()
    base, mult = o_map(ind_wei), o_map(mul_wei)
    want = o_hashed_mul(o_sum([mult])[1], base)
    assert np.array_equal(s.cond_combine_to_row_digest(), want)
    assert np.array_equal(s.combine_to_row_digest(), o_hashed_mul(mul_wei, ind))
    # accumulation is commutative and associative
    t = d.accumulate(c).accumulate(b).accumulate(a)
    assert np.array_equal(t.individual, s.individual) and np.array_equal(t.multiplier, s.multiplier)
    # NEUTRAL is the identity of accumulate
    e = dg.SplitDigestPoint(ctx)
    assert np.array_equal(e.accumulate(a).individual, pts[0]) and not e.accumulate(a).multiplier.any()


def test_row_id_helpers_agree_with_the_fused_digest(ctx, mp2):
    """row_unique_data / compute_row_id / Cell::values_digest composed on the host reproduce the fused
    compute_table_row_digest kernel for a single row; compute_index_digest matches the oracle."""
    import importlib
    dg = importlib.import_module("mapreduce-plonky2_amd.digest")
    rng = np.random.default_rng(5)
    n_cols = 3
    col_ids = O.rand_field(n_cols, 41)
    values = rng.integers(0, 1 << 32, size=(1, n_cols, 8), dtype=np.uint32)
    unique = values[:, :2, :].copy()
    fused, _ = mp2.compute_table_row_digest(ctx, col_ids, values, unique)
    # sum over the cells of map_to_curve(id || value limbs), times the row id
    cells = np.concatenate([col_ids.reshape(n_cols, 1), values[0].astype(np.uint64)], axis=1)
    row_digest = mp2.curve_sum(ctx, mp2.map_to_curve_batch(ctx, cells))
    uh = dg.row_unique_data(ctx, unique[0])
    row_id = dg.compute_row_id(ctx, uh, n_cols)
    assert row_id < 1 << 128
    assert np.array_equal(mp2.scalar_mul_batch(ctx, row_digest.reshape(1, 5), [row_id])[0], fused)
    # index digest
    inputs = np.concatenate([[np.uint64(77)], values[0, 0].astype(np.uint64)])
    w = np.zeros(5, dtype=np.uint64)
    assert O.lib().orc_field_hashed_scalar_mul(0, O.p(O.arr(inputs)), O.sz(inputs.size), O.p(O.arr(fused)), O.p(w), None)
    assert np.array_equal(dg.add_primary_index_to_digest(ctx, 77, values[0, 0], fused), w)


def _o_commitment(primary_id, primary, col_ids, values, unique, old, variant=0):
    out = np.zeros(32, dtype=np.uint8)
    oldb = np.frombuffer(old, dtype=np.uint8).copy() if old is not None else None
    O.lib().orc_update_off_chain_data_commitment(variant, ctypes.c_uint64(primary_id), O.p(O.arr(primary, np.uint32)), O.p(O.arr(col_ids)), O.sz(len(col_ids)),
                                                 O.p(O.arr(values, np.uint32)), O.p(O.arr(unique, np.uint32)), O.sz(unique.shape[1]), O.sz(values.shape[0]),
                                                 O.p(oldb) if oldb is not None else None, O.p(out))
    return out.tobytes()


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("rows,n_groups,n_cols,uniq", [(1, 1, 1, (0,)), (12, 3, 4, (1,)), (200, 17, 5, (0, 2)), (9, 9, 2, ()), (0, 0, 3, (0,))])
def test_update_off_chain_data_commitment(ctx, mp2, variant, rows, n_groups, n_cols, uniq):
    """mp2-v1/src/api.rs:556-603 over the batched kernels (digest.update_off_chain_data_commitment) against the oracle's restatement
    (orc_update_off_chain_data_commitment: group by group, one compute_table_row_digest each): rows scattered over n_groups primary
    values in shuffled order, with and without an old commitment; the update is incremental (committing groups below a cut, then
    the rest, equals committing everything) and does not depend on the order the rows are given in."""
    import importlib
    dg = importlib.import_module("mapreduce-plonky2_amd.digest")
    rng = np.random.default_rng(1000 * rows + n_groups)
    col_ids = O.rand_field(n_cols, 0xC0FFEE04 + n_cols)
    primary_id = int(O.rand_field(1, 7)[0])
    values = rng.integers(0, 1 << 32, size=(rows, n_cols, 8), dtype=np.uint32)
    gvals = rng.integers(0, 1 << 32, size=(max(1, n_groups), 8), dtype=np.uint32)
    if n_groups > 2:
        gvals[1, :7] = gvals[0, :7]  # two primaries that differ in the least significant word only
    which = np.concatenate([np.arange(n_groups), rng.integers(0, max(1, n_groups), size=max(0, rows - n_groups))])[:rows].astype(np.int64)
    rng.shuffle(which)
    primary = gvals[which].reshape(rows, 8)
    ucols = [col_ids[i] for i in uniq]
    unique = np.ascontiguousarray(values[:, list(uniq), :]) if uniq else np.zeros((rows, 0, 8), dtype=np.uint32)
    old = bytes(rng.integers(0, 256, size=32, dtype=np.uint8))
    for oc in (None, old):
        got = dg.update_off_chain_data_commitment(ctx, primary_id, primary, col_ids, values, ucols, oc, variant)
        assert len(got) == 32 and got == _o_commitment(primary_id, primary, col_ids, values, unique, oc, variant)
    if rows == 0:
        assert dg.update_off_chain_data_commitment(ctx, primary_id, primary, col_ids, values, ucols, old, variant) == old
        return
    assert dg.off_chain_data_commitment(ctx, primary_id, primary, col_ids, values, ucols, variant) == _o_commitment(primary_id, primary, col_ids, values, unique, None, variant)
    # row order does not matter
    perm = rng.permutation(rows)
    assert dg.update_off_chain_data_commitment(ctx, primary_id, primary[perm], col_ids, values[perm], ucols, old, variant) == got
    # incremental: the groups below a cut first, then the others on top of that commitment
    ints = [sum(int(x) << (32 * (7 - j)) for j, x in enumerate(pv)) for pv in primary]
    cut = sorted(set(ints))[len(set(ints)) // 2]
    lo = np.array([v < cut for v in ints])
    if lo.any() and (~lo).any():
        first = dg.update_off_chain_data_commitment(ctx, primary_id, primary[lo], col_ids, values[lo], ucols, old, variant)
        assert dg.update_off_chain_data_commitment(ctx, primary_id, primary[~lo], col_ids, values[~lo], ucols, first, variant) == got
    with pytest.raises(ValueError):
        dg.update_off_chain_data_commitment(ctx, primary_id, primary, col_ids, values, [int(col_ids[0]) ^ 1], None, variant)
