"""Cells / rows tree payload hashes through the HIP sponge vs the oracle."""
import importlib

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def limbs(v):
    return [(int(v) >> (32 * (7 - j))) & 0xFFFFFFFF for j in range(8)]


def test_cell_and_row_node_hashes(ctx, mp2):
    ix = importlib.import_module("mapreduce-plonky2_amd.indexing")
    rng = np.random.default_rng(4)
    cnt = 50
    left, right, root = O.rand_field((cnt, 4), 1), O.rand_field((cnt, 4), 2), O.rand_field((cnt, 4), 3)
    ids = O.rand_field(cnt, 4)
    vals = [int.from_bytes(rng.bytes(32), "big") for _ in range(cnt)]
    mins = [int.from_bytes(rng.bytes(32), "big") for _ in range(cnt)]
    maxs = [int.from_bytes(rng.bytes(32), "big") for _ in range(cnt)]
    got = ix.cell_node_hashes(ctx, left, right, ids, vals)
    got_row = ix.row_node_hashes(ctx, left, right, mins, maxs, ids, vals, root)
    for i in range(cnt):
        inp = list(left[i]) + list(right[i]) + [ids[i]] + limbs(vals[i])
        assert np.array_equal(got[i], O.hash_n_to_m_no_pad(inp, 4))
        inp = list(left[i]) + list(right[i]) + limbs(mins[i]) + limbs(maxs[i]) + [ids[i]] + limbs(vals[i]) + list(root[i])
        assert len(inp) == 37
        assert np.array_equal(got_row[i], O.hash_n_to_m_no_pad(inp, 4))


def test_cells_tree_root_and_empty_hash(ctx, mp2):
    ix = importlib.import_module("mapreduce-plonky2_amd.indexing")
    empty = ix.empty_poseidon_hash(ctx)
    assert not empty.any()  # hash_no_pad(&[]) squeezes the all-zero state
    ids = [int(x) for x in O.rand_field(5, 9)]
    vals = [3, 2 ** 200 + 7, 0, 2 ** 256 - 1, 12345]
    root, hashes = ix.cells_tree_root(ctx, ids, vals)

    def node(i):
        if i >= 5:
            return [0, 0, 0, 0]
        return [int(x) for x in O.hash_n_to_m_no_pad(node(2 * i + 1) + node(2 * i + 2) + [ids[i]] + limbs(vals[i]), 4)]
    assert [int(x) for x in root] == node(0)


def test_column_identifiers(ctx, mp2):
    """values_extraction column ids over the C ABI vs the oracle sponge on the same byte strings"""
    import importlib
    ids = importlib.import_module("mapreduce-plonky2_amd.identifiers")
    addr, chain = bytes(range(1, 21)), 31337

    def want(data, variant):
        return int(O.hash_n_to_m_no_pad(np.frombuffer(data, dtype=np.uint8).astype(np.uint64), 4, variant)[0])

    for variant in (0, 1):
        assert ids.identifier_block_column(ctx, variant) == want(b"BLOCK_NUMBER", variant)
        assert ids.identifier_offchain_column(ctx, "t", "col", variant) == want(b"OFFCHAIN_TABLEtcol", variant)
        extra = addr + chain.to_bytes(8, "big") + b"x"
        assert ids.identifier_for_value_column(ctx, 3, 4, 128, 1, addr, chain, b"x", variant) == \
            want(bytes([3]) + (4).to_bytes(8, "big") + (128).to_bytes(8, "big") + (1).to_bytes(4, "big") + extra, variant)
        assert ids.identifier_for_mapping_key_column(ctx, 7, addr, chain, b"x", variant) == want(b"\0KEY" + bytes([7]) + extra, variant)
        assert ids.identifier_for_outer_mapping_key_column(ctx, 7, addr, chain, b"x", variant) == want(b"\0OUT_KEY" + bytes([7]) + extra, variant)
        assert ids.identifier_for_inner_mapping_key_column(ctx, 7, addr, chain, b"x", variant) == want(b"\0\0IN_KEY" + bytes([7]) + extra, variant)
    assert ids.identifier_block_column(ctx, 0) != ids.identifier_block_column(ctx, 1)
    # a whole table's columns in one batch (mixed lengths are grouped per launch, order kept)
    slots = [(s, o, ln, w) for s in range(6) for (o, ln, w) in ((0, 256, 0), (12, 32, 1))]
    got = ids.table_column_identifiers(ctx, slots, addr, chain)
    assert got == [ids.identifier_for_value_column(ctx, s, o, ln, w, addr, chain) for (s, o, ln, w) in slots]
    mixed = [b"BLOCK_NUMBER", b"x" * 40, b"OFFCHAIN_TABLEtcol", b"y" * 40]
    assert ids.identifiers_batch(ctx, mixed) == [want(m, 0) for m in mixed]


def test_index_node_hash(ctx, mp2):
    """IndexNode::aggregate: 37 limbs, row tree root hash last; min / max rules"""
    ix = importlib.import_module("mapreduce-plonky2_amd.indexing")
    empty = ix.empty_poseidon_hash(ctx)
    left = O.rand_field((1, 4), 1)
    rth = O.rand_field((1, 4), 2)
    value, lmin = 2228671, 17
    mn, mx = ix.index_node_min_max(value, left=(lmin, 99))
    assert (mn, mx) == (lmin, value) and ix.index_node_min_max(value) == (value, value)
    assert ix.index_node_min_max(5, (1, 2), (7, 9)) == (1, 9)
    with pytest.raises(ValueError):
        ix.index_node_min_max(5, None, (7, 9))
    got = ix.index_node_hashes(ctx, left, empty[None], [mn], [mx], [77], [value], rth)[0]
    limbs = np.concatenate([left[0], empty, mp2.u256_to_limbs([mn])[0].astype(np.uint64), mp2.u256_to_limbs([mx])[0].astype(np.uint64),
                            [np.uint64(77)], mp2.u256_to_limbs([value])[0].astype(np.uint64), rth[0]])
    assert limbs.size == 37 and np.array_equal(got, O.hash_n_to_m_no_pad(limbs, 4))
