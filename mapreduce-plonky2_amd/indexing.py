"""Off-circuit Merkle payload hashes of the cells / rows trees, batched per tree level on the GPU.

Mirrors mp2-v1/src/indexing/cell.rs:120-157 (`MerkleCell::aggregate`) and row.rs:257-317
(`RowPayload::aggregate`): the reference hashes one node per call while walking a ryhope work plan;
here all nodes of a level go through one `mp2g_hash_no_pad_batch` launch. Packing only -- the
arithmetic is the HIP sponge kernel.
"""
import importlib

import numpy as np

_mp2 = importlib.import_module(__name__.rsplit(".", 1)[0])


def empty_poseidon_hash(ctx, variant=0):
    """mp2-common/src/poseidon.rs:44-46: H::hash_no_pad(&[])  (= all-zero state squeezed)."""
    return ctx.hash_no_pad_batch(np.zeros((1, 0), dtype=np.uint64), 4, variant)[0]


def u256_limbs(values):
    """[count] python ints -> [count][8] field limbs (u256.rs:870-877: 8 big-endian u32 words)."""
    return _mp2.u256_to_limbs(values).astype(np.uint64)


def cell_node_hashes(ctx, left, right, ids, values, variant=0):
    """H(H(left) || H(right) || id || value): 4 + 4 + 1 + 8 = 17 limbs per node.
    left/right: [count][4] child hashes (use empty_poseidon_hash for a missing child)."""
    left, right = np.asarray(left, dtype=np.uint64), np.asarray(right, dtype=np.uint64)
    ids = np.asarray(ids, dtype=np.uint64).reshape(-1, 1)
    inputs = np.concatenate([left, right, ids, u256_limbs(values)], axis=1)
    assert inputs.shape[1] == 17
    return ctx.hash_no_pad_batch(inputs, 4, variant)


def row_node_hashes(ctx, left, right, mins, maxs, ids, values, cells_root, variant=0):
    """H(hL || hR || min || max || id || value || cells_root): 4+4+8+8+1+8+4 = 37 limbs per node."""
    left, right = np.asarray(left, dtype=np.uint64), np.asarray(right, dtype=np.uint64)
    ids = np.asarray(ids, dtype=np.uint64).reshape(-1, 1)
    inputs = np.concatenate([left, right, u256_limbs(mins), u256_limbs(maxs), ids, u256_limbs(values),
                             np.asarray(cells_root, dtype=np.uint64)], axis=1)
    assert inputs.shape[1] == 37
    return ctx.hash_no_pad_batch(inputs, 4, variant)


def cells_tree_root(ctx, ids, values, variant=0):
    """Root hash of a complete binary cells tree in heap order (node i has children 2i+1, 2i+2,
    missing children hash as empty): bottom-up, one batched launch per level."""
    n = len(ids)
    empty = empty_poseidon_hash(ctx, variant)
    hashes = np.tile(empty, (n, 1))
    depth = max(1, (n).bit_length())
    for level in range(depth - 1, -1, -1):
        lo, hi = (1 << level) - 1, min(n, (1 << (level + 1)) - 1)
        if lo >= hi:
            continue
        idx = np.arange(lo, hi)
        lch, rch = 2 * idx + 1, 2 * idx + 2
        left = np.where((lch < n)[:, None], hashes[np.minimum(lch, n - 1)], empty)
        right = np.where((rch < n)[:, None], hashes[np.minimum(rch, n - 1)], empty)
        hashes[idx] = cell_node_hashes(ctx, left, right, [ids[i] for i in idx], [values[i] for i in idx], variant)
    return hashes[0], hashes


def index_node_hashes(ctx, left, right, mins, maxs, ids, values, row_tree_hash, variant=0):
    """`IndexNode::aggregate` (mp2-v1/src/indexing/index.rs:61-101), the block / primary-index tree:
    H(hL || hR || min || max || id || value || row_tree_hash), the same 37-limb layout as a row node with the
    row tree's root hash in the place of the cells root. min / max follow the children as in the reference:
    no child -> (value, value); left only -> (left.min, value); both -> (left.min, right.max)."""
    return row_node_hashes(ctx, left, right, mins, maxs, ids, values, row_tree_hash, variant)


def index_node_min_max(value, left=None, right=None):
    """min / max bookkeeping of IndexNode::aggregate; left / right are (min, max) tuples or None"""
    if left is None and right is None:
        return value, value
    if right is None:
        return left[0], value
    if left is None:
        raise ValueError("ryhope sbbst is wrong")  # the reference panics: a right child without a left one
    return left[0], right[1]
