"""Off-circuit digest bookkeeping of the reference over the C ABI: host mirror of mp2-common/src/digest.rs
(`SplitDigestPoint`, lines 19-55) and of the conditional hashed scalar multiplication of
mp2-common/src/group_hashing/mod.rs:220-234. Points travel as their 5-limb Ecgfp5 encodings
(`Point::encode`); NEUTRAL encodes as five zeros. Every group operation runs in libmp2gpu."""
import numpy as np

from . import POSEIDON2, curve_sum, curve_sum_ranges, field_hashed_scalar_mul, map_to_curve_batch, row_digests, scalar_mul_batch

NEUTRAL = np.zeros(5, dtype=np.uint64)


def point_to_fields(ctx, w):
    """`Point::to_fields` = the 11 Weierstrass limbs x[5] || y[5] || is_inf (group_hashing/mod.rs:163-180)."""
    return curve_sum(ctx, np.asarray(w, dtype=np.uint64).reshape(1, 5), weierstrass=True)[1]


def add_curve_point(ctx, a, b):
    """curve_add.rs:17-33 off-circuit: a + b"""
    return curve_sum(ctx, np.stack([np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)]))


def cond_field_hashed_scalar_mul(ctx, cond, mul_w, base_w, variant=POSEIDON2):
    """group_hashing/mod.rs:228-234: HashToInt(mul.to_fields()) * base when cond, else base"""
    if not cond:
        return np.asarray(base_w, dtype=np.uint64).copy()
    return field_hashed_scalar_mul(ctx, point_to_fields(ctx, mul_w), base_w, variant)[0]


class SplitDigestPoint:
    """digest.rs:19-55: an `individual` and a `multiplier` accumulator."""

    def __init__(self, ctx, individual=NEUTRAL, multiplier=NEUTRAL, variant=POSEIDON2):
        self.ctx, self.variant = ctx, variant
        self.individual = np.asarray(individual, dtype=np.uint64).copy()
        self.multiplier = np.asarray(multiplier, dtype=np.uint64).copy()

    @classmethod
    def from_single_digest_point(cls, ctx, digest, is_multiplier, variant=POSEIDON2):
        return cls(ctx, NEUTRAL, digest, variant) if is_multiplier else cls(ctx, digest, NEUTRAL, variant)

    def accumulate(self, other):
        return SplitDigestPoint(self.ctx, add_curve_point(self.ctx, other.individual, self.individual),
                                add_curve_point(self.ctx, other.multiplier, self.multiplier), self.variant)

    def is_merge_case(self):
        return bool(self.multiplier.any())

    def cond_combine_to_row_digest(self):
        pts = np.stack([point_to_fields(self.ctx, self.individual), point_to_fields(self.ctx, self.multiplier)])
        base, mult = map_to_curve_batch(self.ctx, pts, self.variant)
        return cond_field_hashed_scalar_mul(self.ctx, self.is_merge_case(), mult, base, self.variant)

    def combine_to_row_digest(self):
        return field_hashed_scalar_mul(self.ctx, point_to_fields(self.ctx, self.multiplier), self.individual, self.variant)[0]


# ---- the scalar side of the table digest (row ids, index digests) ---------------------------------------------
def hash_to_int_value(h):
    """mp2-common/src/poseidon.rs:120-133: the 128-bit integer e0 + e1 * 2^64 of the two low hash limbs"""
    return int(h[0]) | (int(h[1]) << 64)


def row_unique_data(ctx, columns_u32be, variant=POSEIDON2):
    """mp2-v1/src/values_extraction/mod.rs:499-510: H(left_pad32(column).pack(Big) for every column), 4 limbs.
    columns_u32be: uint32 [n_unique][8], most significant word first (the packing of `u256_to_limbs`)."""
    limbs = np.asarray(columns_u32be, dtype=np.uint32).reshape(1, -1).astype(np.uint64)
    return ctx.hash_no_pad_batch(limbs, 4, variant)[0]


def compute_row_id(ctx, unique_hash, num_actual_columns, variant=POSEIDON2):
    """values_extraction/mod.rs:512-523: H2int(row_unique_data || num_actual_columns) as a python int < 2^128"""
    inputs = np.concatenate([np.asarray(unique_hash, dtype=np.uint64), [np.uint64(num_actual_columns)]]).reshape(1, -1)
    return hash_to_int_value(ctx.hash_no_pad_batch(inputs, 4, variant)[0])


def compute_index_digest(ctx, inputs, digest_w, variant=POSEIDON2):
    """verifiable-db/src/block_tree/mod.rs:49-54: H2int(inputs) * digest (the same map as field_hashed_scalar_mul)"""
    return field_hashed_scalar_mul(ctx, np.asarray(inputs, dtype=np.uint64), digest_w, variant)[0]


def add_primary_index_to_digest(ctx, primary_index_id, index_value_u32be, digest_w, variant=POSEIDON2):
    """block_tree/mod.rs:36-46: inputs = id || index_value.to_fields() (8 big-endian u32 words)"""
    inputs = np.concatenate([[np.uint64(primary_index_id)], np.asarray(index_value_u32be, dtype=np.uint32).astype(np.uint64)])
    return compute_index_digest(ctx, inputs, digest_w, variant)


# ---- the provable commitment of an off-chain table (mp2-v1/src/api.rs:553-612) -----------------------------------
def flatten_poseidon_hash_value(h):
    """mp2-common/src/poseidon.rs:92-103: per limb [high 32 bits, low 32 bits]"""
    return [x for limb in h for x in (int(limb) >> 32, int(limb) & 0xFFFFFFFF)]


def update_off_chain_data_commitment(ctx, primary_index_id, primary_values, col_ids, values, row_unique_columns, old_commitment=None,
                                     variant=POSEIDON2):
    """update_off_chain_data_commitment (mp2-v1/src/api.rs:556-603): the new rows of an off-chain table folded into its commitment.
    A TableRow is its primary-index cell and its other columns: primary_values uint32 [rows][8], col_ids [n_cols] (the other
    columns' identifiers, the same for every row), values uint32 [rows][n_cols][8] -- every U256 as 8 big-endian u32 words
    (`u256_to_limbs`); row_unique_columns: identifiers among col_ids (the reference refuses others, values_extraction/mod.rs:535);
    old_commitment: 32 bytes (a HashOutput) or None. Rows are grouped by increasing primary value (the BTreeMap of :562-570); per
    group g: commitment <- flatten(H(commitment || add_primary_index_to_digest(id, g, compute_table_row_digest(rows of g)).to_fields())).
    Returns the new commitment, 32 bytes.

    The group work is batched on the device -- one mp2g_row_digests call for every row's term, one mp2g_curve_sum_ranges for the
    groups' sums, one hash batch + mp2g_scalar_mul_batch for the primary-index scalars -- and only the chain of G small hashes,
    which is sequential by definition, runs group by group."""
    ids = [int(x) for x in np.asarray(col_ids, dtype=np.uint64).ravel()]
    v = np.ascontiguousarray(values, dtype=np.uint32)
    pv = np.ascontiguousarray(primary_values, dtype=np.uint32).reshape(-1, 8)
    rows = pv.shape[0]
    assert v.shape == (rows, len(ids), 8)
    uniq = []
    for c in row_unique_columns:
        if int(c) not in ids:
            raise ValueError(f"row-unique column {int(c)} is not a column of the table")  # ensure!(...) of mod.rs:535
        uniq.append(ids.index(int(c)))
    com = [0] * 8
    if old_commitment is not None:
        b = bytes(old_commitment)
        assert len(b) == 32
        com = [int.from_bytes(b[4 * i:4 * i + 4], "little") for i in range(8)]  # HashOutput.pack(Endianness::Little), :572-580
    if rows:
        order = np.lexsort(tuple(pv[:, j] for j in range(7, -1, -1)))  # increasing U256: most significant word first
        pv, v = pv[order], v[order]
        new_group = np.concatenate([[True], np.any(pv[1:] != pv[:-1], axis=1)])
        starts = np.flatnonzero(new_group)
        ranges = np.stack([starts, np.concatenate([starts[1:], [rows]])], axis=1).astype(np.uint32)
        unique = np.ascontiguousarray(v[:, uniq, :]) if uniq else np.zeros((rows, 0, 8), dtype=np.uint32)
        row_w, _ = row_digests(ctx, np.asarray(ids, dtype=np.uint64), v, unique, variant)
        group_w, _ = curve_sum_ranges(ctx, row_w, ranges)
        # add_primary_index_to_digest (block_tree/mod.rs:37-53) for every group at once
        inputs = np.concatenate([np.full((len(starts), 1), int(primary_index_id), dtype=np.uint64), pv[starts].astype(np.uint64)], axis=1)
        scalars = [hash_to_int_value(h) for h in ctx.hash_no_pad_batch(inputs, 4, variant)]
        _, fields = scalar_mul_batch(ctx, group_w, scalars, weierstrass=True)
        for g in range(len(starts)):
            payload = np.concatenate([np.asarray(com, dtype=np.uint64), fields[g]])
            com = flatten_poseidon_hash_value(ctx.hash_no_pad(payload, variant))
    return b"".join(int(x).to_bytes(4, "little") for x in com)


def off_chain_data_commitment(ctx, primary_index_id, primary_values, col_ids, values, row_unique_columns, variant=POSEIDON2):
    """off_chain_data_commitment (api.rs:606-612): the commitment of a whole table = the update from no commitment"""
    return update_off_chain_data_commitment(ctx, primary_index_id, primary_values, col_ids, values, row_unique_columns, None, variant)
