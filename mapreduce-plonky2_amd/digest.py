"""Off-circuit digest bookkeeping of the reference over the C ABI: host mirror of mp2-common/src/digest.rs
(`SplitDigestPoint`, lines 19-55) and of the conditional hashed scalar multiplication of
mp2-common/src/group_hashing/mod.rs:220-234. Points travel as their 5-limb Ecgfp5 encodings
(`Point::encode`); NEUTRAL encodes as five zeros. Every group operation runs in libmp2gpu."""
import numpy as np

from . import POSEIDON2, curve_sum, field_hashed_scalar_mul, map_to_curve_batch

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
