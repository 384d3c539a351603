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
