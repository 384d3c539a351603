"""HIP path vs CPU oracle, bit for bit, through the C ABI: NTT/LDE, sponge hashing, Merkle trees,
PolynomialBatch commitments."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
P = O.P


@pytest.mark.parametrize("log_n", [1, 2, 3, 4, 5, 7, 9, 11, 12, 13, 14, 15, 16, 18])
@pytest.mark.parametrize("bitrev_out", [False, True])
def test_ntt_forward_matches_oracle(ctx, log_n, bitrev_out):
    n = 1 << log_n
    batch = 3 if log_n <= 14 else 1
    a = O.rand_field((batch, n), 0xC0FFEE02 + log_n)
    want = O.fft(a)
    if bitrev_out:
        want = want[:, O.bitrev_perm(n)]
    got = ctx.ntt(a, bitrev_out=bitrev_out)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("log_n", [1, 4, 8, 12, 13, 15, 17])
def test_ntt_inverse_and_coset(ctx, log_n):
    n = 1 << log_n
    a = O.rand_field((2, n), 77 + log_n)
    assert np.array_equal(ctx.ntt(a, inverse=True), O.fft(a, inverse=True))
    assert np.array_equal(ctx.ntt(a, coset_shift=O.MULT_GEN), O.fft(a, coset_shift=O.MULT_GEN))
    assert np.array_equal(ctx.ntt(a, inverse=True, coset_shift=O.MULT_GEN), O.fft(a, inverse=True, coset_shift=O.MULT_GEN))
    # round trip property
    assert np.array_equal(ctx.ntt(ctx.ntt(a), inverse=True), a)


@pytest.mark.parametrize("log_n", list(range(1, 25)))
def test_ntt_every_size_both_directions(ctx, log_n):
    """Every tile shape of the radix-8 kernels (single pass 2^1..2^12, two-pass 2^13..2^24), forward and inverse:
    size-independent properties -- inverse(forward(a)) = a, a delta transforms to all ones, linearity on a point
    sample, the coset transform of the inverse is the inverse of the coset transform -- plus the oracle where it
    is quick."""
    n = 1 << log_n
    a = O.rand_field((1, n), 900 + log_n)
    f = ctx.ntt(a)
    assert np.array_equal(ctx.ntt(f, inverse=True), a)
    assert np.array_equal(ctx.ntt(ctx.ntt(a, inverse=True, coset_shift=O.MULT_GEN), coset_shift=O.MULT_GEN), a)
    delta = np.zeros((1, n), dtype=np.uint64)
    delta[0, 0] = 1
    assert (ctx.ntt(delta) == 1).all()
    b = O.rand_field((1, n), 1900 + log_n)
    s = ((a.astype(object) + b.astype(object)) % P).astype(np.uint64)
    fs, fb = ctx.ntt(s), ctx.ntt(b)
    idx = np.unique(np.concatenate([[0, n - 1, n // 2], np.random.default_rng(log_n).integers(0, n, 64)]))
    assert all((int(f[0, i]) + int(fb[0, i])) % P == int(fs[0, i]) for i in idx)
    if log_n <= 17:
        assert np.array_equal(f, O.fft(a))
        assert np.array_equal(ctx.ntt(a, inverse=True), O.fft(a, inverse=True))


def test_ntt_edge_values(ctx):
    n = 1 << 10
    for fill in (0, 1, P - 1):
        a = np.full((1, n), fill, dtype=np.uint64)
        assert np.array_equal(ctx.ntt(a), O.fft(a))
    a = O.rand_field((1, n), 5)
    a[0, ::7] = np.uint64(P - 1)
    a[0, ::11] = 0
    assert np.array_equal(ctx.ntt(a), O.fft(a))


_V1_CHILD = """
import hashlib, importlib, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
import oracle as O
mp2 = importlib.import_module("mapreduce-plonky2_amd")
ctx = mp2.Context(0)
h = hashlib.sha256()
for log_n in (3, 7, 10, 12, 13, 16, 18, 22):
    a = O.rand_field((3 if log_n < 20 else 1, 1 << log_n), 4100 + log_n)
    for kw in ({}, {"bitrev_out": True}, {"inverse": True}, {"coset_shift": O.MULT_GEN, "bitrev_out": True}):
        h.update(ctx.ntt(a, **kw).tobytes())
for log_n, w in ((5, 7), (9, 5), (12, 3), (13, 5), (14, 2), (16, 1)):  # LDE: 8 cosets per polynomial, leaves in Merkle order
    h.update(ctx.lde_leaves(O.rand_field((w, 1 << log_n), 4200 + log_n), 3).tobytes())
print(h.hexdigest())
"""


def test_ntt_barrier_per_round_kernels_agree_with_the_one_barrier_ones():
    """the two kernel families of ntt.hip (default: one-barrier tiles; MP2G_NTT_V1=1: a barrier per round, also the
    fallback for the tile shapes the first does not cover) give the same transforms, size by size"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for v1 in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", _V1_CHILD, root], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MP2G_NTT_V1=v1))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and len(outs[0]) == 64


def test_ntt_2p22_config2(ctx):
    """BASELINE config 2(i): one 2^22-point polynomial, seed 0xC0FFEE02; fwd, inv, coset."""
    n = 1 << 22
    a = O.rand_field((1, n), 0xC0FFEE02)
    v = ctx.ntt(a)
    assert np.array_equal(v, O.fft(a))
    assert np.array_equal(ctx.ntt(v, inverse=True), a)
    assert np.array_equal(ctx.ntt(a, coset_shift=O.MULT_GEN, bitrev_out=True),
                          O.fft(a, coset_shift=O.MULT_GEN)[:, O.bitrev_perm(n)])


def test_commit_135_x_2p15_config2(ctx, mp2):
    """BASELINE config 2(ii) at full size: 135 polynomials of 2^15 values, seed 0xC0FFEE02 -> coefficients, LDE to 2^18 leaves of 135
    limbs, Poseidon2 Merkle cap of height 4: every coefficient, the cap and openings at the ends and inside against the oracle"""
    w, log_n = 135, 15
    vals = O.rand_field((w, 1 << log_n), 0xC0FFEE02)
    b = mp2.PolynomialBatch.from_values(ctx, vals, 3, 4, mp2.POSEIDON2)
    coeffs = O.fft(vals, inverse=True)
    assert np.array_equal(b.coeffs, coeffs)
    leaves = O.lde_leaves(coeffs, 3)
    cap = O.merkle_cap(O.merkle_build(leaves, 4, 0), 4)
    assert np.array_equal(b.cap, cap)
    N = 1 << (log_n + 3)
    idx = [0, 1, N - 1, N // 3, N // 2, 12345, N - 2]
    got_leaves, sib = b.open(idx)
    for k, i in enumerate(idx):
        assert np.array_equal(got_leaves[k], leaves[i])
        assert O.merkle_verify(got_leaves[k], i, sib[k], cap, 0)
    b.free()


@pytest.mark.parametrize("log_n,w", [(3, 2), (6, 5), (10, 7), (12, 9), (13, 3), (15, 2)])
def test_lde_leaves_match_oracle(ctx, log_n, w):
    c = O.rand_field((w, 1 << log_n), 31 + log_n)
    assert np.array_equal(ctx.lde_leaves(c, 3), O.lde_leaves(c, 3))


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("in_len,out_len", [(0, 4), (1, 4), (7, 4), (8, 4), (9, 5), (12, 4), (17, 4), (37, 4), (135, 4), (3, 11)])
def test_hash_no_pad_batch(ctx, variant, in_len, out_len):
    count = 300
    x = O.rand_field((count, in_len), 1000 + in_len) if in_len else np.zeros((count, 0), dtype=np.uint64)
    got = ctx.hash_no_pad_batch(x, out_len, variant)
    want = O.hash_no_pad_batch(x, out_len, variant)
    assert np.array_equal(got, want)


def test_hash_golden_column_id(ctx, mp2):
    """parsil/tests/context.json:88 through the HIP path."""
    limbs = [int.from_bytes(b"BLOCK_NUMBER"[i:i + 4], "big") for i in (0, 4, 8)]
    assert int(ctx.hash_no_pad(limbs, mp2.POSEIDON)[0]) == 17422912802427138938


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("log_leaves,leaf_len,cap_h", [(0, 5, 0), (1, 1, 0), (3, 4, 0), (3, 4, 3), (6, 7, 2), (10, 135, 4), (8, 32, 4), (2, 1, 0)])
def test_merkle_tree(ctx, mp2, variant, log_leaves, leaf_len, cap_h):
    L = 1 << log_leaves
    leaves = O.rand_field((L, leaf_len), 500 + log_leaves)
    t = mp2.MerkleTree(ctx, leaves, cap_h, variant)
    levels = O.merkle_build(leaves, cap_h, variant)
    cap = O.merkle_cap(levels, cap_h)
    assert np.array_equal(t.cap, cap)
    idx = sorted({0, L - 1, L // 2, min(3, L - 1)})
    got_leaves, sib = t.prove(idx)
    for k, i in enumerate(idx):
        assert np.array_equal(got_leaves[k], leaves[i])
        assert np.array_equal(sib[k], O.merkle_prove(levels, log_leaves, cap_h, i))
        assert O.merkle_verify(leaves[i], i, sib[k], cap, variant)
    t.free()


def test_circuit_set_tree_shape(ctx, mp2):
    """recursion-framework circuit_set.rs:173-191: 4-limb vk digests padded with [0] leaves,
    cap height 0; hash_or_noop keeps <=4-limb leaves verbatim."""
    digests = O.rand_field((3, 4), 42)
    leaves = np.zeros((4, 4), dtype=np.uint64)
    leaves[:3] = digests
    t = mp2.MerkleTree(ctx, leaves, 0)
    lv = O.merkle_build(leaves, 0)
    assert np.array_equal(t.cap, O.merkle_cap(lv, 0))
    assert np.array_equal(lv[:16].reshape(4, 4), leaves)  # leaf digests are the leaves themselves


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("log_n,w,cap_h", [(4, 3, 4), (7, 5, 4), (10, 20, 4), (12, 16, 4), (13, 9, 4), (12, 135, 4)])
def test_commit_from_values(ctx, mp2, variant, log_n, w, cap_h):
    n = 1 << log_n
    vals = O.rand_field((w, n), 0xC0FFEE01 + log_n * 100 + w)
    b = mp2.PolynomialBatch.from_values(ctx, vals, 3, cap_h, variant)
    coeffs = O.fft(vals, inverse=True)
    assert np.array_equal(b.coeffs, coeffs)
    leaves = O.lde_leaves(coeffs, 3)
    levels = O.merkle_build(leaves, cap_h, variant)
    cap = O.merkle_cap(levels, cap_h)
    assert np.array_equal(b.cap, cap)
    N = n << 3
    idx = [0, 1, N - 1, N // 3]
    got_leaves, sib = b.open(idx)
    for k, i in enumerate(idx):
        assert np.array_equal(got_leaves[k], leaves[i])
        assert O.merkle_verify(got_leaves[k], i, sib[k], cap, variant)
    b.free()


def test_api_errors(ctx, mp2):
    with pytest.raises(mp2.Mp2gError):
        mp2.MerkleTree(ctx, np.zeros((4, 3), dtype=np.uint64), 3)  # cap_height > log2(leaves)
    t = mp2.MerkleTree(ctx, np.zeros((4, 3), dtype=np.uint64), 0)
    with pytest.raises(mp2.Mp2gError):
        t.prove([4])
    with pytest.raises(mp2.Mp2gError):
        ctx.hash_no_pad_batch(np.zeros((1, 3), dtype=np.uint64), 4, variant=2)


class _At:
    """a view into a DeviceBuffer at a byte offset, for the *_dev entry points (they take .ptr)"""

    def __init__(self, buf, offset):
        import ctypes
        self.ptr = ctypes.c_void_p(buf.ptr.value + offset)


@pytest.mark.parametrize("log_n,batch", [(3, 5), (6, 7), (9, 3), (10, 5), (11, 3), (12, 3), (13, 3), (14, 2), (16, 1)])
def test_ntt_and_lde_stay_inside_their_buffers(ctx, log_n, batch):
    """no GPU AddressSanitizer on this pool: the transforms run between guard bands of a sentinel pattern -- ragged batches, so
    that the last tile of every pass is partly empty -- and the bands must come back untouched (in place and out of place,
    both output orders, inverse with the coset scaling, and the LDE with its 8 cosets per polynomial)"""
    n, guard = 1 << log_n, 1 << 12  # words
    sentinel = np.uint64(0xDEADBEEFCAFEF00D)
    a = O.rand_field((batch, n), 5200 + log_n)

    def banded(payload_words):
        host = np.full(payload_words + 2 * guard, sentinel, dtype=np.uint64)
        return ctx.to_device(host), host

    for kw in ({"bitrev_out": True}, {}, {"inverse": True, "coset_shift": O.MULT_GEN}):
        d_in, h_in = banded(batch * n)
        d_out, h_out = banded(batch * n)
        d_in.upload_at(a.ravel(), guard * 8)
        ctx.ntt_dev(_At(d_in, guard * 8), _At(d_out, guard * 8), log_n, batch, **kw)
        got = d_out.download((batch * n + 2 * guard,))
        assert (got[:guard] == sentinel).all() and (got[-guard:] == sentinel).all()
        assert np.array_equal(got[guard:-guard].reshape(batch, n), ctx.ntt(a, **kw))
        back = d_in.download((batch * n + 2 * guard,))
        assert (back[:guard] == sentinel).all() and (back[-guard:] == sentinel).all() and np.array_equal(back[guard:-guard], a.ravel())
        ctx.ntt_dev(_At(d_in, guard * 8), _At(d_in, guard * 8), log_n, batch, **kw)  # in place
        got = d_in.download((batch * n + 2 * guard,))
        assert (got[:guard] == sentinel).all() and (got[-guard:] == sentinel).all()
        assert np.array_equal(got[guard:-guard].reshape(batch, n), ctx.ntt(a, **kw))
        d_in.free(); d_out.free()
    if log_n <= 14:
        d_c, _ = banded(batch * n)
        d_v, _ = banded(batch * n * 8)
        d_c.upload_at(a.ravel(), guard * 8)
        ctx.lde_dev(_At(d_c, guard * 8), log_n, batch, 3, _At(d_v, guard * 8))
        got = d_v.download((batch * n * 8 + 2 * guard,))
        assert (got[:guard] == sentinel).all() and (got[-guard:] == sentinel).all()
        assert np.array_equal(got[guard:-guard].reshape(batch, n * 8).T, ctx.lde_leaves(a, 3))
        d_c.free(); d_v.free()


def test_degenerate_shapes(ctx, mp2):
    """the empty and one-element cases of every batched entry point: defined results (the identity transform, no output rows,
    the hash of the empty input, the neutral point), or the documented error -- never a crash or a launch with an empty grid"""
    a1 = O.rand_field((3, 1), 1)
    assert np.array_equal(ctx.ntt(a1), a1) and np.array_equal(ctx.ntt(a1, inverse=True), a1)  # n = 1: the identity
    assert ctx.ntt(np.zeros((0, 8), dtype=np.uint64)).shape == (0, 8)
    assert ctx.hash_no_pad_batch(np.zeros((0, 5), dtype=np.uint64)).shape == (0, 4)
    assert np.array_equal(ctx.hash_no_pad_batch(np.zeros((2, 0), dtype=np.uint64)), np.stack([O.hash_n_to_m_no_pad([], 4)] * 2))
    c = O.rand_field((2, 8), 3)
    assert np.array_equal(ctx.lde_leaves(c, 0), O.lde_leaves(c, 0))  # rate_bits 0: the coset transform alone
    for bad in (lambda: ctx.lde_leaves(O.rand_field((2, 1), 3), 3), lambda: mp2.PolynomialBatch.from_values(ctx, a1, 3, 0)):
        with pytest.raises(mp2.Mp2gError, match="log_n"):
            bad()
    one = O.rand_field((1, 7), 4)
    assert np.array_equal(mp2.MerkleTree(ctx, one, 0).cap, O.merkle_cap(O.merkle_build(one, 0), 0))  # a tree of one leaf
    four = O.rand_field((4, 7), 4)
    assert np.array_equal(mp2.MerkleTree(ctx, four, 2).cap, O.merkle_cap(O.merkle_build(four, 2), 2))  # cap = the leaf hashes
    leaves, siblings = mp2.MerkleTree(ctx, four, 1).prove([])
    assert leaves.shape == (0, 7) and siblings.shape[0] == 0
    # Ecgfp5: empty sums are the neutral point (encoding 0, Weierstrass form with is_inf = 1), empty batches give no rows
    neutral_w = np.zeros(5, dtype=np.uint64)
    assert np.array_equal(mp2.curve_sum(ctx, np.zeros((0, 5), dtype=np.uint64)), neutral_w)
    assert mp2.map_to_curve_batch(ctx, np.zeros((0, 9), dtype=np.uint64)).shape == (0, 5)
    assert mp2.scalar_mul_batch(ctx, np.zeros((0, 5), dtype=np.uint64), []).shape == (0, 5)
    w, wei = mp2.compute_table_row_digest(ctx, np.array([1, 2], dtype=np.uint64), np.zeros((0, 2, 8), dtype=np.uint32), np.zeros((0, 1, 8), dtype=np.uint32))
    assert np.array_equal(w, neutral_w) and wei.tolist() == [0] * 10 + [1]
