"""Regenerate plonky2's Poseidon-12 round constants.

[dep] plonky2 0.2.x `plonky2/src/bin/generate_constants.rs`: ChaCha8Rng::seed_from_u64(0),
360 x `rng.gen_range(0..GoldilocksField::ORDER)`.  Restated from the published algorithms of
rand_core 0.6 (`SeedableRng::seed_from_u64`: PCG32 expansion), rand_chacha 0.3 (ChaCha, 8 rounds,
64-bit block counter, 4-block buffer consumed in order) and rand 0.8 (`UniformInt::sample_single`:
widening multiply with the `(range << lz) - 1` rejection zone).
Pinned by the four leading constants recalled in SURVEY.md Appendix B, by the published
plonky2 all-zero / 0..11 permutation vectors, and by the in-tree column id
`parsil/tests/context.json:88` (see tests/test_oracle_pins.py).
"""
M32 = 0xFFFFFFFF
M64 = (1 << 64) - 1

def seed_from_u64(state):
    MUL, INC = 6364136223846793005, 11634580027462260723
    seed = b""
    for _ in range(8):
        state = (state * MUL + INC) & M64
        xorshifted = (((state >> 18) ^ state) >> 27) & M32
        rot = state >> 59
        x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & M32
        seed += x.to_bytes(4, "little")
    return seed

def rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & M32

def chacha_block(key_words, counter, rounds=8):
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [
        counter & M32, (counter >> 32) & M32, 0, 0]
    x = st[:]
    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = rotl(x[b] ^ x[c], 7)
    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[i] + st[i]) & M32 for i in range(16)]

class ChaCha8Rng:
    def __init__(self, seed_u64):
        seed = seed_from_u64(seed_u64)
        self.key = [int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(8)]
        self.counter = 0
        self.buf = []
    def next_u32(self):
        if not self.buf:
            self.buf = chacha_block(self.key, self.counter)
            self.counter += 1
        return self.buf.pop(0)
    def next_u64(self):
        lo = self.next_u32(); hi = self.next_u32()
        return lo | (hi << 32)

def gen_range_u64(rng, high):
    rng_range = high  # low = 0
    lz = 64 - rng_range.bit_length()
    zone = ((rng_range << lz) & M64) - 1
    while True:
        v = rng.next_u64()
        m = v * rng_range
        hi, lo = m >> 64, m & M64
        if lo <= zone:
            return hi

def poseidon12_round_constants():
    rng = ChaCha8Rng(0)
    out = [gen_range_u64(rng, 0xFFFFFFFF00000001) for _ in range(360)]
    return out

if __name__ == "__main__":
    c = poseidon12_round_constants()
    print([hex(v) for v in c[:6]])
    expect = [0xb585f766f2144405, 0x7746a55f43921ad7, 0xb2fb0d31cee799b4, 0x0f6760a4803427d7]
    print("MATCH" if c[:4] == expect else "MISMATCH")
