/* TEST INFRASTRUCTURE -- a second, separately written generator of the permutation constant tables.
 *
 * oracle/constants.h and mapreduce-plonky2_amd/csrc/perm_constants.h both come out of tools/gen_constants.py (Python): a slip in that
 * one generator would sit in the product and in its checker alike and no parity test would see it. This program derives the same
 * tables again from the published procedures, in C, sharing no code with the Python tools; tests/test_oracle_pins.py compiles it,
 * runs it and requires every table of both headers to equal its output.
 *
 *  - Poseidon2 (Goldilocks, t = 12, R_F = 8, R_P = 22): the Grain LFSR of the Poseidon / Poseidon2 reference scripts
 *    (poseidon2_rust_params.sage of the HorizenLabs instance [dep poseidon2_plonky2 takes its RC12 from]): 80-bit state seeded with
 *    the parameter encoding, taps 62 51 38 23 13 0, 160 discarded clocks, self-shrinking output, 64-bit big-endian draws, rejection
 *    of values >= p; 4 x 12 external, 22 internal (one per round), 4 x 12 external.
 *  - Poseidon (plonky2, t = 12, 8 + 22 rounds): plonky2/src/bin/generate_constants.rs: ChaCha8Rng::seed_from_u64(0), 360 draws of
 *    gen_range(0..ORDER) -- rand_core 0.6 seed expansion (PCG32), rand_chacha 0.3 (8 rounds, 64-bit counter), rand 0.8 uniform
 *    sampling by widening multiply with a rejection zone.
 *  - GL_TWO_GEN_POW2[k] = POWER_OF_TWO_GENERATOR^(2^k), k = 0..32.
 *
 * Output: one line per table, `NAME count v0 v1 ...` in hex. */
#include <stdint.h>
#include <stdio.h>

#define GL_P 0xFFFFFFFF00000001ULL
typedef unsigned __int128 u128;

/* ---- Grain LFSR: the 80-bit register as bit 0..79 of (lo, hi) with bit 0 the OLDEST (the one shifted out next) ---- */
typedef struct { uint64_t lo; uint16_t hi; } grain;
static unsigned g_bit(const grain* g, unsigned i) { return i < 64 ? (unsigned)((g->lo >> i) & 1) : (unsigned)((g->hi >> (i - 64)) & 1); }
static unsigned g_clock(grain* g) {
  unsigned nb = g_bit(g, 62) ^ g_bit(g, 51) ^ g_bit(g, 38) ^ g_bit(g, 23) ^ g_bit(g, 13) ^ g_bit(g, 0);
  g->lo = (g->lo >> 1) | ((uint64_t)(g->hi & 1) << 63);
  g->hi = (uint16_t)((g->hi >> 1) | (nb << 15));
  return nb;
}
static void g_init(grain* g, unsigned field, unsigned sbox, unsigned n, unsigned t, unsigned rf, unsigned rp) {
  /* the register is filled most significant parameter bit first; bit position 0 = first bit written */
  unsigned widths[6] = {2, 4, 12, 12, 10, 10}, vals[6] = {field, sbox, n, t, rf, rp}, pos = 0;
  g->lo = 0; g->hi = 0;
  for (int f = 0; f < 6; f++)
    for (int b = (int)widths[f] - 1; b >= 0; b--, pos++)
      if ((vals[f] >> b) & 1) { if (pos < 64) g->lo |= 1ULL << pos; else g->hi |= (uint16_t)(1u << (pos - 64)); }
  for (; pos < 80; pos++) { if (pos < 64) g->lo |= 1ULL << pos; else g->hi |= (uint16_t)(1u << (pos - 64)); }
  for (int i = 0; i < 160; i++) g_clock(g);
}
static unsigned g_out(grain* g) { /* self-shrinking: a pair (a, b) yields b when a = 1 */
  for (;;) {
    unsigned a = g_clock(g), b = g_clock(g);
    if (a) return b;
  }
}
static uint64_t g_field(grain* g) {
  for (;;) {
    uint64_t v = 0;
    for (int i = 0; i < 64; i++) v = (v << 1) | g_out(g);
    if (v < GL_P) return v;
  }
}

/* ---- ChaCha8Rng::seed_from_u64 + gen_range ---- */
#define ROTL(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
#define QR(a, b, c, d) a += b; d ^= a; d = ROTL(d, 16); c += d; b ^= c; b = ROTL(b, 12); a += b; d ^= a; d = ROTL(d, 8); c += d; b ^= c; b = ROTL(b, 7);
typedef struct { uint32_t key[8]; uint64_t counter; uint32_t buf[16]; int at; } chacha8;
static void cc_block(chacha8* r) {
  uint32_t s[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u}, x[16];
  for (int i = 0; i < 8; i++) s[4 + i] = r->key[i];
  s[12] = (uint32_t)r->counter; s[13] = (uint32_t)(r->counter >> 32); s[14] = 0; s[15] = 0;
  for (int i = 0; i < 16; i++) x[i] = s[i];
  for (int i = 0; i < 4; i++) {  /* 8 rounds = 4 double rounds */
    QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
    QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; i++) r->buf[i] = x[i] + s[i];
  r->counter++;
  r->at = 0;
}
static void cc_seed(chacha8* r, uint64_t state) {
  for (int i = 0; i < 8; i++) {  /* rand_core: PCG32 (XSH RR), the state advanced BEFORE each output */
    state = state * 6364136223846793005ULL + 11634580027462260723ULL;
    uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
    r->key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
  }
  r->counter = 0;
  r->at = 16;
}
static uint32_t cc_u32(chacha8* r) { if (r->at == 16) cc_block(r); return r->buf[r->at++]; }
static uint64_t cc_u64(chacha8* r) { uint64_t lo = cc_u32(r), hi = cc_u32(r); return lo | (hi << 32); }
static uint64_t cc_below(chacha8* r, uint64_t range) {
  int lz = __builtin_clzll(range);
  uint64_t zone = (range << lz) - 1;
  for (;;) {
    u128 m = (u128)cc_u64(r) * range;
    if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
  }
}

static uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % GL_P); }

static void line(const char* name, const uint64_t* v, int n) {
  printf("%s %d", name, n);
  for (int i = 0; i < n; i++) printf(" %016llx", (unsigned long long)v[i]);
  printf("\n");
}

int main(void) {
  uint64_t ext[96], in[22], rc[360], tw[33];
  grain g;
  g_init(&g, 1, 0, 64, 12, 8, 22);
  for (int i = 0; i < 48; i++) ext[i] = g_field(&g);
  for (int i = 0; i < 22; i++) in[i] = g_field(&g);
  for (int i = 48; i < 96; i++) ext[i] = g_field(&g);
  chacha8 r;
  cc_seed(&r, 0);
  for (int i = 0; i < 360; i++) rc[i] = cc_below(&r, GL_P);
  tw[0] = 7277203076849721926ULL;
  for (int k = 1; k < 33; k++) tw[k] = mulmod(tw[k - 1], tw[k - 1]);
  line("POSEIDON_RC", rc, 360);
  line("POSEIDON2_RC_EXT", ext, 96);
  line("POSEIDON2_RC_INT", in, 22);
  line("GL_TWO_GEN_POW2", tw, 33);
  return 0;
}
