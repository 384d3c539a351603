// TEST INFRASTRUCTURE -- CPU oracle. Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything under oracle/. The product (libmp2gpu) never links this.
//
// Goldilocks field p = 2^64 - 2^32 + 1, its quadratic extension F[X]/(X^2-7) and quintic
// extension F[z]/(z^5-3).
// Restates [dep] plonky2_field 0.2.2 (Lagrange-Labs/plonky2 @22c42f6, absent from /root/reference):
//   field/src/goldilocks_field.rs (ORDER, MULTIPLICATIVE_GROUP_GENERATOR, POWER_OF_TWO_GENERATOR,
//   reduce128), field/src/extension/quadratic.rs (W = 7), field/src/extension/quintic.rs (W = 3).
// In-tree anchors: mp2-common/src/group_hashing/utils.rs:51 (ORDER = 0xFFFFFFFF00000001),
//   utils.rs:19-21 (B = [0,263,0,0,0] => z^5 = 3 is consistent with the published curve).
// All values are canonical (< p) on entry and exit of every function.
#ifndef MP2_ORACLE_GL_H
#define MP2_ORACLE_GL_H
#include <stdint.h>
#include <string.h>

typedef uint64_t gl_t;
#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
// [dep] goldilocks_field.rs: MULTIPLICATIVE_GROUP_GENERATOR = 14293326489335486720,
// POWER_OF_TWO_GENERATOR = 7277203076849721926 = MULT_GEN^((p-1)/2^32) (checked in
// tests/test_oracle_field.py; Plonky3 uses the other consistent pair 7 / 1753635133440165772 --
// listed in DESIGN.md "reference reconciliation").
#define GL_MULT_GEN 14293326489335486720ULL
#define GL_TWO_GEN 7277203076849721926ULL
#define GL_TWO_ADICITY 32

// branch-free throughout: the conditions are data dependent and unpredictable, a mispredicted branch costs more
// than the whole multiplication
static inline gl_t gl_add(gl_t a, gl_t b) {
  gl_t s;
  uint64_t c = __builtin_add_overflow(a, b, &s);
  return s - ((0 - (c | (uint64_t)(s >= GL_P))) & GL_P);
}
static inline gl_t gl_sub(gl_t a, gl_t b) {
  gl_t d;
  uint64_t br = __builtin_sub_overflow(a, b, &d);
  return d + ((0 - br) & GL_P);
}
static inline gl_t gl_neg(gl_t a) { return a ? GL_P - a : 0; }
static inline gl_t gl_reduce128(unsigned __int128 x) {
  uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  uint64_t t0, r;
  uint64_t br = __builtin_sub_overflow(lo, hi_hi, &t0);
  t0 -= (0 - br) & GL_EPS;
  uint64_t t1 = hi_lo * GL_EPS;
  uint64_t c = __builtin_add_overflow(t0, t1, &r);
  r += (0 - c) & GL_EPS;
  return r - ((0 - (uint64_t)(r >= GL_P)) & GL_P);
}
static inline gl_t gl_mul(gl_t a, gl_t b) { return gl_reduce128((unsigned __int128)a * b); }
static inline gl_t gl_sqr(gl_t a) { return gl_mul(a, a); }
static inline gl_t gl_from_u64(uint64_t x) { return x >= GL_P ? x - GL_P : x; }
static inline gl_t gl_pow(gl_t b, uint64_t e) {
  gl_t r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_sqr(b);
    e >>= 1;
  }
  return r;
}
static inline gl_t gl_inv(gl_t a) { return gl_pow(a, GL_P - 2); }  // 0 -> 0
static inline gl_t gl_pow7(gl_t x) {
  gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
  return gl_mul(x3, x4);
}
// primitive 2^k-th root of unity: POWER_OF_TWO_GENERATOR^(2^(32-k))
static inline gl_t gl_two_gen(void) { return GL_TWO_GEN; }
static inline gl_t gl_root_of_unity(unsigned k) {
  gl_t g = gl_two_gen();
  for (unsigned i = k; i < 32; i++) g = gl_sqr(g);
  return g;
}

// ---- quadratic extension, X^2 = 7 -------------------------------------------------------
typedef struct { gl_t c[2]; } gl2_t;
static inline gl2_t gl2_add(gl2_t a, gl2_t b) { return (gl2_t){{gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1])}}; }
static inline gl2_t gl2_sub(gl2_t a, gl2_t b) { return (gl2_t){{gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1])}}; }
static inline gl2_t gl2_mul(gl2_t a, gl2_t b) {
  gl_t a0b0 = gl_mul(a.c[0], b.c[0]), a1b1 = gl_mul(a.c[1], b.c[1]);
  gl2_t r;
  r.c[0] = gl_add(a0b0, gl_mul(7, a1b1));
  r.c[1] = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
  return r;
}
static inline gl2_t gl2_scale(gl2_t a, gl_t s) { return (gl2_t){{gl_mul(a.c[0], s), gl_mul(a.c[1], s)}}; }
static inline gl2_t gl2_from(gl_t a) { return (gl2_t){{a, 0}}; }
static inline gl2_t gl2_inv(gl2_t a) {
  // (a0 - a1 X) / (a0^2 - 7 a1^2)
  gl_t n = gl_sub(gl_sqr(a.c[0]), gl_mul(7, gl_sqr(a.c[1])));
  gl_t ni = gl_inv(n);
  return (gl2_t){{gl_mul(a.c[0], ni), gl_mul(gl_neg(a.c[1]), ni)}};
}
static inline gl2_t gl2_pow(gl2_t b, uint64_t e) {
  gl2_t r = gl2_from(1);
  while (e) {
    if (e & 1) r = gl2_mul(r, b);
    b = gl2_mul(b, b);
    e >>= 1;
  }
  return r;
}
static inline int gl2_eq(gl2_t a, gl2_t b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1]; }

// ---- quintic extension, z^5 = 3 ---------------------------------------------------------
typedef struct { gl_t c[5]; } gl5_t;
static inline gl5_t gl5_zero(void) { gl5_t r; memset(&r, 0, sizeof r); return r; }
static inline gl5_t gl5_from(gl_t a) { gl5_t r = gl5_zero(); r.c[0] = a; return r; }
static inline int gl5_is_zero(gl5_t a) { return !(a.c[0] | a.c[1] | a.c[2] | a.c[3] | a.c[4]); }
static inline int gl5_eq(gl5_t a, gl5_t b) { return memcmp(&a, &b, sizeof a) == 0; }
static inline gl5_t gl5_add(gl5_t a, gl5_t b) { gl5_t r; for (int i = 0; i < 5; i++) r.c[i] = gl_add(a.c[i], b.c[i]); return r; }
static inline gl5_t gl5_sub(gl5_t a, gl5_t b) { gl5_t r; for (int i = 0; i < 5; i++) r.c[i] = gl_sub(a.c[i], b.c[i]); return r; }
static inline gl5_t gl5_neg(gl5_t a) { gl5_t r; for (int i = 0; i < 5; i++) r.c[i] = gl_neg(a.c[i]); return r; }
static inline gl5_t gl5_scale(gl5_t a, gl_t s) { gl5_t r; for (int i = 0; i < 5; i++) r.c[i] = gl_mul(a.c[i], s); return r; }
static inline gl5_t gl5_mul(gl5_t a, gl5_t b) {
  gl_t t[9] = {0};
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) t[i + j] = gl_add(t[i + j], gl_mul(a.c[i], b.c[j]));
  gl5_t r;
  for (int i = 0; i < 4; i++) r.c[i] = gl_add(t[i], gl_mul(3, t[i + 5]));
  r.c[4] = t[4];
  return r;
}
static inline gl5_t gl5_sqr(gl5_t a) { return gl5_mul(a, a); }
// Frobenius x -> x^p: coefficient i is scaled by (3^((p-1)/5))^i
// ([dep] quintic DTH_ROOT = 1041288259238279555; the powers are recomputed in the field tests)
static const gl_t GL5_FROB[5] = {1ULL, 1041288259238279555ULL, 15820824984080659046ULL,
                                 211587555138949697ULL, 1373043270956696022ULL};
static inline gl5_t gl5_frob(gl5_t a) {
  gl5_t r;
  for (int i = 0; i < 5; i++) r.c[i] = gl_mul(a.c[i], GL5_FROB[i]);
  return r;
}
static inline gl5_t gl5_inv(gl5_t a) {  // inverse_or_zero
  gl5_t f1 = gl5_frob(a), f2 = gl5_frob(f1), f3 = gl5_frob(f2), f4 = gl5_frob(f3);
  gl5_t q = gl5_mul(gl5_mul(f1, f2), gl5_mul(f3, f4));  // a^(r-1)
  gl5_t n = gl5_mul(a, q);                              // norm, in GF(p)
  return gl5_scale(q, gl_inv(n.c[0]));
}
static inline gl_t gl5_norm(gl5_t a) {
  gl5_t f1 = gl5_frob(a), f2 = gl5_frob(f1), f3 = gl5_frob(f2), f4 = gl5_frob(f3);
  return gl5_mul(a, gl5_mul(gl5_mul(f1, f2), gl5_mul(f3, f4))).c[0];
}
#endif
