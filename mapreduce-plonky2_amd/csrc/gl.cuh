// Goldilocks field arithmetic for gfx950 device code: p = 2^64 - 2^32 + 1, quadratic extension
// X^2 = 7 (FRI challenges / openings), quintic extension z^5 = 3 (Ecgfp5 base field).
// Replaces [dep] plonky2_field (goldilocks_field.rs, extension/{quadratic,quintic}.rs) on the path
// entered at recursion-framework/src/circuit_builder.rs:308.
// Every value is canonical (< p) on entry and exit, so results compare bit-for-bit with the
// CPU side of the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
// plonky2 goldilocks_field.rs MULTIPLICATIVE_GROUP_GENERATOR / POWER_OF_TWO_GENERATOR
#define GL_MULT_GEN 14293326489335486720ULL
#define GL_TWO_GEN 7277203076849721926ULL

#define GLD __device__ __forceinline__
#define GLHD __host__ __device__ __forceinline__

GLHD u64 gl_add(u64 a, u64 b) {
  u64 s = a + b;
  // a,b < p: a wrapped sum is < 2p - 2^64 < 2^32, so adding EPS (= 2^64 - p) cannot wrap again
  if (s < a) s += GL_EPS;
  if (s >= GL_P) s -= GL_P;
  return s;
}
GLHD u64 gl_sub(u64 a, u64 b) {
  u64 d = a - b;
  if (a < b) d += GL_P;
  return d;
}
GLHD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
GLHD u64 gl_dbl(u64 a) { return gl_add(a, a); }

// x = hi*2^64 + lo  ->  x mod p, using 2^64 = 2^32 - 1 and 2^96 = -1 (mod p)
GLHD u64 gl_reduce128(u64 lo, u64 hi) {
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GL_EPS;
  u64 t1 = (hi_lo << 32) - hi_lo;
  u64 r = t0 + t1;
  if (r < t1) r += GL_EPS;
  if (r >= GL_P) r -= GL_P;
  return r;
}
GLHD void gl_mul_wide(u64 a, u64 b, u64& lo, u64& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  // four 32x32->64 products; hipcc lowers the accumulations to v_mad_u64_u32
  u64 a0 = (u32)a, a1 = a >> 32, b0 = (u32)b, b1 = b >> 32;
  u64 p00 = a0 * b0;
  u64 m1 = a0 * b1 + (p00 >> 32);
  u64 m2 = a1 * b0 + (m1 & GL_EPS);
  lo = (m2 << 32) | (p00 & GL_EPS);
  hi = a1 * b1 + (m1 >> 32) + (m2 >> 32);
#else
  unsigned __int128 x = (unsigned __int128)a * b;
  lo = (u64)x;
  hi = (u64)(x >> 64);
#endif
}
GLHD u64 gl_mul(u64 a, u64 b) {
  u64 lo, hi;
  gl_mul_wide(a, b, lo, hi);
  return gl_reduce128(lo, hi);
}
GLHD u64 gl_sqr(u64 a) { return gl_mul(a, a); }
// a * c for a small constant c < 2^32 (MDS / M4 rows, W=7, W=3, 263)
GLHD u64 gl_mul_small(u64 a, u32 c) {
  u64 a0 = (u32)a, a1 = a >> 32;
  u64 p0 = a0 * c;
  u64 p1 = a1 * c + (p0 >> 32);
  u64 lo = (p1 << 32) | (p0 & GL_EPS);
  u64 hi = p1 >> 32;  // < 2^32
  u64 t1 = (hi << 32) - hi;
  u64 r = lo + t1;
  if (r < t1) r += GL_EPS;
  if (r >= GL_P) r -= GL_P;
  return r;
}
GLHD u64 gl_pow7(u64 x) {
  u64 x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
  return gl_mul(x3, x4);
}
GLHD u64 gl_pow(u64 b, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_sqr(b);
    e >>= 1;
  }
  return r;
}
GLHD u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }
GLHD u64 gl_root_of_unity(unsigned k) {
  u64 g = GL_TWO_GEN;
  for (unsigned i = k; i < 32; i++) g = gl_sqr(g);
  return g;
}
GLHD u32 bitrev32(u32 x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
  u32 r = 0;
  for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
#endif
}

// ---- quadratic extension ------------------------------------------------------------------
struct gl2 { u64 a, b; };
GLHD gl2 gl2_make(u64 a, u64 b) { gl2 r; r.a = a; r.b = b; return r; }
GLHD gl2 gl2_add(gl2 x, gl2 y) { return gl2_make(gl_add(x.a, y.a), gl_add(x.b, y.b)); }
GLHD gl2 gl2_sub(gl2 x, gl2 y) { return gl2_make(gl_sub(x.a, y.a), gl_sub(x.b, y.b)); }
GLHD gl2 gl2_mul(gl2 x, gl2 y) {
  u64 aa = gl_mul(x.a, y.a), bb = gl_mul(x.b, y.b);
  u64 cross = gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a));
  return gl2_make(gl_add(aa, gl_mul_small(bb, 7)), cross);
}
GLHD gl2 gl2_scale(gl2 x, u64 s) { return gl2_make(gl_mul(x.a, s), gl_mul(x.b, s)); }
GLHD gl2 gl2_inv(gl2 x) {
  u64 n = gl_sub(gl_sqr(x.a), gl_mul_small(gl_sqr(x.b), 7));
  u64 ni = gl_inv(n);
  return gl2_make(gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni));
}
GLHD gl2 gl2_pow(gl2 b, u64 e) {
  gl2 r = gl2_make(1, 0);
  while (e) {
    if (e & 1) r = gl2_mul(r, b);
    b = gl2_mul(b, b);
    e >>= 1;
  }
  return r;
}
