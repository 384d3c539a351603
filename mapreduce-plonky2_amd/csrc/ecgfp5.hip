// Batched Ecgfp5 arithmetic for the off-circuit multiset digest, for gfx950.
//
// Replaces, on the value (non-circuit) side:
//   mp2-common/src/group_hashing/field_to_curve.rs:36-48   map_to_curve_point
//   mp2-common/src/group_hashing/sswu_value.rs:31-77       simple_swu (same operation order)
//   mp2-common/src/group_hashing/utils.rs:9-82             SWU constants
//   mp2-common/src/group_hashing/curve_add.rs:17-33        add_curve_point
//   mp2-common/src/group_hashing/mod.rs:163-174,220-225    Weierstrass limbs, field_hashed_scalar_mul
//   mp2-common/src/poseidon.rs:120-133                     hash_to_int_value
//   mp2-v1/src/values_extraction/mod.rs:499-571            row_unique_data / compute_row_id /
//                                                          compute_table_row_digest
//   verifiable-db/src/cells_tree/mod.rs:65-72              Cell::values_digest
// and [dep] plonky2_ecgfp5 curve/{base_field,curve}.rs (GF(p^5) sqrt / inverse / sgn0 / legendre,
// Point decode / encode / add / double / scalar mul).
//
// ALU-bound: one lane owns one point in fractional coordinates (X:Z:U:T), x = X/Z, u = U/T,
// using the complete 10M addition and 4M+5S doubling of the ecgfp5 paper (checked against the
// affine chord/tangent law of the oracle). GF(p^5) products accumulate the five partial products of an
// output limb in carry-free 64-bit columns (gl_cols) and reduce once. Encodings are canonical, so results are identical
// to the reference's regardless of the coordinate system.
#include "ecgfp5.h"
#include "poseidon.cuh"

namespace mp2g {

// large bodies are real functions: the SWU / scalar-mul kernels call them hundreds of times
// EC_WAVES_ATTR (tools/dbg/ec_variants.sh): the occupancy the register allocator is held to on the kernels (the attribute is for
// kernels only; the out-of-line bodies are compiled for any workgroup size, i.e. within 128 VGPRs); empty = the allocator's own
// choice (it fills the 256 VGPRs a 128-lane block can have)
#ifndef EC_WAVES_ATTR
#define EC_WAVES_ATTR
#endif
// EC_LB: the launch bound DECLARED on the three heavy kernels (they are always launched with 128 lanes). The bound reaches the
// out-of-line bodies too (the flat work-group size propagates to callees): 1024 would hold everything to 128 VGPRs, 768 to 168
#ifndef EC_LB
#define EC_LB 128
#endif
#define GLN __device__ __noinline__
struct gl5 { u64 c[5]; };

GLD gl5 gl5_zero() { gl5 r; for (int i = 0; i < 5; i++) r.c[i] = 0; return r; }
GLD gl5 gl5_from(u64 a) { gl5 r = gl5_zero(); r.c[0] = a; return r; }
GLD gl5 gl5_make(u64 a, u64 b, u64 c, u64 d, u64 e) { gl5 r; r.c[0] = a; r.c[1] = b; r.c[2] = c; r.c[3] = d; r.c[4] = e; return r; }
GLD bool gl5_is_zero(const gl5& a) { return (a.c[0] | a.c[1] | a.c[2] | a.c[3] | a.c[4]) == 0; }
GLD bool gl5_eq(const gl5& a, const gl5& b) {
  bool e = true;
#pragma unroll
  for (int i = 0; i < 5; i++) e = e && a.c[i] == b.c[i];
  return e;
}
GLD gl5 gl5_add(const gl5& a, const gl5& b) { gl5 r;
#pragma unroll
  for (int i = 0; i < 5; i++) r.c[i] = gl_add(a.c[i], b.c[i]);
  return r; }
GLD gl5 gl5_sub(const gl5& a, const gl5& b) { gl5 r;
#pragma unroll
  for (int i = 0; i < 5; i++) r.c[i] = gl_sub(a.c[i], b.c[i]);
  return r; }
GLD gl5 gl5_neg(const gl5& a) { gl5 r;
#pragma unroll
  for (int i = 0; i < 5; i++) r.c[i] = gl_neg(a.c[i]);
  return r; }
GLD gl5 gl5_dbl(const gl5& a) { return gl5_add(a, a); }
GLD gl5 gl5_scale(const gl5& a, u64 s) { gl5 r;
#pragma unroll
  for (int i = 0; i < 5; i++) r.c[i] = gl_mul(a.c[i], s);
  return r; }
GLD gl5 gl5_small(const gl5& a, u32 s) { gl5 r;
#pragma unroll
  for (int i = 0; i < 5; i++) r.c[i] = gl_mul_small(a.c[i], s);
  return r; }
// a * (k z): coefficients rotate, the wrapped one picks up z^5 = 3
GLD gl5 gl5_mul_kz(const gl5& a, u32 k) {
  gl5 r;
  r.c[0] = gl_mul_small(a.c[4], 3 * k);
#pragma unroll
  for (int i = 1; i < 5; i++) r.c[i] = gl_mul_small(a.c[i - 1], k);
  return r;
}
// The one out-of-line body of a GF(p^5) product takes its ten limbs as scalars: clang's AMDGPU ABI keeps at most 16 dwords of
// aggregate arguments in registers and sends the rest through the stack, scalars all travel in VGPRs. (With both operands by
// reference every 600-instruction multiplication began with six flat loads from the stack, and row_digest_kernel sat parked for
// 43 % of its cycles -- tools/dbg/step_pmc.sh.)
GLN gl5 gl5_mul_limbs(u64 x0, u64 x1, u64 x2, u64 x3, u64 x4, u64 y0, u64 y1, u64 y2, u64 y3, u64 y4) {
  const u64 a[5] = {x0, x1, x2, x3, x4}, b[5] = {y0, y1, y2, y3, y4};
  u64 a3[5];
#pragma unroll
  for (int j = 1; j < 5; j++) a3[j] = gl_mul_small_w(a[j], 3);  // only ever a multiplicand: a weak representative will do
  a3[0] = 0;
  gl5 r;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    gl_cols acc;  // five partial products per output limb in carry-free columns, one reduction
#pragma unroll
    for (int j = 0; j < 5; j++) {
      if (j <= i) acc.add(a[j], b[i - j]); else acc.add(a3[j], b[i + 5 - j]);
    }
    r.c[i] = acc.value();
  }
  return r;
}
GLD gl5 gl5_mul(const gl5& a, const gl5& b) {
  return gl5_mul_limbs(a.c[0], a.c[1], a.c[2], a.c[3], a.c[4], b.c[0], b.c[1], b.c[2], b.c[3], b.c[4]);
}
// a^2 with the symmetry used: 15 products a_j a_k (j <= k) instead of 25 -- each enters output limb (j + k) mod 5 with the factor
// (2 if j < k) * (3 if j + k >= 5, z^5 = 3) folded into the column accumulation. Squarings are over half of the multiset digest's
// GF(p^5) operations (63 per square root, 5 of the 9 products of a point doubling).
#ifndef EC_NO_SQR
GLN gl5 gl5_sqr_limbs(u64 x0, u64 x1, u64 x2, u64 x3, u64 x4) {
  const u64 a[5] = {x0, x1, x2, x3, x4};
  gl5 r;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    gl_cols acc;
#pragma unroll
    for (int j = 0; j < 5; j++) {
#pragma unroll
      for (int k = j; k < 5; k++) {
        if ((j + k) % 5 != i) continue;
        const u32 f = (j < k ? 2u : 1u) * (j + k >= 5 ? 3u : 1u);
        if (f == 1) acc.add(a[j], a[k]); else acc.add_scaled(a[j], a[k], f);
      }
    }
    r.c[i] = acc.value();
  }
  return r;
}
GLD gl5 gl5_sqr(const gl5& a) { return gl5_sqr_limbs(a.c[0], a.c[1], a.c[2], a.c[3], a.c[4]); }
#else
GLD gl5 gl5_sqr(const gl5& a) { return gl5_mul(a, a); }
#endif
// Frobenius powers: coefficient i times (3^((p-1)/5))^(i*e)
GLD gl5 gl5_frob1(const gl5& a) {
  return gl5_make(a.c[0], gl_mul(a.c[1], 1041288259238279555ULL), gl_mul(a.c[2], 15820824984080659046ULL),
                  gl_mul(a.c[3], 211587555138949697ULL), gl_mul(a.c[4], 1373043270956696022ULL));
}
GLD gl5 gl5_frob2(const gl5& a) {
  return gl5_make(a.c[0], gl_mul(a.c[1], 15820824984080659046ULL), gl_mul(a.c[2], 1373043270956696022ULL),
                  gl_mul(a.c[3], 1041288259238279555ULL), gl_mul(a.c[4], 211587555138949697ULL));
}
GLD u64 gl_sqn(u64 x, int k) {
#pragma unroll 1
  for (int i = 0; i < k; i++) x = gl_sqr(x);
  return x;
}
// o31 = a^(2^31-1), o32 = a^(2^32-1) by an addition chain on runs of ones
GLD void gl_ones(u64 a, u64& o31, u64& o32) {
  u64 x2 = gl_mul(gl_sqr(a), a), x4 = gl_mul(gl_sqn(x2, 2), x2), x8 = gl_mul(gl_sqn(x4, 4), x4);
  u64 x16 = gl_mul(gl_sqn(x8, 8), x8), x24 = gl_mul(gl_sqn(x16, 8), x8), x28 = gl_mul(gl_sqn(x24, 4), x4);
  u64 x30 = gl_mul(gl_sqn(x28, 2), x2);
  o31 = gl_mul(gl_sqr(x30), a);
  o32 = gl_mul(gl_sqr(o31), a);
}
GLD u64 gl_pow_2_32_m1(u64 a) { u64 o31, o32; gl_ones(a, o31, o32); return o32; }
// a^(p-2), p-2 = (2^32-2)*2^32 + (2^32-1); 0 -> 0
GLD u64 gl_inv_chain(u64 a) {
  u64 o31, o32;
  gl_ones(a, o31, o32);
  return gl_mul(gl_sqn(gl_sqr(o31), 32), o32);
}
GLN gl5 gl5_inv(gl5 a) {  // inverse_or_zero
  gl5 f1 = gl5_frob1(a), f2 = gl5_frob2(a);
  gl5 f12 = gl5_mul(f1, f2);           // a^(p+p^2)
  gl5 f34 = gl5_frob2(f12);            // a^(p^3+p^4)
  gl5 q = gl5_mul(f12, f34);           // a^(r-1)
  u64 n = 0;                           // norm = (a*q)[0]
  {
    gl_cols acc;
    acc.add(a.c[0], q.c[0]);
#pragma unroll
    for (int j = 1; j < 5; j++) acc.add(gl_mul_small_w(a.c[j], 3), q.c[5 - j]);
    n = acc.value();
  }
  return gl5_scale(q, gl_inv_chain(n));
}
GLD u64 gl5_norm(const gl5& a) {
  gl5 f12 = gl5_mul(gl5_frob1(a), gl5_frob2(a));
  gl5 q = gl5_mul(f12, gl5_frob2(f12));
  return gl5_mul(a, q).c[0];
}
// Legendre symbol of a base-field element as a bool "is a non-zero square or zero"
GLD bool gl_is_square(u64 a) {
  if (a == 0) return true;
  u64 t = gl_pow_2_32_m1(a);  // a^(2^32-1); a^((p-1)/2) = t^(2^31)
#pragma unroll 1
  for (int i = 0; i < 31; i++) t = gl_sqr(t);
  return t == 1;
}
// Tonelli-Shanks, p - 1 = 2^32 (2^32 - 1); c-table GL_TWO_GEN_POW2[k] = g2^(2^k)
GLN bool gl_sqrt(u64 a, u64& out) {
  if (a == 0) { out = 0; return true; }
  u64 t = gl_pow_2_32_m1(a);  // a^q
  u64 chk = t;
#pragma unroll 1
  for (int i = 0; i < 31; i++) chk = gl_sqr(chk);
  if (chk != 1) { out = 0; return false; }
  u64 R = a;  // a^((q+1)/2) = a^(2^31)
#pragma unroll 1
  for (int i = 0; i < 31; i++) R = gl_sqr(R);
#pragma unroll 1
  while (t != 1) {
    u32 i = 0;
    u64 t2 = t;
    while (t2 != 1) { t2 = gl_sqr(t2); i++; }
    // c has order 2^M; b = c^(2^(M-i-1)) = g2^(2^(31-i)); new c = b^2
    u64 b = GL_TWO_GEN_POW2[31 - i];
    t = gl_mul(t, GL_TWO_GEN_POW2[32 - i]);
    R = gl_mul(R, b);
  }
  out = R;
  return true;
}
GLN bool gl5_sqrt(gl5 x, gl5& out) {
  gl5 v = x;
#pragma unroll 1
  for (int i = 0; i < 31; i++) v = gl5_sqr(v);
  gl5 v32 = v;
#pragma unroll 1
  for (int i = 0; i < 32; i++) v32 = gl5_sqr(v32);
  gl5 d = gl5_mul(gl5_mul(x, v32), gl5_inv(v));       // x^((p+1)/2)
  gl5 e = gl5_frob1(gl5_mul(d, gl5_frob2(d)));        // x^((r-1)/2)
  gl5 f = gl5_sqr(e);
  u64 g = gl5_mul(x, f).c[0];                         // x^r
  u64 s;
  if (!gl_sqrt(g, s)) { out = gl5_zero(); return false; }
  out = gl5_scale(gl5_inv(e), s);
  return true;
}
GLN bool gl5_is_square(const gl5& x) { return gl_is_square(gl5_norm(x)); }
GLD bool gl5_sgn0(const gl5& x) {
  bool sign = false, zero = true;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    bool sign_i = (x.c[i] & 1) != 0, zero_i = x.c[i] == 0;
    sign = sign || (zero && sign_i);
    zero = zero && zero_i;
  }
  return sign;
}

// ---- group ------------------------------------------------------------------------------------
struct pt { gl5 X, Z, U, T; };
#define EC_B1 263u
GLD pt pt_neutral() { pt p; p.X = gl5_zero(); p.Z = gl5_from(1); p.U = gl5_zero(); p.T = gl5_from(1); return p; }
GLN pt pt_add(const pt& p, const pt& q) {
  gl5 t1 = gl5_mul(p.X, q.X), t2 = gl5_mul(p.Z, q.Z), t3 = gl5_mul(p.U, q.U), t4 = gl5_mul(p.T, q.T);
  gl5 t5 = gl5_sub(gl5_sub(gl5_mul(gl5_add(p.X, p.Z), gl5_add(q.X, q.Z)), t1), t2);
  gl5 t6 = gl5_sub(gl5_sub(gl5_mul(gl5_add(p.U, p.T), gl5_add(q.U, q.T)), t3), t4);
  gl5 t7 = gl5_add(t1, gl5_mul_kz(t2, EC_B1));
  gl5 t8 = gl5_mul(t4, t7);
  gl5 t9 = gl5_mul(t3, gl5_add(gl5_mul_kz(t5, 2 * EC_B1), gl5_dbl(t7)));
  gl5 t10 = gl5_mul(gl5_add(t4, gl5_dbl(t3)), gl5_add(t5, t7));
  pt r;
  r.X = gl5_mul_kz(gl5_sub(t10, t8), EC_B1);
  r.Z = gl5_sub(t8, t9);
  r.U = gl5_mul(t6, gl5_sub(gl5_mul_kz(t2, EC_B1), t1));
  r.T = gl5_add(t8, t9);
  return r;
}
GLN pt pt_dbl(const pt& p) {
  gl5 t1 = gl5_mul(p.Z, p.T), t2 = gl5_mul(t1, p.T);
  gl5 X1 = gl5_sqr(t2), Z1 = gl5_mul(t1, p.U), t3 = gl5_sqr(p.U);
  gl5 W1 = gl5_sub(t2, gl5_mul(gl5_dbl(gl5_add(p.X, p.Z)), t3));
  gl5 t4 = gl5_sqr(Z1);
  pt r;
  r.X = gl5_mul_kz(t4, 4 * EC_B1);
  r.Z = gl5_sqr(W1);
  r.U = gl5_sub(gl5_sub(gl5_sqr(gl5_add(W1, Z1)), t4), r.Z);
  r.T = gl5_sub(gl5_sub(gl5_dbl(X1), gl5_small(t4, 4)), r.Z);
  return r;
}
// Four successive doublings (one window of pt_mul128). The first leaves the fractional coordinates for Jacobian ones, x = X / Z^2 and
// w = 1 / u = W / Z, in which doubling is 1M + 7S:  D = W^2 - 2X - 2Z^2,  X' = 16 b (WZ)^4,  W' = 2 W^4 - 4 (WZ)^2 - D^2,  Z' = 2 D W Z
// (the same map as pt_dbl, x' = 4 b w^2 / D_a^2 and w' = (2 w^4 - 4 w^2 - D_a^2) / (2 w D_a) with D_a = w^2 - 2x - 2, on those
// coordinates); (X : Z^2 : Z : W) are fractional coordinates again. 4M + 6S, then 3 x (1M + 7S), then 1S: 595 base products against
// 4 x (4M + 5S) = 700. The neutral element has Z = 0 in the Jacobian form and is put back by hand. (-DEC_DBL_PLAIN: four pt_dbl.)
GLN pt pt_dbl4(const pt& p) {
#ifdef EC_DBL_PLAIN
  return pt_dbl(pt_dbl(pt_dbl(pt_dbl(p))));
#else
  gl5 X, W, Z;
  {
    gl5 t1 = gl5_mul(p.Z, p.T), t2 = gl5_mul(t1, p.T);
    gl5 X1 = gl5_sqr(t2), Z1 = gl5_mul(t1, p.U), t3 = gl5_sqr(p.U);
    gl5 W1 = gl5_sub(t2, gl5_mul(gl5_dbl(gl5_add(p.X, p.Z)), t3));
    gl5 z2 = gl5_sqr(Z1), w2 = gl5_sqr(W1);
    Z = gl5_sub(gl5_sub(gl5_sqr(gl5_add(W1, Z1)), z2), w2);    // 2 W1 Z1
    X = gl5_mul_kz(gl5_sqr(z2), 16 * EC_B1);                    // 16 b Z1^4
    W = gl5_sub(gl5_sub(gl5_dbl(X1), gl5_small(z2, 4)), w2);    // 2 X1 - 4 Z1^2 - W1^2
  }
#pragma unroll 1
  for (int i = 0; i < 3; i++) {
    gl5 w2 = gl5_sqr(W), z2 = gl5_sqr(Z);
    gl5 wz2 = gl5_sub(gl5_sub(gl5_sqr(gl5_add(W, Z)), w2), z2);  // 2 W Z
    gl5 D = gl5_sub(gl5_sub(w2, gl5_dbl(X)), gl5_dbl(z2));
    gl5 a = gl5_sqr(wz2);                                         // 4 (WZ)^2
    X = gl5_mul_kz(gl5_sqr(a), EC_B1);
    W = gl5_sub(gl5_sub(gl5_dbl(gl5_sqr(w2)), a), gl5_sqr(D));
    Z = gl5_mul(D, wz2);
  }
  if (gl5_is_zero(Z)) return pt_neutral();
  pt r;
  r.X = X; r.Z = gl5_sqr(Z); r.U = Z; r.T = W;
  return r;
#endif
}
GLD gl5 pt_encode(const pt& p) { return gl5_mul(p.T, gl5_inv(p.U)); }  // neutral -> 0
// decode(w): x^2 - (w^2 - a) x + b = 0, keep the non-square root; (x, 1, 1, w)
GLN bool pt_decode(gl5 w, pt& out) {
  gl5 e = gl5_sub(gl5_sqr(w), gl5_from(2));
  gl5 b4 = gl5_zero(); b4.c[1] = 4 * EC_B1;
  gl5 delta = gl5_sub(gl5_sqr(e), b4);
  gl5 r;
  if (!gl5_sqrt(delta, r)) { out = pt_neutral(); return gl5_is_zero(w); }
  const u64 half = 0x7FFFFFFF80000001ULL;  // (p+1)/2
  gl5 x1 = gl5_scale(gl5_add(e, r), half), x2 = gl5_scale(gl5_sub(e, r), half);
  gl5 x = gl5_is_square(x1) ? x2 : x1;
  out.X = x; out.Z = gl5_from(1); out.U = gl5_from(1); out.T = w;
  return true;
}
// [x0..x4, y0..y4, is_inf] of the short Weierstrass image (mod.rs:163-174): X = x + 2/3, Y = -w x
GLD void pt_to_weierstrass(const pt& p, u64 out[11]) {
  gl5 w = pt_encode(p);
  gl5 x = gl5_mul(p.X, gl5_inv(p.Z));
  if (gl5_is_zero(x)) {
#pragma unroll
    for (int i = 0; i < 10; i++) out[i] = 0;
    out[10] = 1;
    return;
  }
  gl5 y = gl5_neg(gl5_mul(w, x));
  x.c[0] = gl_add(x.c[0], 6148914689804861441ULL);
#pragma unroll
  for (int i = 0; i < 5; i++) { out[i] = x.c[i]; out[5 + i] = y.c[i]; }
  out[10] = 0;
}
// k * p, k = 128-bit little-endian (k[0] least significant).
// Signed 4-bit windows: the scalar recoded into 33 digits in [-8, 8], {0..8} * p in the lane's scratch, four doublings (pt_dbl4:
// a run in Jacobian coordinates) and one complete addition (of +-table[|digit|]; -P = (X : Z : -U : T)) per digit: 128 doublings +
// 32 additions + 7 for the table. The
// bit-serial double-and-add this replaces (EC_MUL_BITSERIAL) paid close to 128 additions: a wave takes the "bit set" branch
// whenever any of its 64 lanes has the bit. The projective representative differs from the bit-serial one; every consumer reads
// points through the canonical encodings (pt_emit / pt_to_weierstrass) or adds them.
GLD pt pt_mul128(const pt& p, const u32 k[4]) {
#ifdef EC_MUL_BITSERIAL
  pt acc = pt_neutral();
#pragma unroll 1
  for (int i = 127; i >= 0; i--) {
    acc = pt_dbl(acc);
    if ((k[i >> 5] >> (i & 31)) & 1) acc = pt_add(acc, p);
  }
  return acc;
#else
  pt tab[9];
  tab[0] = pt_neutral(); tab[1] = p; tab[2] = pt_dbl(p); tab[3] = pt_add(tab[2], p); tab[4] = pt_dbl(tab[2]);
  tab[5] = pt_add(tab[4], p); tab[6] = pt_dbl(tab[3]); tab[7] = pt_add(tab[6], p); tab[8] = pt_dbl(tab[4]);
  u32 mag[4] = {0, 0, 0, 0}, neg[4] = {0, 0, 0, 0}, carry = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const u32 d = ((k[i >> 3] >> ((i & 7) * 4)) & 15) + carry;  // 0..16
    carry = d > 8;
    mag[i >> 3] |= (carry ? 16 - d : d) << ((i & 7) * 4);
    neg[i >> 3] |= carry << (i & 7);
  }
  pt acc = tab[carry];  // the 33rd digit
#pragma unroll
  for (int w = 3; w >= 0; w--) {
#pragma unroll 1
    for (int i = 7; i >= 0; i--) {
      acc = pt_dbl4(acc);
      pt q = tab[(mag[w] >> (i * 4)) & 15];
      if ((neg[w] >> i) & 1) q.U = gl5_neg(q.U);
      acc = pt_add(acc, q);
    }
  }
  return acc;
#endif
}

// sswu_value.rs:31-77
GLN pt simple_swu(gl5 u) {
  const gl5 two_thirds = gl5_from(6148914689804861441ULL);
  const gl5 a_sw = gl5_make(6148914689804861439ULL, 263, 0, 0, 0);
  const gl5 b_sw = gl5_make(15713893096167979237ULL, 6148914689804861265ULL, 0, 0, 0);
  const gl5 z_sw = gl5_make(GL_P - 4, GL_P - 1, 0, 0, 0);
  const gl5 neg_z_inv = gl5_make(4795794222525505369ULL, 3412737461722269738ULL, 8370187669276724726ULL,
                                 7130825117388110979ULL, 12052351772713910496ULL);
  const gl5 neg_b_div_a = gl5_make(6585749426319121644ULL, 16990361517133133838ULL, 3264760655763595284ULL,
                                   16784740989273302855ULL, 13434657726302040770ULL);
  gl5 denom_part = gl5_mul(z_sw, gl5_sqr(u));
  gl5 denom = gl5_add(gl5_sqr(denom_part), denom_part);
  gl5 tv1 = gl5_inv(denom);
  gl5 x1 = gl5_mul(gl5_is_zero(tv1) ? neg_z_inv : gl5_add(tv1, gl5_from(1)), neg_b_div_a);
  gl5 x2 = gl5_mul(denom_part, x1);
  gl5 gx1 = gl5_add(gl5_add(gl5_mul(x1, gl5_sqr(x1)), gl5_mul(a_sw, x1)), b_sw);
  gl5 x_sw = x1, y_pos;
#ifdef EC_SWU_PLAIN
  if (!gl5_sqrt(gx1, y_pos)) {
    gl5 gx2 = gl5_add(gl5_add(gl5_mul(x2, gl5_sqr(x2)), gl5_mul(a_sw, x2)), b_sw);
    x_sw = x2;
    gl5_sqrt(gx2, y_pos);
  }
#else
  // which candidate has a square g(x) is a Legendre symbol (a norm to GF(p) and 63 base-field squarings), an eighth of the square
  // root whose failure would say the same. Every lane computes g(x2) (three products) and the wave takes ONE square root, of the
  // lane's own choice: branching on the symbol would send a wave through gl5_sqrt twice, its lanes split over the two candidates.
  {
    const bool first = gl5_is_square(gx1);
    const gl5 gx2 = gl5_add(gl5_add(gl5_mul(x2, gl5_sqr(x2)), gl5_mul(a_sw, x2)), b_sw);
    gl5 g;
#pragma unroll
    for (int i = 0; i < 5; i++) { g.c[i] = first ? gx1.c[i] : gx2.c[i]; x_sw.c[i] = first ? x1.c[i] : x2.c[i]; }
    gl5_sqrt(g, y_pos);
  }
#endif
  gl5 x_cand = gl5_sub(x_sw, two_thirds);
  gl5 y_cand = gl5_sgn0(u) == gl5_sgn0(y_pos) ? y_pos : gl5_neg(y_pos);
  pt p;
#ifdef EC_SWU_PLAIN
  pt_decode(gl5_mul(y_cand, gl5_inv(x_cand)), p);
#else
  // Point::decode(w), w = y / x, without its square root: (x_cand, y_cand) is on y^2 = x (x^2 + a x + b), so w^2 - a = x + b / x and
  // the two roots of decode's quadratic x^2 - (w^2 - a) x + b are x_cand and b / x_cand; decode keeps the non-square one (their
  // product b = 263 z is a non-square, so exactly one is). The general path stays for the degenerate encodings.
  const gl5 xi = gl5_inv(x_cand);
  const gl5 w = gl5_mul(y_cand, xi);
  if (gl5_is_zero(w) || gl5_is_zero(x_cand)) {
    pt_decode(w, p);
  } else {
    p.X = gl5_is_square(x_cand) ? gl5_mul_kz(xi, EC_B1) : x_cand;
    p.Z = gl5_from(1); p.U = gl5_from(1); p.T = w;
  }
#endif
  return p;
}
template <int V>
GLD pt map_to_curve(const u64* in, u32 n) {
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  for (u32 p = 0; p < n; p += 8) {
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (p + k < n) s[k] = in[p + k];
    perm<V>(s);
  }
  return simple_swu(gl5_make(s[0], s[1], s[2], s[3], s[4]));
}
GLD void pt_store(u64* d, const pt& p) {
#pragma unroll
  for (int i = 0; i < 5; i++) { d[i] = p.X.c[i]; d[5 + i] = p.Z.c[i]; d[10 + i] = p.U.c[i]; d[15 + i] = p.T.c[i]; }
}
GLD pt pt_load(const u64* d) {
  pt p;
#pragma unroll
  for (int i = 0; i < 5; i++) { p.X.c[i] = d[i]; p.Z.c[i] = d[5 + i]; p.U.c[i] = d[10 + i]; p.T.c[i] = d[15 + i]; }
  return p;
}
GLD void pt_emit(const pt& p, u64* w, u64* wei) {
  if (w) { gl5 e = pt_encode(p); for (int i = 0; i < 5; i++) w[i] = e.c[i]; }
  if (wei) pt_to_weierstrass(p, wei);
}

// ---- kernels ----------------------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(EC_LB) EC_WAVES_ATTR map_to_curve_kernel(const u64* in, u32 in_len, u32 count, u64* w_out, u64* wei_out, u64* frac_out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  pt p = map_to_curve<V>(in + (u64)i * in_len, in_len);
  pt_emit(p, w_out ? w_out + 5 * (u64)i : nullptr, wei_out ? wei_out + 11 * (u64)i : nullptr);
  if (frac_out) pt_store(frac_out + 20 * (u64)i, p);
}
// decode encodings into fractional points; bad[0] is set when an encoding is invalid
__global__ void __launch_bounds__(128) decode_kernel(const u64* w_in, u32 count, u64* frac_out, u32* bad) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  gl5 w;
  for (int k = 0; k < 5; k++) w.c[k] = w_in[5 * (u64)i + k];
  pt p;
  if (!pt_decode(w, p)) atomicOr(bad, 1u);
  pt_store(frac_out + 20 * (u64)i, p);
}
// out[blockIdx] = sum of a strided subset of pts[0..count): lanes accumulate serially, then a
// tree over the block through LDS
__global__ void __launch_bounds__(128) sum_kernel(const u64* pts, u32 count, u64* out) {
  __shared__ u64 sm[128 * 20];
  const u32 t = threadIdx.x, stride = gridDim.x * blockDim.x;
  pt acc = pt_neutral();
  for (u32 i = blockIdx.x * blockDim.x + t; i < count; i += stride) acc = pt_add(acc, pt_load(pts + 20 * (u64)i));
  pt_store(sm + 20 * t, acc);
  __syncthreads();
#pragma unroll 1
  for (u32 s = 64; s > 0; s >>= 1) {
    if (t < s) {
      pt a = pt_load(sm + 20 * t), b = pt_load(sm + 20 * (t + s));
      pt_store(sm + 20 * t, pt_add(a, b));
    }
    __syncthreads();
  }
  if (t < 20) out[20 * (u64)blockIdx.x + t] = sm[t];
}
// out[blockIdx] = sum of pts[ranges[blockIdx][0] .. ranges[blockIdx][1]): one 64-lane block per range (a subtree of a tree laid out in
// order is one contiguous range), lanes stride through the range, then a tree over the wave through LDS
__global__ void __launch_bounds__(64) sum_ranges_kernel(const u64* pts, const u32* ranges, u64* out) {
  __shared__ u64 sm[64 * 20];
  const u32 t = threadIdx.x, lo = ranges[2 * blockIdx.x], hi = ranges[2 * blockIdx.x + 1];
  pt acc = pt_neutral();
  for (u32 i = lo + t; i < hi; i += 64) acc = pt_add(acc, pt_load(pts + 20 * (u64)i));
  pt_store(sm + 20 * t, acc);
  __syncthreads();
#pragma unroll 1
  for (u32 s = 32; s > 0; s >>= 1) {
    if (t < s && lo + t + s < hi) {  // lanes past the range hold the neutral point
      pt a = pt_load(sm + 20 * t), b = pt_load(sm + 20 * (t + s));
      pt_store(sm + 20 * t, pt_add(a, b));
    }
    __syncthreads();
  }
  if (t < 20) out[20 * (u64)blockIdx.x + t] = sm[t];
}
__global__ void emit_kernel(const u64* frac, u32 count, u64* w_out, u64* wei_out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  pt p = pt_load(frac + 20 * (u64)i);
  pt_emit(p, w_out ? w_out + 5 * (u64)i : nullptr, wei_out ? wei_out + 11 * (u64)i : nullptr);
}
__global__ void __launch_bounds__(EC_LB) EC_WAVES_ATTR scalar_mul_kernel(const u64* frac_in, const u32* scalars, u32 count, u64* frac_out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  u32 k[4];
  for (int j = 0; j < 4; j++) k[j] = scalars[4 * (u64)i + j];
  pt_store(frac_out + 20 * (u64)i, pt_mul128(pt_load(frac_in + 20 * (u64)i), k));
}
// one lane per table row: sum_c D(id_c || value_c), row id, row_id * row digest
template <int V>
__global__ void __launch_bounds__(EC_LB) EC_WAVES_ATTR row_digest_kernel(const u64* col_ids, u32 n_cols, const u32* values, const u32* unique,
                                                          u32 n_unique, u32 rows, u64* frac_out) {
  u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  pt rd = pt_neutral();
  for (u32 c = 0; c < n_cols; c++) {
    u64 in[9];
    in[0] = col_ids[c];
    const u32* v = values + ((u64)r * n_cols + c) * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) in[1 + j] = v[j];
    rd = pt_add(rd, map_to_curve<V>(in, 9));
  }
  // row_unique_data = H(unique columns as 8 big-endian u32 limbs each)   (mod.rs:499-510)
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  const u32* uq = unique + (u64)r * n_unique * 8;
  for (u32 c = 0; c < n_unique; c++) {
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = uq[c * 8 + k];
    perm<V>(s);
  }
  // compute_row_id: H(row_unique_data(4) || num_actual_columns)[0..2] -> 128-bit scalar (mod.rs:512-523)
  u64 h[4] = {s[0], s[1], s[2], s[3]};
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  s[0] = h[0]; s[1] = h[1]; s[2] = h[2]; s[3] = h[3]; s[4] = n_cols;
  perm<V>(s);
  u32 k128[4] = {(u32)s[0], (u32)(s[0] >> 32), (u32)s[1], (u32)(s[1] >> 32)};
  pt_store(frac_out + 20 * (u64)r, pt_mul128(rd, k128));
}

// ---- launchers --------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)
static inline dim3 g128(u32 n) { return dim3((n + 127) / 128); }

hipError_t ec_map_to_curve(hipStream_t s, int variant, const u64* in, u32 in_len, u32 count, u64* w_out, u64* wei_out, u64* frac_out) {
  if (!count) return hipSuccess;
  if (variant == MP2G_POSEIDON2) hipLaunchKernelGGL((map_to_curve_kernel<MP2G_POSEIDON2>), g128(count), dim3(128), 0, s, in, in_len, count, w_out, wei_out, frac_out);
  else hipLaunchKernelGGL((map_to_curve_kernel<MP2G_POSEIDON>), g128(count), dim3(128), 0, s, in, in_len, count, w_out, wei_out, frac_out);
  return hipGetLastError();
}
hipError_t ec_decode(hipStream_t s, const u64* w_in, u32 count, u64* frac_out, u32* bad) {
  if (!count) return hipSuccess;
  hipLaunchKernelGGL(decode_kernel, g128(count), dim3(128), 0, s, w_in, count, frac_out, bad);
  return hipGetLastError();
}
// reduces frac[0..count) to one point in scratch[0..20); scratch needs 20*1024 words
hipError_t ec_sum(hipStream_t s, const u64* frac, u32 count, u64* scratch) {
  u32 blocks = (count + 127) / 128;
  if (blocks > 1024) blocks = 1024;
  if (blocks == 0) blocks = 1;
  if (blocks > 1) {
    hipLaunchKernelGGL(sum_kernel, dim3(blocks), dim3(128), 0, s, frac, count, scratch + 20);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(128), 0, s, scratch + 20, blocks, scratch);
  } else {
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(128), 0, s, frac, count, scratch);
  }
  return hipGetLastError();
}
hipError_t ec_sum_ranges(hipStream_t s, const u64* frac, const u32* ranges, u32 n_ranges, u64* frac_out) {
  if (!n_ranges) return hipSuccess;
  hipLaunchKernelGGL(sum_ranges_kernel, dim3(n_ranges), dim3(64), 0, s, frac, ranges, frac_out);
  return hipGetLastError();
}
hipError_t ec_emit(hipStream_t s, const u64* frac, u32 count, u64* w_out, u64* wei_out) {
  if (!count) return hipSuccess;
  hipLaunchKernelGGL(emit_kernel, g128(count), dim3(128), 0, s, frac, count, w_out, wei_out);
  return hipGetLastError();
}
hipError_t ec_scalar_mul(hipStream_t s, const u64* frac_in, const u32* scalars, u32 count, u64* frac_out) {
  if (!count) return hipSuccess;
  hipLaunchKernelGGL(scalar_mul_kernel, g128(count), dim3(128), 0, s, frac_in, scalars, count, frac_out);
  return hipGetLastError();
}
hipError_t ec_row_digest(hipStream_t s, int variant, const u64* col_ids, u32 n_cols, const u32* values, const u32* unique,
                         u32 n_unique, u32 rows, u64* frac_out) {
  if (!rows) return hipSuccess;
  if (variant == MP2G_POSEIDON2) hipLaunchKernelGGL((row_digest_kernel<MP2G_POSEIDON2>), g128(rows), dim3(128), 0, s, col_ids, n_cols, values, unique, n_unique, rows, frac_out);
  else hipLaunchKernelGGL((row_digest_kernel<MP2G_POSEIDON>), g128(rows), dim3(128), 0, s, col_ids, n_cols, values, unique, n_unique, rows, frac_out);
  return hipGetLastError();
}
}  // namespace mp2g
