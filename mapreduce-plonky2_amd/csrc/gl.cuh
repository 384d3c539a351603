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

// What the operations cost on gfx950, in issue slots of a plain 32-bit add (tools/ubench; single-instruction streams measured in
// round 6, profiles/r06/ubench.txt): register move 0.79 (rounds 2-5 had inferred ~0.4 from A/B residuals), any 32-bit VALU operation
// 0.85, v_cndmask with an SGPR condition 1.5, 64-bit add WITHOUT carry-out (v_lshl_add_u64) 1.5-1.6, v_mad_u64_u32 1.7, a carry-chain
// instruction inside an asm chain 1.2, an add_co / addc pair with the hazard nop hipcc puts between them 4.4, a compare ~2.5, a 64-bit
// compare + select + add 6.2. Hence the rules below: a carry (or borrow)
// is taken from an add_co / addc pair or from the multiply-add's own carry-out only where the value really can wrap; every
// correction that provably cannot wrap is a carry-less 64-bit add of a mask; sums of products go through carry-free columns
// (gl_cols) or ride in the addend slots of the multiply-adds (gl_mul_add_wide); compares are avoided altogether.
GLHD u64 gl_mk(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
// a in [0, 2^64) -> canonical: a >= p  <=>  a + EPS carries out of 64 bits
GLHD u64 gl_canon(u64 a) {
  u32 c0, c1;
  u32 t0 = __builtin_addc((u32)a, 0xFFFFFFFFu, 0u, &c0);
  u32 t1 = __builtin_addc((u32)(a >> 32), 0u, c0, &c1);
  return c1 ? gl_mk(t0, t1) : a;
}
GLHD u64 gl_add(u64 a, u64 b) {  // canonical in, canonical out
  u32 c0, c1, d0, d1;
  u32 s0 = __builtin_addc((u32)a, (u32)b, 0u, &c0);
  u32 s1 = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c0, &c1);
  u32 t0 = __builtin_addc(s0, 0xFFFFFFFFu, 0u, &d0);  // s + EPS = s - p (mod 2^64)
  u32 t1 = __builtin_addc(s1, 0u, d0, &d1);
  return (c1 | d1) ? gl_mk(t0, t1) : gl_mk(s0, s1);
}
GLHD u64 gl_sub(u64 a, u64 b) {  // canonical in, canonical out
  u32 c0, c1, d0, d1;
  u32 s0 = __builtin_subc((u32)a, (u32)b, 0u, &c0);
  u32 s1 = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), c0, &c1);
  u32 m = c1 ? 0xFFFFFFFFu : 0u;  // borrow: + p = - EPS (mod 2^64)
  u32 t0 = __builtin_subc(s0, m, 0u, &d0);
  u32 t1 = __builtin_subc(s1, 0u, d0, &d1);
  return gl_mk(t0, t1);
}
GLHD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
GLHD u64 gl_dbl(u64 a) { return gl_add(a, a); }

// ---- weak representatives ---------------------------------------------------------------------
// Inside the hot loops a field element may be ANY u64 congruent to it (not necessarily < p);
// canonicalisation is paid once where a value leaves the kernel. Each routine states its needs.
// a: any u64, b: canonical. (a + b wraps to < p - 1, so the +EPS fix-up cannot wrap again.)
GLHD u64 gl_addw(u64 a, u64 b) {
  u32 c0, c1, d0, d1;
  u32 s0 = __builtin_addc((u32)a, (u32)b, 0u, &c0);
  u32 s1 = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c0, &c1);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GL_REDUCE_CARRYCHAIN)
  return gl_mk(s0, s1) + (u64)(c1 ? 0xFFFFFFFFu : 0u);
#endif
  u32 m = c1 ? 0xFFFFFFFFu : 0u;
  u32 t0 = __builtin_addc(s0, m, 0u, &d0);
  u32 t1 = __builtin_addc(s1, 0u, d0, &d1);
  return gl_mk(t0, t1);
}
// hi*2^64 + lo -> some u64 congruent to it (2^64 = EPS, 2^96 = -1 mod p); any lo, hi.
// With hi = hh 2^32 + hl: x = lo + hl EPS - hh. Device code spells the sequence out: one
// v_mad_u64_u32 forms t = hl EPS + lo and hands its carry c over in an SGPR pair, the subtract chain
// leaves the borrow b of t - hh in vcc, and r = t - hh + (c - b) EPS (mod 2^64) is exact:
//   c = 1, b = 0: t <= 2^64 - 2^33, so + EPS cannot wrap;  c = 0, b = 1: t - hh + 2^64 >= 2^64 - 2^32 + 1, so
//   - EPS cannot wrap;  c = b: the two corrections cancel mod 2^64.
// What the compiler makes of the portable form below re-derives carry and borrow with two v_cmp_*_u64
// (~4 issue slots each): 23 % slower per reduction, 12 % per multiply (tools/ubench, profiles/r01/ubench_alu.txt).
GLHD u64 gl_reduce128w(u64 lo, u64 hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 hl = (u32)hi, hh = (u32)(hi >> 32);
#ifdef GL_REDUCE_VCC
  // A/B of round 6 (variant builds only): the multiply-add leaves its carry in VCC and the mask is selected by VCC in the VOP2
  // encoding (a v_cndmask_b32 that selects by an SGPR pair costs 4.2 cycles as a stream, a VOP2 one 2.3: profiles/r06)
  {
    u64 tv;
    u32 mcv, q0, q1, mbv;
    const u32 ones = 0xFFFFFFFFu;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\t"
        "v_cndmask_b32_e32 %1, 0, %4, vcc"
        : "=&v"(tv), "=&v"(mcv) : "v"(hl), "v"(lo), "v"(ones) : "vcc");
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %4, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %0, %0, vcc"
        : "=&v"(q0), "=&v"(q1), "=&v"(mbv) : "v"((u32)tv), "v"((u32)(tv >> 32)), "v"(hh) : "vcc");
    return gl_mk(q0, q1) + gl_mk(mcv - mbv, mbv & ~mcv);
  }
#endif
  u64 t, c;
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=s"(c) : "v"(hl), "v"(lo));
  u32 t0 = (u32)t, t1 = (u32)(t >> 32), r0, r1, mb, mc;
#ifndef GL_REDUCE_CARRYCHAIN
  // both corrections as ONE signed 64-bit addend K = (c - b) EPS, added with a carry-less 64-bit add (v_lshl_add_u64): three
  // carry-chain instructions instead of seven. K = [mc - mb, mb & ~mc] for the masks mc = -c, mb = -b.
  asm("v_sub_co_u32 %0, vcc, %4, %6\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %0, %0, vcc\n\t"
      "v_cndmask_b32 %3, 0, -1, %7"
      : "=&v"(r0), "=&v"(r1), "=&v"(mb), "=&v"(mc)
      : "v"(t0), "v"(t1), "v"(hh), "s"(c)
      : "vcc");
  return gl_mk(r0, r1) + gl_mk(mc - mb, mb & ~mc);
#endif
  // Hazard discipline (gfx90a+: a VALU-written SGPR / VCC needs 2 wait states before another VALU reads it
  // as an explicit operand; the compiler cannot see inside asm): VCC is only consumed through the implicit
  // carry-in of VOP2 forms, the borrow mask is made by u0 - u0 - borrow, and the mad's carry pair %7 is
  // first read three VALU instructions into this block.
  asm("v_sub_co_u32 %0, vcc, %4, %6\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %0, %0, vcc\n\t"
      "v_cndmask_b32 %3, 0, -1, %7\n\t"
      "v_add_co_u32 %0, vcc, %0, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_sub_co_u32 %0, vcc, %0, %2\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(mb), "=&v"(mc)
      : "v"(t0), "v"(t1), "v"(hh), "s"(c)
      : "vcc");
  return gl_mk(r0, r1);
#else
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0;
  bool b = __builtin_sub_overflow(lo, hi_hi, &t0);  // borrow => t0 >= 2^64 - 2^32 + 1, so -EPS cannot borrow again
  t0 -= b ? GL_EPS : 0;
  u64 t1 = (hi_lo << 32) - hi_lo;                   // hi_lo * EPS <= 2^64 - 2^33 + 1
  u64 r;
  bool c = __builtin_add_overflow(t0, t1, &r);      // carry => r <= 2^64 - 2^33, so +EPS cannot carry again
  return r + (c ? GL_EPS : 0);
#endif
}
// hi*2^64 + lo with hi < 2^32
GLHD u64 gl_reduce96w(u64 lo, u64 hi) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef GL_REDUCE_VCC
  {
    u64 tv;
    u32 mv;
    const u32 ones = 0xFFFFFFFFu;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\t"
        "v_cndmask_b32_e32 %1, 0, %4, vcc"
        : "=&v"(tv), "=&v"(mv) : "v"((u32)hi), "v"(lo), "v"(ones) : "vcc");
    return tv + (u64)mv;
  }
#endif
  u64 t, c;
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=s"(c) : "v"((u32)hi), "v"(lo));
#ifndef GL_REDUCE_CARRYCHAIN
  {
    u32 m;
    asm("s_nop 1\n\t"
        "v_cndmask_b32 %0, 0, -1, %1" : "=v"(m) : "s"(c));
    return t + (u64)m;  // + c EPS cannot wrap (see gl_reduce128w): a carry-less 64-bit add
  }
#endif
  u32 t0 = (u32)t, t1 = (u32)(t >> 32), r0, r1, mc;
  asm("s_nop 1\n\t"  // 2 wait states between the mad's SGPR carry and its first VALU reader
      "v_cndmask_b32 %2, 0, -1, %5\n\t"
      "v_add_co_u32 %0, vcc, %3, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %4, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(mc)
      : "v"(t0), "v"(t1), "s"(c)
      : "vcc");
  return gl_mk(r0, r1);
#else
  u64 t1 = (hi << 32) - hi;
  u64 r;
  bool c = __builtin_add_overflow(lo, t1, &r);
  return r + (c ? GL_EPS : 0);
#endif
}
GLHD u64 gl_reduce128(u64 lo, u64 hi) { return gl_canon(gl_reduce128w(lo, hi)); }
GLHD void gl_mul_wide(u64 a, u64 b, u64& lo, u64& hi) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GL_MULWIDE_CARRY)
  // four independent 32x32->64 products (v_mad_u64_u32 with a zero addend), summed word by word with two carry chains:
  // no 64-bit addend has to be assembled from 32-bit halves (tools/ubench A/B)
  u64 p00 = (u64)(u32)a * (u32)b, x = (u64)(u32)a * (u32)(b >> 32), y = (u64)(u32)(a >> 32) * (u32)b, p11 = (u64)(u32)(a >> 32) * (u32)(b >> 32);
  u32 c0, c1, c2;
  u32 w1 = __builtin_addc((u32)(p00 >> 32), (u32)x, 0u, &c0);
  u32 w2 = __builtin_addc((u32)p11, (u32)(x >> 32), c0, &c1);
  u32 w3 = (u32)(p11 >> 32) + c1;
  w1 = __builtin_addc(w1, (u32)y, 0u, &c0);
  w2 = __builtin_addc(w2, (u32)(y >> 32), c0, &c2);
  w3 += c2;
  lo = gl_mk((u32)p00, w1);
  hi = gl_mk(w2, w3);
#elif defined(__HIP_DEVICE_COMPILE__)
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
// a * b + c as a 128-bit integer (any u64 a, b, c; the sum is below 2^128): the halves of c ride in the addend slots of the first two
// multiply-adds -- [c_lo, 0] where gl_mul_wide adds nothing, c_hi with the carry word of the first product (a0 b1 + 2^33 - 2 still
// fits 64 bits) -- so the addition costs no carry chain at all
GLHD void gl_mul_add_wide(u64 a, u64 b, u64 c, u64& lo, u64& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  u64 a0 = (u32)a, a1 = a >> 32, b0 = (u32)b, b1 = b >> 32;
  u64 p00 = a0 * b0 + (u64)(u32)c;
  u64 m1 = a0 * b1 + ((p00 >> 32) + (c >> 32));
  u64 m2 = a1 * b0 + (m1 & GL_EPS);
  lo = (m2 << 32) | (p00 & GL_EPS);
  hi = a1 * b1 + (m1 >> 32) + (m2 >> 32);
#else
  unsigned __int128 x = (unsigned __int128)a * b + c;
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
  u64 p0 = (u64)(u32)a * c;
  u64 p1 = (a >> 32) * c + (p0 >> 32);
  return gl_canon(gl_reduce96w((p1 << 32) | (p0 & GL_EPS), p1 >> 32));
}
// the same for any u64 a, as some u64 representative (no canonicalisation)
GLHD u64 gl_mul_small_w(u64 a, u32 c) {
  u64 p0 = (u64)(u32)a * c;
  u64 p1 = (a >> 32) * c + (p0 >> 32);
  return gl_reduce96w((p1 << 32) | (p0 & GL_EPS), p1 >> 32);
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

#if defined(__HIP_DEVICE_COMPILE__) && defined(GL_MUL_FUSED_TAIL)
// A/B of round 6 (variant builds only; MEASURED SLOWER, profiles/r06/ubench_fused_tail.txt: 2.55-2.67 against 2.83 G permutations/s --
// fewer instructions, 37.2 against 38.9 G wave-instructions per run, but 3.42 cycles each instead of 2.97: a carry that travels through an
// SGPR pair costs more than the move and the 64-bit add it replaces).
// lo + (g + w) 2^64 -> some u64 congruent to it, for g any u64 and w < 2^32 with g + w < 2^64 + 2^32 (round 6). What gl_mul_wide adds
// last to the high half of a product -- the high word w of the second middle term -- is not added at all: with s = g0 + w (carry k)
// and 2^32 EPS = -1 (mod p) the value is lo + s EPS - g1 - k, so w enters through ONE 32-bit add whose carry-out k is the carry-IN of
// the subtraction that takes g1 off anyway (v_subb_co_u32 with an SGPR-pair carry operand). That replaces a register move and a
// 64-bit add (2 + 4 SIMD cycles) by one carry-chain instruction (~3.2), and the mask of the multiply-add's carry comes from a
// subtract-with-borrow of a register from itself (~3.2) instead of a v_cndmask_b32 selecting by a scalar mask (4). Corrections as in
// gl_reduce128w: c = 1, b = 0: t <= 2^64 - 2^33, + EPS cannot wrap; c = 0, b = 1: g1 + k <= 2^32, so t - g1 - k + 2^64 >= p - 1 and
// - EPS cannot wrap. Hazards: k and c are VALU-written SGPR pairs read by later VALU instructions: two wait states (s_nop 1) stand
// in front of k's reader, three instructions in front of c's.
GLD u64 gl_reduce128w_split(u64 lo, u64 g, u32 w) {
  u32 g0 = (u32)g, g1 = (u32)(g >> 32), s;
  u64 k, t, c;
#ifdef GL_SPLIT_TAIL_SGPR
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(s), "=s"(k) : "v"(g0), "v"(w));
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=s"(c) : "v"(s), "v"(lo));
  u32 t0 = (u32)t, t1 = (u32)(t >> 32), r0, r1, mb, mc;
  asm("s_nop 1\n\t"
      "v_subb_co_u32_e64 %0, vcc, %4, %6, %8\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %0, %0, vcc\n\t"
      "v_subb_co_u32_e64 %3, vcc, %0, %0, %7"
      : "=&v"(r0), "=&v"(r1), "=&v"(mb), "=&v"(mc)
      : "v"(t0), "v"(t1), "v"(g1), "s"(c), "s"(k)
      : "vcc");
#else
  // variant B: k through an SGPR pair, the multiply-add's carry mask by v_cndmask_b32 as in gl_reduce128w, no explicit wait states in
  // front of k's reader (the multiply-add stands between its writer and its reader)
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(s), "=s"(k) : "v"(g0), "v"(w));
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=s"(c) : "v"(s), "v"(lo));
  u32 t0 = (u32)t, t1 = (u32)(t >> 32), r0, r1, mb, mc;
  asm(
#ifdef GL_SPLIT_TAIL_NOP
      "s_nop 0\n\t"
#endif
      "v_subb_co_u32_e64 %0, vcc, %4, %6, %8\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %0, %0, vcc\n\t"
      "v_cndmask_b32 %3, 0, -1, %7"
      : "=&v"(r0), "=&v"(r1), "=&v"(mb), "=&v"(mc)
      : "v"(t0), "v"(t1), "v"(g1), "s"(c), "s"(k)
      : "vcc");
#endif
  return gl_mk(r0, r1) + gl_mk(mc - mb, mb & ~mc);
}
#endif
// a * b + c as some u64 representative (any u64 a, b, c)
GLHD u64 gl_mul_addw(u64 a, u64 b, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GL_MUL_FUSED_TAIL)
  u64 a0 = (u32)a, a1 = a >> 32, b0 = (u32)b, b1 = b >> 32;
  u64 p00 = a0 * b0 + (u64)(u32)c;
  u64 m1 = a0 * b1 + ((p00 >> 32) + (c >> 32));
  u64 m2 = a1 * b0 + (m1 & GL_EPS);
  return gl_reduce128w_split((m2 << 32) | (p00 & GL_EPS), a1 * b1 + (m1 >> 32), (u32)(m2 >> 32));
#else
  u64 lo, hi;
  gl_mul_add_wide(a, b, c, lo, hi);
  return gl_reduce128w(lo, hi);
#endif
}
GLHD u64 gl_mul_add(u64 a, u64 b, u64 c) { return gl_canon(gl_mul_addw(a, b, c)); }  // canonical a b + c, any u64 inputs
GLHD u64 gl_mulw(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GL_MUL_FUSED_TAIL)
  u64 a0 = (u32)a, a1 = a >> 32, b0 = (u32)b, b1 = b >> 32;
  u64 p00 = a0 * b0;
  u64 m1 = a0 * b1 + (p00 >> 32);
  u64 m2 = a1 * b0 + (m1 & GL_EPS);
  return gl_reduce128w_split((m2 << 32) | (p00 & GL_EPS), a1 * b1 + (m1 >> 32), (u32)(m2 >> 32));
#else
  u64 lo, hi;
  gl_mul_wide(a, b, lo, hi);
  return gl_reduce128w(lo, hi);
#endif
}

// ---- sums of products without carry chains ------------------------------------------------------------------------------
// sum_k a_k b_k as four 64-bit columns of 32-bit words: on gfx950 a 64-bit add without carry-out costs 1.6 issue slots, an
// add_co / addc pair with its hazard nop 4.4. Fewer than 2^32 terms, so no column overflows.
struct gl_cols {
  u64 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  GLHD void add(u64 a, u64 b) {
    u64 pl, ph;
    gl_mul_wide(a, b, pl, ph);
    c0 += (u32)pl; c1 += pl >> 32; c2 += (u32)ph; c3 += ph >> 32;
  }
  // f times the product a b for a small factor f (a symmetric term counted twice, a wrapped term times z^5 = 3: f <= 6): the
  // factor multiplies the 32-bit words on their way into the columns (one multiply-add per column instead of one add)
  GLHD void add_scaled(u64 a, u64 b, u32 f) {
    u64 pl, ph;
    gl_mul_wide(a, b, pl, ph);
    c0 += (u64)(u32)pl * f; c1 += (pl >> 32) * f; c2 += (u64)(u32)ph * f; c3 += (ph >> 32) * f;
  }
  // the sum as a canonical element: lo + hi 2^64 + top 2^128, 2^128 = -2^32 (mod p)
  GLHD u64 value() const {
    const u64 w1 = c1 + (c0 >> 32), w2 = c2 + (w1 >> 32), w3 = c3 + (w2 >> 32);
    return gl_sub(gl_reduce128(gl_mk((u32)c0, (u32)w1), gl_mk((u32)w2, (u32)w3)), (w3 >> 32) << 32);
  }
};

// ---- quadratic extension ------------------------------------------------------------------
struct gl2 { u64 a, b; };
GLHD gl2 gl2_make(u64 a, u64 b) { gl2 r; r.a = a; r.b = b; return r; }
GLHD gl2 gl2_add(gl2 x, gl2 y) { return gl2_make(gl_add(x.a, y.a), gl_add(x.b, y.b)); }
GLHD gl2 gl2_sub(gl2 x, gl2 y) { return gl2_make(gl_sub(x.a, y.a), gl_sub(x.b, y.b)); }
GLHD gl2 gl2_mul(gl2 x, gl2 y) {
  // (a + b X)(c + d X) = ac + 7 bd + (ad + bc) X: the second product of each coefficient rides in the first one's addend slots
  return gl2_make(gl_mul_add(x.a, y.a, gl_mul_small_w(gl_mulw(x.b, y.b), 7)), gl_mul_add(x.a, y.b, gl_mulw(x.b, y.a)));
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
