// Width-12 permutations over Goldilocks for gfx950: Poseidon2 (the reference's default hasher,
// mp2-common/src/lib.rs:37-42) and Poseidon (WrapC, verifiable-db/src/api.rs:148), plus the
// plonky2 sponge conventions restated in-tree at mp2-common/src/hash.rs:24-45 and
// mp2-common/src/poseidon.rs:151-170 (overwrite-mode absorb, rate 8, squeeze from the front).
//
// One lane owns one 12-limb state (24 VGPRs). Round constants sit in __constant__ memory and
// are read with wave-uniform indices, so they arrive through the scalar cache into SGPRs and
// cost no vector registers. Rounds are loops, not unrolled: the external-round body is ~6 KB
// of ISA and stays resident in the instruction cache shared by neighbouring CUs.
#pragma once
#include "gl.cuh"
#include "perm_constants.h"

#define MP2G_POSEIDON2 0
#define MP2G_POSEIDON 1

#define c_p2_ext POSEIDON2_RC_EXT
#define c_p2_int POSEIDON2_RC_INT
#define c_p2_diag POSEIDON2_DIAG_M1
#define c_p_rc POSEIDON_RC

// ---- Poseidon2, weak-representative form ------------------------------------------------------
// State limbs are arbitrary u64 representatives between rounds (gl.cuh "weak"); outputs are
// canonicalised once at the end. Linear layers run on the 32-bit halves of the limbs in plain
// 64-bit arithmetic (all weights are small) and reduce once per output limb.
// x <- (x + rc)^7 ; x any u64, rc canonical
GLHD u64 p2_sbox(u64 x, u64 rc) {
  u64 t = gl_addw(x, rc);
  u64 t2 = gl_mulw(t, t), t4 = gl_mulw(t2, t2), t3 = gl_mulw(t, t2);
  return gl_mulw(t3, t4);
}
// M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] with additions only (HorizenLabs matmul_m4), on
// un-reduced integers: inputs < 2^32 give outputs < 2^37.
GLHD void p2_m4_plain(u64& a, u64& b, u64& c, u64& d) {
  u64 t0 = a + b, t1 = c + d;
  u64 t2 = (b << 1) + t1, t3 = (d << 1) + t0;
  u64 t4 = (t1 << 2) + t3, t5 = (t0 << 2) + t2;
  a = t3 + t5; b = t5; c = t2 + t4; d = t4;
}
GLHD void p2_external_half(u64 v[12]) {
  p2_m4_plain(v[0], v[1], v[2], v[3]);
  p2_m4_plain(v[4], v[5], v[6], v[7]);
  p2_m4_plain(v[8], v[9], v[10], v[11]);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u64 sum = v[i] + v[4 + i] + v[8 + i];
    v[i] += sum; v[4 + i] += sum; v[8 + i] += sum;  // < 2^39
  }
}
// circ(2 M4, M4, M4) on weak limbs; RC: also add the next round's constants rc[0..12) before the one
// reduction per limb (a carry into the top word instead of a separate weak addition per limb)
template <bool RC>
GLHD void p2_external_rc(u64 s[12], const u64* rc) {
  u64 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = s[i] >> 32; }
  p2_external_half(lo);
  p2_external_half(hi);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(P2_EXTERNAL_CARRY)
#pragma unroll
  for (int i = 0; i < 12; i++) {
    // value = lo + hi * 2^32 (+ rc), lo, hi < 2^39: carry-less 64-bit adds of the halves, the high part of the low half moves up
    u64 L = lo[i], H = hi[i];
    if (RC) { L += (u64)(u32)rc[i]; H += rc[i] >> 32; }
    H += L >> 32;  // < 2^41
    s[i] = gl_reduce96w(gl_mk((u32)L, (u32)H), H >> 32);
  }
  return;
#endif
#pragma unroll
  for (int i = 0; i < 12; i++) {
    // value = lo + hi * 2^32, lo, hi < 2^39
    u32 c0, c1, top = (u32)(hi[i] >> 32);
    u32 l0 = (u32)lo[i];
    u32 l1 = __builtin_addc((u32)(lo[i] >> 32), (u32)hi[i], 0u, &c0);
    top += c0;
    if (RC) {
      const u64 k = rc[i];
      l0 = __builtin_addc(l0, (u32)k, 0u, &c0);
      l1 = __builtin_addc(l1, (u32)(k >> 32), c0, &c1);
      top += c1;
    }
    s[i] = gl_reduce96w(gl_mk(l0, l1), top);
  }
}
GLHD void p2_external(u64 s[12]) { p2_external_rc<false>(s, nullptr); }
// x^7 for a limb whose round constant is already in (p2_external_rc)
GLHD u64 p2_sbox0(u64 t) {
  u64 t2 = gl_mulw(t, t), t4 = gl_mulw(t2, t2), t3 = gl_mulw(t, t2);
  return gl_mulw(t3, t4);
}
// s_i <- d_i s_i + sum_j s_j. Device code: sum reduced once, then a weak multiply and a weak add per limb (2.60 -> 2.71 G perm/s against the
// fused form below -- 128-bit product plus the 68-bit sum, one reduction --, whose carry chains cost more than the second reduction saves;
// -DP2_INTERNAL_FUSED / -DP2_EXTERNAL_CARRY restore the carry-chain forms for A/B runs)
GLHD void p2_internal(u64 s[12]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(P2_INTERNAL_FUSED)
  // the 68-bit sum from the 32-bit halves (each v_mad_u64_u32 adds a zero-extended word into a 64-bit accumulator: no carry
  // chains), reduced ONCE to a canonical element; every limb is then weak multiply + weak add
  u64 al = 0, ah = 0;
#ifdef P2_INTERNAL_MADSUM
  // A/B (round 6, variant builds only): every 32-bit word enters its 64-bit accumulator through the addend path of a multiply-add
  // by 1 -- one v_mad_u64_u32 (1.7 slots) instead of the two register moves + v_lshl_add_u64 (2.4 slots) hipcc makes of a
  // zero-extended 64-bit add; two accumulators per half keep the chains at six
  {
    u64 l0 = 0, l1 = 0, h0 = 0, h1 = 0, dmy;
#pragma unroll
    for (int i = 0; i < 12; i += 2) {
      asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(l0), "=s"(dmy) : "v"((u32)s[i]), "v"(l0));
      asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(h0), "=s"(dmy) : "v"((u32)(s[i] >> 32)), "v"(h0));
      asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(l1), "=s"(dmy) : "v"((u32)s[i + 1]), "v"(l1));
      asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(h1), "=s"(dmy) : "v"((u32)(s[i + 1] >> 32)), "v"(h1));
    }
    al = l0 + l1; ah = h0 + h1;
  }
#else
#pragma unroll
  for (int i = 0; i < 12; i++) {
    al += (u64)(u32)s[i];
    ah += s[i] >> 32;
  }
#endif
  ah += al >> 32;  // < 2^37
#ifdef P2_INTERNAL_ADDW
  const u64 sum = gl_canon(gl_reduce96w(gl_mk((u32)al, (u32)ah), ah >> 32));
#else
  const u64 sum = gl_reduce96w(gl_mk((u32)al, (u32)ah), ah >> 32);  // any representative will do below
#endif
#pragma unroll
#ifdef P2_INTERNAL_ADDW
  for (int i = 0; i < 12; i++) s[i] = gl_addw(gl_mulw(s[i], c_p2_diag[i]), sum);
#else
  for (int i = 0; i < 12; i++) s[i] = gl_mul_addw(s[i], c_p2_diag[i], sum);  // the sum rides in the product's addend slots
#endif
  return;
#endif
  u64 acc = s[0];
  u64 top = 0;
#pragma unroll
  for (int i = 1; i < 12; i++) {
    bool c = __builtin_add_overflow(acc, s[i], &acc);
    top += c ? 1 : 0;
  }
#pragma unroll
  for (int i = 0; i < 12; i++) {
    u64 lo, hi;
    gl_mul_wide(s[i], c_p2_diag[i], lo, hi);  // hi <= 2^64 - 2^32 - 1: adding top + carry cannot wrap
    bool c = __builtin_add_overflow(lo, acc, &lo);
    s[i] = gl_reduce128w(lo, hi + top + (c ? 1 : 0));
  }
}
#ifndef P2_UNROLL_EXT
#define P2_UNROLL_EXT 1
#endif
#ifndef P2_UNROLL_INT
#define P2_UNROLL_INT 11  // 22 internal rounds in two unrolled halves: 2.79 -> 2.84 G perm/s against no unrolling (tools/ubench; 4: 2.82, 22: 2.82)
#endif
#define P2_PRAGMA(x) _Pragma(#x)
#define P2_UNROLL(n) P2_PRAGMA(unroll n)
GLHD void poseidon2_perm(u64 s[12]) {
  p2_external_rc<true>(s, c_p2_ext);  // the constants of a full round ride on the preceding linear layer
P2_UNROLL(P2_UNROLL_EXT)
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = p2_sbox0(s[i]);
    if (r < 3) p2_external_rc<true>(s, c_p2_ext + 12 * (r + 1)); else p2_external(s);
  }
P2_UNROLL(P2_UNROLL_INT)
  for (int r = 0; r < 22; r++) {
    s[0] = p2_sbox(s[0], c_p2_int[r]);
    p2_internal(s);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_addw(s[i], c_p2_ext[48 + i]);  // round 4 follows an internal layer
P2_UNROLL(P2_UNROLL_EXT)
  for (int r = 4; r < 8; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = p2_sbox0(s[i]);
    if (r < 7) p2_external_rc<true>(s, c_p2_ext + 12 * (r + 1)); else p2_external(s);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}

// Two independent states through the permutation side by side (the two-sponges-per-lane experiment of merkle.hip: the 22 partial
// rounds are a chain of four dependent multiplications on one limb, and a second state gives the scheduler an independent chain)
GLHD void poseidon2_perm2(u64 s[12], u64 t[12]) {
  p2_external_rc<true>(s, c_p2_ext);
  p2_external_rc<true>(t, c_p2_ext);
P2_UNROLL(P2_UNROLL_EXT)
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) { s[i] = p2_sbox0(s[i]); t[i] = p2_sbox0(t[i]); }
    if (r < 3) { p2_external_rc<true>(s, c_p2_ext + 12 * (r + 1)); p2_external_rc<true>(t, c_p2_ext + 12 * (r + 1)); }
    else { p2_external(s); p2_external(t); }
  }
P2_UNROLL(P2_UNROLL_INT)
  for (int r = 0; r < 22; r++) {
    s[0] = p2_sbox(s[0], c_p2_int[r]);
    t[0] = p2_sbox(t[0], c_p2_int[r]);
    p2_internal(s);
    p2_internal(t);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) { s[i] = gl_addw(s[i], c_p2_ext[48 + i]); t[i] = gl_addw(t[i], c_p2_ext[48 + i]); }
P2_UNROLL(P2_UNROLL_EXT)
  for (int r = 4; r < 8; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) { s[i] = p2_sbox0(s[i]); t[i] = p2_sbox0(t[i]); }
    if (r < 7) { p2_external_rc<true>(s, c_p2_ext + 12 * (r + 1)); p2_external_rc<true>(t, c_p2_ext + 12 * (r + 1)); }
    else { p2_external(s); p2_external(t); }
  }
#pragma unroll
  for (int i = 0; i < 12; i++) { s[i] = gl_canon(s[i]); t[i] = gl_canon(t[i]); }
}

// Poseidon (WrapC), weak-representative form like Poseidon2 above. MDS: circulant
// [17,15,41,16,2,28,13,13,39,18,34,20] + diag [8,0,...]; all entries < 2^6, so the 32-bit halves of the
// limbs accumulate in u64 without overflow (< 2^41) and reduce once per row.
// RC: also add the next round's constants rc[0..12) before the one reduction per limb
template <bool RC>
GLHD void poseidon_mds_rc(u64 s[12], const u64* rc) {
  const u32 circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u64 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = s[i] >> 32; }
  u64 out[12];
#pragma unroll
  for (int r = 0; r < 12; r++) {
    u64 al = 0, ah = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
      al += lo[(i + r) % 12] * circ[i];
      ah += hi[(i + r) % 12] * circ[i];
    }
    if (r == 0) { al += lo[0] * 8; ah += hi[0] * 8; }
    // value = al + ah * 2^32, al, ah < 2^41
    u32 c0, c1, top = (u32)(ah >> 32);
    u32 l0 = (u32)al;
    u32 l1 = __builtin_addc((u32)(al >> 32), (u32)ah, 0u, &c0);
    top += c0;
    if (RC) {
      const u64 k = rc[r];
      l0 = __builtin_addc(l0, (u32)k, 0u, &c0);
      l1 = __builtin_addc(l1, (u32)(k >> 32), c0, &c1);
      top += c1;
    }
    out[r] = gl_reduce96w(gl_mk(l0, l1), top);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = out[i];
}
GLHD void poseidon_mds(u64 s[12]) { poseidon_mds_rc<false>(s, nullptr); }
GLHD void poseidon_perm(u64 s[12]) {
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_addw(s[i], c_p_rc[i]);
#pragma unroll 1
  for (int r = 0; r < 30; r++) {  // the constants of round r + 1 ride on round r's MDS reduction
    if (r < 4 || r >= 26) {
#pragma unroll
      for (int i = 0; i < 12; i++) s[i] = p2_sbox0(s[i]);
    } else {
      s[0] = p2_sbox0(s[0]);
    }
    if (r < 29) poseidon_mds_rc<true>(s, c_p_rc + 12 * (r + 1)); else poseidon_mds(s);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}
template <int VARIANT>
GLHD void perm(u64 s[12]) {
  if (VARIANT == MP2G_POSEIDON2) poseidon2_perm(s); else poseidon_perm(s);
}
// compress(l, r) = perm(l || r || 0)[0..4]   (plonky2 hashing.rs)
template <int VARIANT>
GLD void two_to_one(const u64 l[4], const u64 r[4], u64 out[4]) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 4; i++) { s[i] = l[i]; s[4 + i] = r[i]; s[8 + i] = 0; }
  perm<VARIANT>(s);
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = s[i];
}
