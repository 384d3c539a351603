// Lane-cooperative Poseidon2 for the latency-bound corners of the prover (Fiat-Shamir duplexes, the
// top levels of Merkle trees): the 12 limbs of ONE state sit in lanes 0..11 of an aligned 16-lane
// group, four states per wave. A lone lane needs ~24 k dependent VALU instructions per permutation
// (~45 us: one wave issues one instruction per ~4 cycles whatever its ILP); spread over 12 lanes the
// S-boxes of a full round run in parallel and the linear layers become a handful of cross-lane
// exchanges (ds_bpermute through __shfl), ~6 k instructions per permutation. Throughput per lane is
// worse (12 of 16 lanes busy, partial rounds keep 11 lanes idle during the S-box), so the
// one-lane-per-state form in poseidon.cuh stays the workhorse wherever there are >= ~10^5 states.
//
// Same function as poseidon2_perm (poseidon.cuh): limbs are weak representatives inside, canonical out.
#pragma once
#include "poseidon.cuh"

namespace mp2g {

// value of `v` held by lane `src` (0..15) of this lane's 16-lane group
GLD u64 wp_shfl(u64 v, int src) { return __shfl(v, src, 16); }

// Cross-lane moves as DPP operands (a 16-lane group = one DPP row): no LDS round trip. A ds_bpermute costs an LDS instruction and
// its latency (~100 cycles) per 32-bit half, and a permutation needs ~580 of them (tools/dbg/lone_proof.sh: ~15 us per permutation,
// most of it these); a DPP move is one VALU instruction. quad_perm serves the exchanges inside a quad (one M4 block), row_ror the
// rotations by whole quads. -DWP_BPERMUTE restores the __shfl forms for A/B runs.
template <int CTRL> GLD u64 wp_dpp(u64 v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(u32)v, CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), CTRL, 0xF, 0xF, false);
  return gl_mk((u32)lo, (u32)hi);
}
#define WP_QUAD(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define WP_ROR(n) (0x120 + (n))  // lane i takes the value of lane (i - n) mod 16: WP_ROR(16 - k) reads lane i + k

// external layer circ(2 M4, M4, M4) on lanes: l = lane & 15 (< 12 meaningful)
GLD u64 wp2_external(u64 x, int l) {
  const int r = l & 3;
  u64 a[4];
#ifdef WP_BPERMUTE
  const int cb = l & 12;
#pragma unroll
  for (int j = 0; j < 4; j++) a[j] = wp_shfl(x, cb + j);
#else
  a[0] = wp_dpp<WP_QUAD(0, 0, 0, 0)>(x); a[1] = wp_dpp<WP_QUAD(1, 1, 1, 1)>(x);
  a[2] = wp_dpp<WP_QUAD(2, 2, 2, 2)>(x); a[3] = wp_dpp<WP_QUAD(3, 3, 3, 3)>(x);
#endif
  // row r of M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
  const u32 c0 = r == 0 ? 5u : (r == 1 ? 4u : 1u);
  const u32 c1 = r == 0 ? 7u : (r == 1 ? 6u : (r == 2 ? 3u : 1u));
  const u32 c2 = r < 2 ? 1u : (r == 2 ? 5u : 4u);
  const u32 c3 = r == 0 ? 3u : (r == 1 ? 1u : (r == 2 ? 7u : 6u));
  u64 tl = (u64)(u32)a[0] * c0 + (u64)(u32)a[1] * c1 + (u64)(u32)a[2] * c2 + (u64)(u32)a[3] * c3;
  u64 th = (a[0] >> 32) * c0 + (a[1] >> 32) * c1 + (a[2] >> 32) * c2 + (a[3] >> 32) * c3;  // each < 2^36
  // same row of the two other chunks
#ifdef WP_BPERMUTE
  const int s1 = l < 8 ? l + 4 : l - 8, s2 = l < 4 ? l + 8 : l - 4;
  u64 yl = 2 * tl + wp_shfl(tl, s1 & 15) + wp_shfl(tl, s2 & 15);
  u64 yh = 2 * th + wp_shfl(th, s1 & 15) + wp_shfl(th, s2 & 15);  // < 2^39
#else
  // with the fourth quad (lanes 12..15) contributing zero, the sum over the other chunks' row is the sum over ALL quads' row
  // minus the lane's own: three rotations by whole quads
  const u64 zl = l < 12 ? tl : 0, zh = l < 12 ? th : 0;
  u64 yl = 2 * tl + wp_dpp<WP_ROR(4)>(zl) + wp_dpp<WP_ROR(8)>(zl) + wp_dpp<WP_ROR(12)>(zl);
  u64 yh = 2 * th + wp_dpp<WP_ROR(4)>(zh) + wp_dpp<WP_ROR(8)>(zh) + wp_dpp<WP_ROR(12)>(zh);  // < 2^39
#endif
  u64 lo;
  bool c = __builtin_add_overflow(yl, yh << 32, &lo);
  return gl_reduce96w(lo, (yh >> 32) + (c ? 1 : 0));
}
// internal layer: x_l <- d_l x_l + sum_j x_j
GLD u64 wp2_internal(u64 x, int l, u64 d) {
  // 68-bit sum of the 12 lanes as (lo64, top): quad butterflies, then the two other quads
  u64 lo = l < 12 ? x : 0, top = 0;
#ifdef WP_BPERMUTE
#pragma unroll
  for (int m = 1; m <= 2; m <<= 1) {
    u64 olo = wp_shfl(lo, l ^ m), otop = wp_shfl(top, l ^ m);
    bool c = __builtin_add_overflow(lo, olo, &lo);
    top += otop + (c ? 1 : 0);
  }
  {
    const int s1 = (l + 4) & 15, s2 = (l + 8) & 15;  // lanes 12..15 hold zeros: any rotation of the 4 quads sums all of them
    const int s3 = (l + 12) & 15;
    u64 l1 = wp_shfl(lo, s1), t1 = wp_shfl(top, s1), l2 = wp_shfl(lo, s2), t2 = wp_shfl(top, s2);
    u64 l3 = wp_shfl(lo, s3), t3 = wp_shfl(top, s3);
    bool c1 = __builtin_add_overflow(lo, l1, &lo);
    bool c2 = __builtin_add_overflow(lo, l2, &lo);
    bool c3 = __builtin_add_overflow(lo, l3, &lo);
    top += t1 + t2 + t3 + (c1 ? 1 : 0) + (c2 ? 1 : 0) + (c3 ? 1 : 0);
  }
#else
  // the sum as two 64-bit columns of 32-bit halves (carry-free: 12 terms < 2^32 each): quad butterflies and quad rotations by DPP
  u64 sl = (u32)lo, sh = lo >> 32;
  sl += wp_dpp<WP_QUAD(1, 0, 3, 2)>(sl); sh += wp_dpp<WP_QUAD(1, 0, 3, 2)>(sh);
  sl += wp_dpp<WP_QUAD(2, 3, 0, 1)>(sl); sh += wp_dpp<WP_QUAD(2, 3, 0, 1)>(sh);
  sl += wp_dpp<WP_ROR(4)>(sl) + wp_dpp<WP_ROR(8)>(sl) + wp_dpp<WP_ROR(12)>(sl);  // lanes 12..15 hold zeros
  sh += wp_dpp<WP_ROR(4)>(sh) + wp_dpp<WP_ROR(8)>(sh) + wp_dpp<WP_ROR(12)>(sh);
  sh += sl >> 32;  // < 2^37
  lo = gl_mk((u32)sl, (u32)sh);
  top = sh >> 32;
#endif
  u64 plo, phi;
  gl_mul_wide(x, d, plo, phi);
  bool c = __builtin_add_overflow(plo, lo, &plo);
  return gl_reduce128w(plo, phi + top + (c ? 1 : 0));
}
// one permutation of the state spread over lanes 0..11 of the group; every lane of the group must call.
GLD u64 wp2_perm(u64 x, int l) {
  const int li = l < 12 ? l : 0;
  u64 rc[8];
#pragma unroll
  for (int r = 0; r < 8; r++) rc[r] = c_p2_ext[12 * r + li];
  const u64 d = c_p2_diag[li];
  x = wp2_external(x, l);
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    u64 k = r == 0 ? rc[0] : (r == 1 ? rc[1] : (r == 2 ? rc[2] : rc[3]));
    x = wp2_external(p2_sbox(x, k), l);
  }
#pragma unroll 1
  for (int r = 0; r < 22; r++) {
    u64 y = p2_sbox(x, c_p2_int[r]);
    x = wp2_internal(l == 0 ? y : x, l, d);
  }
#pragma unroll 1
  for (int r = 4; r < 8; r++) {
    u64 k = r == 4 ? rc[4] : (r == 5 ? rc[5] : (r == 6 ? rc[6] : rc[7]));
    x = wp2_external(p2_sbox(x, k), l);
  }
  return gl_canon(x);
}
}  // namespace mp2g
