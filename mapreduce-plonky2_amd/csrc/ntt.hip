// Batched radix-2 NTT / LDE over Goldilocks for gfx950.
//
// Replaces [dep] plonky2_field fft.rs (fft / ifft / coset_fft) and the per-polynomial
// `lde(rate_bits).coset_fft(g)` loop of plonky2 fri/oracle.rs PolynomialBatch::from_coeffs, as
// reached from recursion-framework/src/circuit_builder.rs:308 and wrap_circuit.rs:143.
//
// Structure (16 B of algorithmic traffic per point; in practice the butterflies' instruction stream is the limit):
//   * n <= 2^13 : one launch; a block owns 4096 points in LDS (several small transforms per block; one 8192-point
//     transform of 1024 lanes at 2^13), every global access is a contiguous 8 B/lane stream.
//   * n  > 2^13 : Cooley-Tukey n = n1*n2 in two launches. Pass A transforms the strided
//     dimension for a tile of 2^LC adjacent columns (>= 32..128 B contiguous per row), multiplies
//     by w_n^(i2*k1) (streamed from a precomputed table: one multiply per point) and leaves row j
//     in DIF order; pass B transforms contiguous rows in place.
//   * inside a block every lane keeps 8 points in registers between LDS exchanges and runs a true radix-8
//     DIF butterfly on them: the twiddles inside the 8-point DFT are powers of w_8 = 2^24 (plonky2's
//     POWER_OF_TWO_GENERATOR gives w_64 = 8), i.e. shifts plus a short reduction instead of general
//     multiplications, and the 7 general twiddles w^(E r) are applied once per output; the last round
//     has no general twiddle at all. (radix-16 halves the LDS round trips but also the waves per tile
//     and measured 15 % slower.) LDS indices are padded by 1/16 so the strided rounds are bank-conflict
//     free; the twiddles of all rounds but the first are staged in LDS once per block.
//   * a tile has ONE block barrier (two at 2^13): the first round reads global memory directly, every later round stays
//     inside the 512 points a wave owns ("one-barrier tiles" below; the barrier-per-round kernels remain for the shapes
//     that do not fit that scheme and for A/B runs with MP2G_NTT_V1=1).
//   * LDE: the 2^r cosets of the blown-up domain are 2^r independent size-n transforms of the
//     same coefficients scaled by (g w_{N}^j)^i; outputs land bit-reversed, i.e. already in
//     Merkle-leaf order, so there is no separate transpose / bit-reverse pass.
#include "ntt.h"
#include <cstdlib>

namespace mp2g {

__device__ __forceinline__ int lds_pad(int a) { return a + (a >> 4); }

struct NttArgs {
  const u64* in;
  u64* out;
  u64 in_poly_stride, out_poly_stride;
  u32 logK;              // cosets per polynomial (batch index b -> poly b>>logK, coset b&(K-1))
  u32 log_n, log_n1, log_n2;
  u32 batch;             // polys << logK
  const u64* tw;         // inner-DFT twiddles w_T^k, k < T/2 (this pass)
  const u64* tw4_lo;     // w_n^e, e < 4096
  const u64* tw4_hi;     // w_n^(e << 12)
  const u64* pre_lo;     // [K][n2] (or [K][n] when single pass)
  const u64* pre_hi;     // [K][n1]
  const u64* tw4_full;   // [n1][n2] in DIF row order: row j holds w_n^(i2 * bitrev(j)); null -> two-level tables
  const u64* pre_full;   // [K][n]: s_j^i; null -> pre_lo * pre_hi
  u64 post;              // scalar multiplied on the final store (n^-1 for inverse), 0 = none
  u32 bitrev_out;
  u32 inverse;           // tables hold powers of the inverse root: the w_8 constants inside a butterfly follow
  const u64* tc;         // shift-twiddle scheme (ShiftGeom): [64][2^(LT-6)] merged twiddles w_T^(b (r + 8 r')) of this pass; null = classic tables
};

// lanes per block: one radix-2^NTT_RMAX item per lane and round
// 8 waves per SIMD (64 VGPRs) for the one-barrier kernels unless built with -DNTT_WAVES_ATTR= (tools/dbg A/B)
#ifndef NTT_WAVES_ATTR
#define NTT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(8)))
#endif
// NTT_DBG (tools/dbg only): 1 = no butterflies (memory phases alone), 2 = no global traffic (butterflies alone)
#ifndef NTT_DBG
#define NTT_DBG 0
#endif
#ifndef NTT_RMAX
#define NTT_RMAX 3
#endif
template <int LT, int LW> struct NttGeom {
  static constexpr int E = (1 << LT) << LW;
  static constexpr int NT = (E >> NTT_RMAX) < 64 ? 64 : ((E >> NTT_RMAX) > 1024 ? 1024 : (E >> NTT_RMAX));
};
// Twiddles of the inner DFT, one table per radix-8 round. Round rho works on index bits HI = LT-1-3 rho
// .. HI-2 of the transform; a lane's 8 points share the low bits `below` (< 2^LO, LO = HI-2), and output m
// of its butterfly needs w_T^(E r), E = below << (3 rho), r = bitrev3(m) = 1..7. Table rho holds
// [r-1][below] at offset off(rho); rounds with LO = 0 (the last one) need none. Lanes of a wave walk
// consecutive `below`, so both the LDS copy and the global table of round 0 (read through L1 for
// T >= 2^11, where LDS is needed for the points) are conflict-free / coalesced.
template <int LT> struct R8Tw {
  static constexpr int lo(int rho) { return LT - 3 - 3 * rho; }
  static constexpr int n_tables() { int r = 0; while (lo(r) >= 1) r++; return r; }
  static constexpr int off(int rho) { int o = 0; for (int q = 0; q < rho; q++) o += 7 << lo(q); return o; }
  static constexpr int TOTAL = off(n_tables());
  static constexpr int GLOBAL_ROUNDS = (LT >= 11 && n_tables() >= 1) ? 1 : 0;
  static constexpr int LDS_OFF = off(GLOBAL_ROUNDS);
  static constexpr int LDS_WORDS = TOTAL - LDS_OFF;
};
// x * 2^24, 2^48, 2^72 (= w_8, w_8^2, w_8^3), canonical in and out
__device__ __forceinline__ u64 gl_mul_2p24(u64 x) { return gl_canon(gl_reduce96w(x << 24, x >> 40)); }
__device__ __forceinline__ u64 gl_mul_2p48(u64 x) { return gl_reduce128(x << 48, x >> 16); }
__device__ __forceinline__ u64 gl_mul_2p72(u64 x) {
  // x 2^72 = (x << 8) 2^64 with x << 8 = t_hi 2^64 + t_lo, and 2^128 = -2^32 (mod p)
  return gl_sub(gl_reduce128(0, x << 8), (x >> 56) << 32);
}
template <int K> __device__ __forceinline__ u64 gl_mul_w8(u64 x) {  // x * w_8^K, K = 1..3
  return K == 1 ? gl_mul_2p24(x) : (K == 2 ? gl_mul_2p48(x) : gl_mul_2p72(x));
}
// (u - v) * 2^48 and (u - v) * 2^72 without canonicalising the difference first: with s = u - v mod 2^64 and the borrow b, the
// difference is s - b 2^64, and -2^64 2^48 = -2^112 = +2^16, -2^64 2^72 = -2^136 = +2^40 (mod p, 2^96 = -1): the borrow becomes one
// bit of the low word that the shift leaves empty (saves the second subtract chain of gl_sub)
__device__ __forceinline__ u64 gl_sub_mul_2p48(u64 u, u64 v) {
  u32 c0, c1;
  const u32 s0 = __builtin_subc((u32)u, (u32)v, 0u, &c0);
  const u32 s1 = __builtin_subc((u32)(u >> 32), (u32)(v >> 32), c0, &c1);
  const u64 s = gl_mk(s0, s1);
  return gl_reduce128(gl_mk(c1 ? 0x10000u : 0u, s0 << 16), s >> 16);
}
__device__ __forceinline__ u64 gl_sub_mul_2p72(u64 u, u64 v) {
  u32 c0, c1;
  const u32 s0 = __builtin_subc((u32)u, (u32)v, 0u, &c0);
  const u32 s1 = __builtin_subc((u32)(u >> 32), (u32)(v >> 32), c0, &c1);
  const u64 s = gl_mk(s0, s1);
  return gl_sub(gl_reduce128(gl_mk(0u, c1 ? 0x100u : 0u), s << 8), (s >> 56) << 32);
}
template <int K> __device__ __forceinline__ u64 gl_sub_mul_w8(u64 u, u64 v) {  // (u - v) * w_8^K
#ifdef NTT_UNFUSED_SUB
  return gl_mul_w8<K>(gl_sub(u, v));
#else
  return K == 1 ? gl_mul_2p24(gl_sub(u, v)) : (K == 2 ? gl_sub_mul_2p48(u, v) : gl_sub_mul_2p72(u, v));
#endif
}
// (u - v) * w_8^K for the forward transform, (u - v) * w_8^-K = (v - u) * w_8^(4-K) for the inverse (w_8^4 = -1)
template <int K> __device__ __forceinline__ u64 bfly_lo(u64 u, u64 v, bool inverse) {
  return inverse ? gl_sub_mul_w8<4 - K>(v, u) : gl_sub_mul_w8<K>(u, v);
}
// x * 2^S (mod p) for a compile-time S in [0, 192), canonical in and out: 2^96 = -1, 2^64 = 2^32 - 1. POWER_OF_TWO_GENERATOR gives
// w_64 = 2^3, so every twiddle of a sub-transform of at most 64 points is such a shift (12-18 issue slots against 29 for a general
// multiplication with a table load in front of it)
template <int S> __device__ __forceinline__ u64 gl_mul_2pow(u64 x) {
  static_assert(S >= 0 && S < 192, "exponent mod 192");
  if constexpr (S == 0) return x;
  else if constexpr (S >= 96) return gl_neg(gl_mul_2pow<S - 96>(x));
  else if constexpr (S <= 32) return gl_canon(gl_reduce96w(x << S, x >> (64 - S)));
  else if constexpr (S < 64) return gl_reduce128(x << S, x >> (64 - S));
  else if constexpr (S == 64) return gl_reduce128(0, x);
  else {  // x 2^S = (x << K) 2^64 with x << K = h 2^64 + t, and 2^128 = -2^32: h < 2^31, so h 2^32 is canonical
    constexpr int K = S - 64;
    return gl_sub(gl_reduce128(0, x << K), (x >> (64 - K)) << 32);
  }
}
// Shift-twiddle scheme (one-barrier kernels, ShiftGeom below). The twiddle after the first radix-8 round is w_T^(j r) with j the
// L0 = LT - 3 index bits still to transform and r the round's output frequency. Split j = a 2^(L0-3) + b (a = its top three bits):
//   w_T^(j r) = w_64^(a r) * w_T^(b r).
// The first factor is a power of two -- and `a` is the SAME for all lanes of a wave in these tile shapes, so a scalar switch picks
// code with compile-time shift amounts; the second factor does not depend on the bits the NEXT round transforms (a), so it moves
// behind that round and merges with its own twiddle w_{T/8}^(b r') into ONE table entry w_T^(b (r + 8 r')) (args.tc, [64][2^(L0-3)],
// row r + 8 r'; r is wave-uniform in the second round: a wave owns one output block of the first). Two table multiplications
// per point become one shift and one table multiplication.
template <int A, bool INV, int M> __device__ __forceinline__ u64 shift_tw1(u64 x) {
  constexpr int R = ((M & 1) << 2) | (M & 2) | (M >> 2);
  constexpr int E = (A * R) & 63;
  return gl_mul_2pow<3 * (INV ? (64 - E) & 63 : E)>(x);
}
template <int A, bool INV> __device__ __forceinline__ void shift_tw(u64* x) {
  x[1] = shift_tw1<A, INV, 1>(x[1]); x[2] = shift_tw1<A, INV, 2>(x[2]); x[3] = shift_tw1<A, INV, 3>(x[3]); x[4] = shift_tw1<A, INV, 4>(x[4]);
  x[5] = shift_tw1<A, INV, 5>(x[5]); x[6] = shift_tw1<A, INV, 6>(x[6]); x[7] = shift_tw1<A, INV, 7>(x[7]);
}
// a: wave-uniform (the caller passes it through readfirstlane, so this is a scalar branch, not eight masked passes)
template <bool INV> __device__ __forceinline__ void shift_twiddles_dir(u64* x, int a) {
  switch (a) {
    case 1: shift_tw<1, INV>(x); break;
    case 2: shift_tw<2, INV>(x); break;
    case 3: shift_tw<3, INV>(x); break;
    case 4: shift_tw<4, INV>(x); break;
    case 5: shift_tw<5, INV>(x); break;
    case 6: shift_tw<6, INV>(x); break;
    case 7: shift_tw<7, INV>(x); break;
    default: break;
  }
}
__device__ __forceinline__ void shift_twiddles(u64* x, int a, bool inverse) {
  if (inverse) shift_twiddles_dir<true>(x, a); else shift_twiddles_dir<false>(x, a);
}

// the 2^R-point DIF butterfly of one lane (R = 3: true radix-8 with the w_8 shifts; R < 3: the last, partial round) followed by
// the round's general twiddles w_T^(E r), r = bitrev3(m), looked up at [r-1][below]
// MERGED: the second round of the shift-twiddle scheme -- twg = the wave's rows of args.tc (row r of the first round's output
// block), every output m (0 included) is multiplied by twg[(8 r'(m)) << LO | below]
template <int LT, int HI, int R, bool TWG = false, bool MERGED = false>
__device__ __forceinline__ void butterfly(u64* x, int below, const u64* tw, const u64* __restrict__ twg, bool inverse) {
  constexpr int LO = HI - R + 1;
  constexpr int RHO = (LT - 1 - HI) / 3;
  static_assert(R == 3 || LO == 0, "a partial round can only be the last one");
  if constexpr (R == 3) {
    // stage 0: pairs (m, m + 4), lower output times w_8^m
    { u64 u = x[0], v = x[4]; x[0] = gl_add(u, v); x[4] = gl_sub(u, v); }
    { u64 u = x[1], v = x[5]; x[1] = gl_add(u, v); x[5] = bfly_lo<1>(u, v, inverse); }
    { u64 u = x[2], v = x[6]; x[2] = gl_add(u, v); x[6] = bfly_lo<2>(u, v, inverse); }
    { u64 u = x[3], v = x[7]; x[3] = gl_add(u, v); x[7] = bfly_lo<3>(u, v, inverse); }
    // stage 1: pairs (m, m + 2) inside each half, lower output times w_4^(m & 1)
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
      { u64 u = x[h], v = x[h + 2]; x[h] = gl_add(u, v); x[h + 2] = gl_sub(u, v); }
      { u64 u = x[h + 1], v = x[h + 3]; x[h + 1] = gl_add(u, v); x[h + 3] = bfly_lo<2>(u, v, inverse); }
    }
    // stage 2: pairs (m, m + 1). A sum that goes straight into a general twiddle multiplication (LO > 0, m > 0) may stay a weak
    // representative: gl_mul takes any u64 and returns the canonical product
#pragma unroll
    for (int m = 0; m < 8; m += 2) {
      u64 u = x[m], v = x[m + 1];
      x[m] = (MERGED || (LO > 1 && m > 0)) ? gl_addw(u, v) : gl_add(u, v);  // weak only where a table multiplication follows
      x[m + 1] = gl_sub(u, v);
    }
    if constexpr (MERGED) {
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int r = ((m & 1) << 2) | (m & 2) | (m >> 2);
        x[m] = gl_mul(x[m], twg[((8 * r) << LO) + below]);
      }
    } else if constexpr (LO == 1) {
      // sub-transforms of 16 points: w_16 = 2^12, the twiddle of output m is 2^(12 r) for the odd `below`, 1 for the even one
      if (inverse) {
        x[1] = below ? gl_mul_2pow<192 - 48>(x[1]) : x[1]; x[2] = below ? gl_mul_2pow<192 - 24>(x[2]) : x[2]; x[3] = below ? gl_mul_2pow<192 - 72>(x[3]) : x[3];
        x[4] = below ? gl_mul_2pow<192 - 12>(x[4]) : x[4]; x[5] = below ? gl_mul_2pow<192 - 60>(x[5]) : x[5]; x[6] = below ? gl_mul_2pow<192 - 36>(x[6]) : x[6];
        x[7] = below ? gl_mul_2pow<192 - 84>(x[7]) : x[7];
      } else {
        x[1] = below ? gl_mul_2pow<48>(x[1]) : x[1]; x[2] = below ? gl_mul_2pow<24>(x[2]) : x[2]; x[3] = below ? gl_mul_2pow<72>(x[3]) : x[3];
        x[4] = below ? gl_mul_2pow<12>(x[4]) : x[4]; x[5] = below ? gl_mul_2pow<60>(x[5]) : x[5]; x[6] = below ? gl_mul_2pow<36>(x[6]) : x[6];
        x[7] = below ? gl_mul_2pow<84>(x[7]) : x[7];
      }
    } else if constexpr (LO > 0) {
      const u64* tab = (TWG || RHO < R8Tw<LT>::GLOBAL_ROUNDS) ? twg + R8Tw<LT>::off(RHO) : tw + (R8Tw<LT>::off(RHO) - R8Tw<LT>::LDS_OFF);
#pragma unroll
      for (int m = 1; m < 8; m++) {
        const int r = ((m & 1) << 2) | (m & 2) | (m >> 2);
        x[m] = gl_mul(x[m], tab[((r - 1) << LO) + below]);
      }
    }
  } else if constexpr (R == 2) {
    { u64 u = x[0], v = x[2]; x[0] = gl_add(u, v); x[2] = gl_sub(u, v); }
    { u64 u = x[1], v = x[3]; x[1] = gl_add(u, v); x[3] = bfly_lo<2>(u, v, inverse); }
    { u64 u = x[0], v = x[1]; x[0] = gl_add(u, v); x[1] = gl_sub(u, v); }
    { u64 u = x[2], v = x[3]; x[2] = gl_add(u, v); x[3] = gl_sub(u, v); }
  } else {
    u64 u = x[0], v = x[1]; x[0] = gl_add(u, v); x[1] = gl_sub(u, v);
  }
}
template <int LT, int HI, int R, bool COLS, int LW, int NTT_THREADS>
__device__ __forceinline__ void dif_round(u64* s, const u64* tw, const u64* __restrict__ twg, int tid, bool inverse) {
  constexpr int T = 1 << LT, LO = HI - R + 1, W = 1 << LW;
  constexpr int ITEMS = W << (LT - R);
  for (int item = tid; item < ITEMS; item += NTT_THREADS) {
    int c, rest;
    if (COLS) { c = item & (W - 1); rest = item >> LW; }
    else { rest = item & ((T >> R) - 1); c = item >> (LT - R); }
    int below = rest & ((1 << LO) - 1), above = rest >> LO;
    int j0 = (above << (HI + 1)) | below;
    u64 x[1 << R];
#pragma unroll
    for (int m = 0; m < (1 << R); m++) {
      int j = j0 + (m << LO);
      x[m] = s[lds_pad(COLS ? (j << LW) + c : (c << LT) + j)];
    }
    butterfly<LT, HI, R>(x, below, tw, twg, inverse);
#pragma unroll
    for (int m = 0; m < (1 << R); m++) {
      int j = j0 + (m << LO);
      s[lds_pad(COLS ? (j << LW) + c : (c << LT) + j)] = x[m];
    }
  }
}
template <int LT, int HI, bool COLS, int LW, int NTT_THREADS>
__device__ __forceinline__ void dif_all(u64* s, const u64* tw, const u64* __restrict__ twg, int tid, bool inverse) {
  if constexpr (HI >= 0) {
    constexpr int R = (HI + 1 >= 3) ? 3 : HI + 1;
    dif_round<LT, HI, R, COLS, LW, NTT_THREADS>(s, tw, twg, tid, inverse);
    __syncthreads();
    dif_all<LT, HI - R, COLS, LW, NTT_THREADS>(s, tw, twg, tid, inverse);
  }
}

__device__ __forceinline__ const u64* in_base(const NttArgs& a, u32 b) {
  return a.in + (u64)(b >> a.logK) * a.in_poly_stride;
}
__device__ __forceinline__ u64* out_base(const NttArgs& a, u32 b) {
  u32 coset = b & ((1u << a.logK) - 1);
  return a.out + (u64)(b >> a.logK) * a.out_poly_stride + ((u64)bitrev32(coset, a.logK) << a.log_n);
}

// (A persistent, register-prefetching variant of these kernels was measured in round 1 and dropped: the 16
// staged points per lane cost 32 VGPRs, pushed the kernels into spills and ran 15-35 % slower.)

// ---- pass B / single pass: contiguous rows of length T = 2^LT, 2^LW rows per block ----------
template <int LT, int LW>
__global__ void __launch_bounds__((NttGeom<LT, LW>::NT)) ntt_rows_kernel(NttArgs a) {
  constexpr int T = 1 << LT, E = T << LW, NT = NttGeom<LT, LW>::NT;
  extern __shared__ __align__(16) u64 smem[];
  u64* s = smem;
  u64* tw = smem + lds_pad(E) + 1;
  const int tid = threadIdx.x;
  for (int i = tid; i < R8Tw<LT>::LDS_WORDS; i += NT) tw[i] = a.tw[R8Tw<LT>::LDS_OFF + i];
  const u32 n1 = 1u << a.log_n1;
  const u64 total_rows = (u64)a.batch << a.log_n1;
  const u64 row0 = (u64)blockIdx.x << LW;
  const bool two_pass = a.log_n1 != 0;
  for (int e = tid; e < E; e += NT) {
    int r = e >> LT, j = e & (T - 1);
    u64 g = row0 + r;
    u64 v = 0;
    if (g < total_rows) {
      u32 b = (u32)(g >> a.log_n1), jr = (u32)(g & (n1 - 1));
      if (two_pass) {
#if NTT_DBG == 2
        v = (u64)e * 0x9E3779B97F4A7C15ull + jr;
#else
        v = out_base(a, b)[((u64)jr << LT) + j];  // pass A left row jr in place
#endif
      } else {
        v = in_base(a, b)[j];
        if (a.pre_lo) v = gl_mul(v, a.pre_lo[((u64)(b & ((1u << a.logK) - 1)) << LT) + j]);
      }
    }
    s[lds_pad(e)] = v;
  }
  __syncthreads();
#if NTT_DBG != 1
  dif_all<LT, LT - 1, false, LW, NT>(s, tw, a.tw, tid, a.inverse != 0);
#endif
  for (int e = tid; e < E; e += NT) {
    int r = e >> LT, p = e & (T - 1);
    u64 g = row0 + r;
    if (g >= total_rows) continue;
    u32 b = (u32)(g >> a.log_n1), jr = (u32)(g & (n1 - 1));
    u64 v = a.bitrev_out ? s[lds_pad(e)] : s[lds_pad((r << LT) + (int)bitrev32((u32)p, LT))];
    if (a.post) v = gl_mul(v, a.post);
#if NTT_DBG == 2
    if (v == 0x123456789ull)
#endif
    out_base(a, b)[((u64)jr << LT) + p] = v;
  }
}
// Natural-order pass B of a two-pass transform: X[k1 + n1*k2]. A tile takes the rows
// jr = (r << lo_bits) | jr_lo, r = 0..W-1 of the dense scratch buffer, whose k1 = bitrev(jr) share
// their high bits, so each k2 yields W*8 B of contiguous output.
template <int LT, int LW>
__global__ void __launch_bounds__((NttGeom<LT, LW>::NT)) ntt_rows_nat_kernel(NttArgs a) {
  constexpr int T = 1 << LT, W = 1 << LW, E = T << LW, NT = NttGeom<LT, LW>::NT;
  extern __shared__ __align__(16) u64 smem[];
  u64* s = smem;
  u64* tw = smem + lds_pad(E) + 1;
  const int tid = threadIdx.x;
  for (int i = tid; i < R8Tw<LT>::LDS_WORDS; i += NT) tw[i] = a.tw[R8Tw<LT>::LDS_OFF + i];
  const u32 lo_bits = a.log_n1 - LW;
  const u32 tile = blockIdx.x;
  const u32 b = tile >> lo_bits, jr_lo = tile & ((1u << lo_bits) - 1);
  const u64* src = a.in + (u64)b * ((u64)1 << a.log_n);  // scratch is dense [batch][n]
  for (int e = tid; e < E; e += NT) {
    u32 jr = ((u32)(e >> LT) << lo_bits) | jr_lo;
    s[lds_pad(e)] = src[((u64)jr << LT) + (e & (T - 1))];
  }
  __syncthreads();
  dif_all<LT, LT - 1, false, LW, NT>(s, tw, a.tw, tid, a.inverse != 0);
  u64* dst = out_base(a, b);
  const u32 k1_hi = bitrev32(jr_lo, lo_bits) << LW;
  for (int e = tid; e < E; e += NT) {
    int q = e & (W - 1), k2 = e >> LW;
    int row = (int)bitrev32((u32)q, LW);
    u64 v = s[lds_pad((row << LT) + (int)bitrev32((u32)k2, LT))];
    if (a.post) v = gl_mul(v, a.post);
    dst[(u64)(k1_hi | (u32)q) + ((u64)k2 << a.log_n1)] = v;
  }
}

// ---- pass A: strided dimension, T = n1 rows x 2^LW adjacent columns per block ---------------
template <int LT, int LW>
__global__ void __launch_bounds__((NttGeom<LT, LW>::NT)) ntt_cols_kernel(NttArgs a, u64* dst_dense) {
  constexpr int T = 1 << LT, W = 1 << LW, E = T << LW, NT = NttGeom<LT, LW>::NT;
  extern __shared__ __align__(16) u64 smem[];
  u64* s = smem;
  u64* tw = smem + lds_pad(E) + 1;
  const int tid = threadIdx.x;
  for (int i = tid; i < R8Tw<LT>::LDS_WORDS; i += NT) tw[i] = a.tw[R8Tw<LT>::LDS_OFF + i];
  const u32 tiles_per = 1u << (a.log_n2 - LW);
  // XCD-aware order: blocks i and i+8 land on one XCD (round-robin dispatch, speed only), so give
  // each XCD a contiguous run of column tiles -- neighbouring tiles share 128-B lines and L2 sets
  u32 bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const u32 b = bid / tiles_per, c0 = (bid % tiles_per) << LW;
  const u32 coset = b & ((1u << a.logK) - 1);
  const u64* src = in_base(a, b);
  for (int e = tid; e < E; e += NT) {
    int c = e & (W - 1), j = e >> LW;
#if NTT_DBG == 2
    u64 v = (u64)e * 0x9E3779B97F4A7C15ull + c0;
#else
    u64 v = src[((u64)j << a.log_n2) + c0 + c];
#endif
    if (a.pre_full) {
      v = gl_mul(v, a.pre_full[((u64)coset << a.log_n) + ((u64)j << a.log_n2) + c0 + c]);
    } else if (a.pre_lo) {
      v = gl_mul(v, a.pre_lo[((u64)coset << a.log_n2) + c0 + c]);
      v = gl_mul(v, a.pre_hi[((u64)coset << LT) + j]);
    }
    s[lds_pad(e)] = v;
  }
  __syncthreads();
#if NTT_DBG != 1
  dif_all<LT, LT - 1, true, LW, NT>(s, tw, a.tw, tid, a.inverse != 0);
#endif
  // row j holds k1 = bitrev(j); multiply by w_n^(i2*k1) and leave it at row j
  u64* dst = dst_dense ? dst_dense + (u64)b * ((u64)1 << a.log_n) : out_base(a, b);
  for (int e = tid; e < E; e += NT) {
    int c = e & (W - 1), j = e >> LW;
    u64 v = s[lds_pad(e)];
    u64 w;
    if (a.tw4_full) {
#if NTT_DBG == 2
      w = (u64)e * 0x9E3779B97F4A7C55ull + c0;
#else
      w = a.tw4_full[((u64)j << a.log_n2) + c0 + c];
#endif
    } else {
      u32 k1 = bitrev32((u32)j, LT);
      u32 ex = (c0 + c) * k1;  // < n <= 2^24
      w = gl_mul(a.tw4_lo[ex & 4095], a.tw4_hi[ex >> 12]);
    }
#if NTT_DBG == 1
    dst[((u64)j << a.log_n2) + c0 + c] = v ^ w;
#elif NTT_DBG == 2
    v = gl_mul(v, w);
    if (v == 0x123456789ull) dst[((u64)j << a.log_n2) + c0 + c] = v;
#else
    dst[((u64)j << a.log_n2) + c0 + c] = gl_mul(v, w);
#endif
  }
}

// ---- one-barrier tiles ----------------------------------------------------------------------------
// The kernels above put a block barrier around every round and stage the tile through LDS on the way in; measured on the
// 2^22 transform (tools/dbg/ntt_phases.sh, tools/dbg/ntt_pmc.sh) their waves sit parked for 56 % of their cycles. Here
//   * the first radix-8 round takes its inputs straight from global memory (the coalesced load pattern of a tile IS the
//     item pattern of that round: lane `below` needs elements below + m T/8) and its twiddles from the global table;
//   * its outputs go to LDS and the one block barrier of the tile follows;
//   * every later round works on index bits below T/8 <= 2^9, i.e. inside 512 points that one wave owns (rows: 512
//     consecutive points of the tile; columns: T/8 rows x 512/(T/8) adjacent columns), so the wave runs them alone: its LDS
//     operations execute in issue order, and between rounds only the compiler has to be told not to move them;
//   * a row tile is stored by the owning waves (no barrier), a column tile after one more barrier so that the stores keep
//     whole 64-byte row segments.
template <int LT, int LW, bool COLS> struct WaveGeom {
  static constexpr int NBR = LT > 12 ? 2 : 1;  // rounds that cross waves (T = 2^13: the second one too, with a barrier of its own)
  static constexpr int LTS = LT - 3 * NBR;     // index bits left after them
  static constexpr int LWL = 9 - LTS;          // log2 of the sub-transforms (rows) / columns of a wave's 512 points
  static constexpr bool OK = LT >= 3 && LT <= 13 && (!COLS || LT <= 12) && ((1 << LT) << LW) == 8 * NttGeom<LT, LW>::NT && (!COLS || LW >= LWL);
};
// tile shapes that run the shift-twiddle scheme: the three index bits `a` below the first round's are wave-uniform in the first round
// (rows: lane = position, so 2^(L0-3) >= 64; columns: lane = (row bits, column), so 2^(LW + L0 - 3) >= 64) and the first wave-local
// round is a full radix-8 one whose block (the first round's output index) is wave-uniform
template <int LT, int LW, bool COLS> struct ShiftGeom {
  using G = WaveGeom<LT, LW, COLS>;
  static constexpr int L0 = LT - 3;
  static constexpr bool OK = G::OK && G::NBR == 1 && G::LTS >= 6 && (COLS ? (LW + L0 - 3 >= 6) : (LW == 0 && L0 - 3 >= 6));
  static constexpr int TC_WORDS = 64 << (L0 - 3);
};
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// one round of the sub-transforms of length 2^LTS inside the 512 points at `base`; LWS = log2 of the tile's row stride (COLS)
template <int LT, int LTS, int HI, int R, bool COLS, int LWL, int LWS, bool MERGED = false>
__device__ __forceinline__ void wave_round(u64* s, int base, const u64* tw, int lane, bool inverse, const u64* __restrict__ tcw = nullptr) {
  constexpr int LO = HI - R + 1, WL = 1 << LWL;
  constexpr int ITEMS = 512 >> R;
#pragma unroll
  for (int item0 = 0; item0 < ITEMS; item0 += 64) {
    int item = item0 + lane;
    int c, rest;
    if (COLS) { c = item & (WL - 1); rest = item >> LWL; }
    else { rest = item & ((1 << (LTS - R)) - 1); c = item >> (LTS - R); }
    int below = rest & ((1 << LO) - 1), above = rest >> LO;
    int j0 = (above << (HI + 1)) | below;
    u64 x[1 << R];
#pragma unroll
    for (int m = 0; m < (1 << R); m++) {
      int j = j0 + (m << LO);
      x[m] = s[lds_pad(base + (COLS ? (j << LWS) + c : (c << LTS) + j))];
    }
    butterfly<LT, HI, R, false, MERGED>(x, below, tw, tcw, inverse);
#pragma unroll
    for (int m = 0; m < (1 << R); m++) {
      int j = j0 + (m << LO);
      s[lds_pad(base + (COLS ? (j << LWS) + c : (c << LTS) + j))] = x[m];
    }
  }
}
// tcw != nullptr only in the instantiations with SHIFT: the first of these rounds then takes the merged table
template <int LT, int LTS, int HI, bool COLS, int LWL, int LWS, bool SHIFT = false>
__device__ __forceinline__ void wave_rounds(u64* s, int base, const u64* tw, int lane, bool inverse, const u64* __restrict__ tcw = nullptr) {
  if constexpr (HI >= 0) {
    constexpr int R = (HI + 1 >= 3) ? 3 : HI + 1;
    wave_sync();
    wave_round<LT, LTS, HI, R, COLS, LWL, LWS, SHIFT>(s, base, tw, lane, inverse, tcw);
    wave_rounds<LT, LTS, HI - R, COLS, LWL, LWS, false>(s, base, tw, lane, inverse);
  }
}

template <int LT, int LW, bool SHIFT = false>
__global__ void __launch_bounds__((NttGeom<LT, LW>::NT)) NTT_WAVES_ATTR ntt_rows_v2_kernel(NttArgs a) {
  using G = WaveGeom<LT, LW, false>;
  static_assert(!SHIFT || ShiftGeom<LT, LW, false>::OK, "tile shape without wave-uniform shift twiddles");
  constexpr int T = 1 << LT, E = T << LW, NT = NttGeom<LT, LW>::NT, LTS = G::LTS, L0 = LT - 3;
  extern __shared__ __align__(16) u64 smem[];
  u64* s = smem;
  u64* tw = smem + lds_pad(E) + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < R8Tw<LT>::LDS_WORDS; i += NT) tw[i] = a.tw[R8Tw<LT>::LDS_OFF + i];
  const u32 n1 = 1u << a.log_n1;
  const u64 total_rows = (u64)a.batch << a.log_n1;
  const u64 row0 = (u64)blockIdx.x << LW;
  const bool two_pass = a.log_n1 != 0, inverse = a.inverse != 0;
  {
    const int c = tid >> L0, below = tid & ((1 << L0) - 1);
    const u64 g = row0 + c;
    u64 x[8];
    if (g < total_rows) {
      u32 b = (u32)(g >> a.log_n1), jr = (u32)(g & (n1 - 1));
      if (two_pass) {
        const u64* src = out_base(a, b) + ((u64)jr << LT);  // pass A left row jr in place
#pragma unroll
#if NTT_DBG == 2
        for (int m = 0; m < 8; m++) x[m] = (u64)(tid + m) * 0x9E3779B97F4A7C15ull + (u64)(size_t)src;
#else
        for (int m = 0; m < 8; m++) x[m] = src[below + (m << L0)];
#endif
      } else {
        const u64* src = in_base(a, b);
#pragma unroll
        for (int m = 0; m < 8; m++) x[m] = src[below + (m << L0)];
        if (a.pre_lo) {
          const u64* pre = a.pre_lo + ((u64)(b & ((1u << a.logK) - 1)) << LT);
#pragma unroll
          for (int m = 0; m < 8; m++) x[m] = gl_mul(x[m], pre[below + (m << L0)]);
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < 8; m++) x[m] = 0;
    }
#if NTT_DBG != 1
    if constexpr (SHIFT) {
      butterfly<LT, 2, 3>(x, 0, nullptr, nullptr, inverse);  // the radix-8 butterfly alone (HI = 2: no table)
      shift_twiddles(x, __builtin_amdgcn_readfirstlane(below >> (L0 - 3)), inverse);
    } else {
      butterfly<LT, LT - 1, 3, true>(x, below, tw, a.tw, inverse);
    }
#endif
#pragma unroll
    for (int m = 0; m < 8; m++) s[lds_pad((c << LT) + below + (m << L0))] = x[m];
  }
  __syncthreads();
#if NTT_DBG != 1
  if constexpr (G::NBR == 2) {
    dif_round<LT, LT - 4, 3, false, LW, NT>(s, tw, a.tw, tid, inverse);
    __syncthreads();
  }
  if constexpr (SHIFT) {
    // wave m owns the first round's output block m, whose frequency is r = bitrev3(m): rows r + 8 r' of the merged table
    const int r = __builtin_amdgcn_readfirstlane((int)bitrev32((u32)(wave & 7), 3));
    wave_rounds<LT, LTS, LTS - 1, false, G::LWL, 0, true>(s, wave << 9, tw, lane, inverse, a.tc + (r << (L0 - 3)));
  } else {
    wave_rounds<LT, LTS, LTS - 1, false, G::LWL, 0>(s, wave << 9, tw, lane, inverse);
  }
#endif
  if (a.bitrev_out) {
    wave_sync();
    u64 v[8];  // all eight LDS reads in flight before the first store (one read - wait - store per point otherwise)
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = s[lds_pad((wave << 9) + lane + (k << 6))];
    if (a.post) {
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = gl_mul(v[k], a.post);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int e = (wave << 9) + lane + (k << 6);
      int r = e >> LT, p = e & (T - 1);
      u64 g = row0 + r;
      if (g >= total_rows) continue;
      u32 b = (u32)(g >> a.log_n1), jr = (u32)(g & (n1 - 1));
#if NTT_DBG == 2
      if (v[k] == 0x123456789ull)
#endif
      out_base(a, b)[((u64)jr << LT) + p] = v[k];
    }
  } else {
    __syncthreads();
    for (int e = tid; e < E; e += NT) {
      int r = e >> LT, p = e & (T - 1);
      u64 g = row0 + r;
      if (g >= total_rows) continue;
      u32 b = (u32)(g >> a.log_n1), jr = (u32)(g & (n1 - 1));
      u64 v = s[lds_pad((r << LT) + (int)bitrev32((u32)p, LT))];
      if (a.post) v = gl_mul(v, a.post);
      out_base(a, b)[((u64)jr << LT) + p] = v;
    }
  }
}

template <int LT, int LW, bool SHIFT = false>
__global__ void __launch_bounds__((NttGeom<LT, LW>::NT)) NTT_WAVES_ATTR ntt_cols_v2_kernel(NttArgs a, u64* dst_dense) {
  using G = WaveGeom<LT, LW, true>;
  static_assert(!SHIFT || ShiftGeom<LT, LW, true>::OK, "tile shape without wave-uniform shift twiddles");
  constexpr int W = 1 << LW, E = (1 << LT) << LW, NT = NttGeom<LT, LW>::NT, LTS = G::LTS, LWL = G::LWL;
  extern __shared__ __align__(16) u64 smem[];
  u64* s = smem;
  u64* tw = smem + lds_pad(E) + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < R8Tw<LT>::LDS_WORDS; i += NT) tw[i] = a.tw[R8Tw<LT>::LDS_OFF + i];
  const u32 tiles_per = 1u << (a.log_n2 - LW);
  u32 bid = blockIdx.x;  // XCD-aware order, as in ntt_cols_kernel
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const u32 b = bid / tiles_per, c0 = (bid % tiles_per) << LW;
  const u32 coset = b & ((1u << a.logK) - 1);
  const u64* src = in_base(a, b);
  const bool inverse = a.inverse != 0;
  {
    const int c = tid & (W - 1), below = tid >> LW;
    u64 x[8];
#pragma unroll
#if NTT_DBG == 2
    for (int m = 0; m < 8; m++) x[m] = (u64)(tid + m) * 0x9E3779B97F4A7C15ull + (u64)(size_t)src + c0;
#else
    for (int m = 0; m < 8; m++) x[m] = src[((u64)(below + (m << LTS)) << a.log_n2) + c0 + c];
#endif
    if (a.pre_full) {
      const u64* pre = a.pre_full + ((u64)coset << a.log_n) + c0 + c;
#pragma unroll
      for (int m = 0; m < 8; m++) x[m] = gl_mul(x[m], pre[(u64)(below + (m << LTS)) << a.log_n2]);
    } else if (a.pre_lo) {
      u64 pl = a.pre_lo[((u64)coset << a.log_n2) + c0 + c];
#pragma unroll
      for (int m = 0; m < 8; m++) x[m] = gl_mul(gl_mul(x[m], pl), a.pre_hi[((u64)coset << LT) + below + (m << LTS)]);
    }
#if NTT_DBG != 1
    if constexpr (SHIFT) {
      butterfly<LT, 2, 3>(x, 0, nullptr, nullptr, inverse);
      shift_twiddles(x, __builtin_amdgcn_readfirstlane(below >> (LTS - 3)), inverse);
    } else {
      butterfly<LT, LT - 1, 3, true>(x, below, tw, a.tw, inverse);
    }
#endif
#pragma unroll
    for (int m = 0; m < 8; m++) s[lds_pad(((below + (m << LTS)) << LW) + c)] = x[m];
  }
  __syncthreads();
#if NTT_DBG != 1
  {
    constexpr int GW = W >> LWL;  // column groups of a tile; wave = (top three row bits, column group)
    const int cg = wave & (GW - 1), jh = wave / GW;
    if constexpr (SHIFT) {
      const int r = __builtin_amdgcn_readfirstlane((int)bitrev32((u32)jh, 3));
      wave_rounds<LT, LTS, LTS - 1, true, LWL, LW, true>(s, ((jh << LTS) << LW) + (cg << LWL), tw, lane, inverse, a.tc + (r << (LTS - 3)));
    } else {
      wave_rounds<LT, LTS, LTS - 1, true, LWL, LW>(s, ((jh << LTS) << LW) + (cg << LWL), tw, lane, inverse);
    }
  }
#endif
  __syncthreads();
  // row j holds k1 = bitrev(j); multiply by w_n^(i2*k1) and leave it at row j
  u64* dst = dst_dense ? dst_dense + (u64)b * ((u64)1 << a.log_n) : out_base(a, b);
  // (one load - wait - multiply - store chain per point: requesting the 4-step twiddles of a lane eight or four at a time, here or
  // before the barrier, costs registers this kernel does not have at 8 waves per SIMD -- 2-14 spills, 31.9 -> 33.5-34.7 us)
#pragma unroll
  for (int k = 0; k < 8; k++) {
    int e = tid + k * NT;
    int c = e & (W - 1), j = e >> LW;
    u64 v = s[lds_pad(e)];
    u64 w;
    if (a.tw4_full) {
#if NTT_DBG == 2
      w = (u64)e * 0x9E3779B97F4A7C55ull + c0;
#else
      w = a.tw4_full[((u64)j << a.log_n2) + c0 + c];
#endif
    } else {
      u32 k1 = bitrev32((u32)j, LT);
      u32 ex = (c0 + c) * k1;  // < n <= 2^24
      w = gl_mul(a.tw4_lo[ex & 4095], a.tw4_hi[ex >> 12]);
    }
#if NTT_DBG == 1
    dst[((u64)j << a.log_n2) + c0 + c] = v ^ w;
#elif NTT_DBG == 2
    v = gl_mul(v, w);
    if (v == 0x123456789ull) dst[((u64)j << a.log_n2) + c0 + c] = v;
#else
    dst[((u64)j << a.log_n2) + c0 + c] = gl_mul(v, w);
#endif
  }
}

// table of one radix-8 round: out[(r-1) << lo | below] = w^((below << shift) * r), r = 1..7, below < 2^lo
__global__ void r8_round_twiddles_kernel(u64* out, u64 w, u32 lo, u32 shift) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (7u << lo)) return;
  u32 r = (i >> lo) + 1, below = i & ((1u << lo) - 1);
  out[i] = gl_pow(w, ((u64)below << shift) * r);
}
// merged table of the shift-twiddle scheme: out[q << lb | b] = w^(b q), q < 64, b < 2^lb
__global__ void merged_twiddles_kernel(u64* out, u64 w, u32 lb) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (64u << lb)) return;
  out[i] = gl_pow(w, (u64)(i & ((1u << lb) - 1)) * (i >> lb));
}
__global__ void powers_kernel(u64* out, u64 base, u64 first, u64 stride_exp, u32 count) {
  // out[i] = first * base^(i * stride_exp)
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  out[i] = gl_mul(first, gl_pow(base, (u64)i * stride_exp));
}
// pre_lo[j][i2] = s_j^i2, pre_hi[j][i1] = s_j^(i1*n2), s_j = shift * w_{n*K}^j
__global__ void coset_tables_kernel(u64* lo, u64* hi, u64 shift, u64 w_nk, u32 log_n1, u32 log_n2, u32 K) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  u32 n1 = 1u << log_n1, n2 = 1u << log_n2;
  u32 per = n1 + n2;
  if (i >= per * K) return;
  u32 j = i / per, r = i % per;
  u64 sj = gl_mul(shift, gl_pow(w_nk, j));
  if (r < n2) lo[(u64)j * n2 + r] = gl_pow(sj, r);
  else hi[(u64)j * n1 + (r - n2)] = gl_pow(sj, (u64)(r - n2) << log_n2);
}
// tw4_full[j][i2] = w_n^(i2 * bitrev(j))
__global__ void tw4_full_kernel(u64* out, u64 wn, u32 log_n1, u32 log_n2) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ((u64)1 << (log_n1 + log_n2))) return;
  u32 j = (u32)(i >> log_n2), i2 = (u32)(i & ((1u << log_n2) - 1));
  out[i] = gl_pow(wn, (u64)i2 * bitrev32(j, log_n1));
}
// pre_full[c][i] = (shift * w_nk^c)^i
__global__ void pre_full_kernel(u64* out, u64 shift, u64 w_nk, u32 log_n, u32 K) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ((u64)K << log_n)) return;
  u32 c = (u32)(i >> log_n);
  out[i] = gl_pow(gl_mul(shift, gl_pow(w_nk, c)), i & (((u64)1 << log_n) - 1));
}
__global__ void scale_powers_kernel(u64* data, u32 log_n, u32 batch, u64 base, u64 first) {
  // data[b][i] *= first * base^i
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ((u64)batch << log_n)) return;
  u64 k = i & (((u64)1 << log_n) - 1);
  data[i] = gl_mul(data[i], gl_mul(first, gl_pow(base, k)));
}

// ---- host side ------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

static hipError_t dev_alloc(u64** p, size_t words) { return hipMalloc((void**)p, words * sizeof(u64)); }

NttPlan::~NttPlan() {
  (void)hipFree(tw_a); (void)hipFree(tw_b); (void)hipFree(tw4_lo); (void)hipFree(tw4_hi); (void)hipFree(tw4_full);
  (void)hipFree(tc_a); (void)hipFree(tc_b);
}
CosetTables::~CosetTables() { (void)hipFree(lo); (void)hipFree(hi); (void)hipFree(full); }

// Cooley-Tukey split n = n1 * n2 of the sizes that do not fit one block (n1 = the strided pass)
static u32 split_log_n1(u32 log_n) {
  if (log_n <= 13) return 0;  // 2^13 points = one 64 KB tile of 1024 lanes: a single pass over HBM
  u32 l1 = (log_n + 1) / 2;
  if (log_n >= 22) l1 = log_n - 12;  // measured (tools/dbg/ntt22.py): 2^10 x 2^12 runs 5-9 % faster than 2^11 x 2^11 at 2^22
  if (const char* e = getenv("MP2G_NTT_N1")) {  // tuning aid: the strided dimension's size
    int v = atoi(e);
    if (v >= 7 && v <= 12 && (int)log_n - v >= 1 && (int)log_n - v <= 12) l1 = (u32)v;
  }
  return l1;
}
// points per block (measured on MI355X, tools/dbg/ntt_only.py): 4096 (256 lanes) for T <= 2^10 and
// T = 2^12, 8192 (512 lanes, two blocks per CU) for T = 2^11 where the twiddle table is amortised
template <int LT> static constexpr int rows_lw() { return LT >= 12 ? 0 : (LT == 11 ? 2 : 12 - LT); }  // 2^13: one row of 8192 points
template <int LT> static constexpr int cols_lw() { return LT >= 11 ? 2 : (LT == 10 ? 3 : 12 - LT); }  // 2^10 x 8 columns: +4 % at 2^22 (ntt22.py)
// pass sizes whose default tile shape runs the shift-twiddle scheme (ShiftGeom); MP2G_NTT_NOSHIFT=1 keeps the classic tables (A/B)
static bool shift_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MP2G_NTT_NOSHIFT"); v = (e && atoi(e)) ? 0 : 1; }
  return v != 0;
}
template <int LT> static constexpr bool shift_rows_lt() { return ShiftGeom<LT, rows_lw<LT>(), false>::OK; }
template <int LT> static constexpr bool shift_cols_lt() { return ShiftGeom<LT, cols_lw<LT>(), true>::OK; }
static bool shift_rows_ok(u32 lt) {
  if (!shift_enabled()) return false;
  switch (lt) { case 12: return shift_rows_lt<12>(); case 13: return shift_rows_lt<13>(); default: return false; }
}
static bool shift_cols_ok(u32 lt) {
  if (!shift_enabled()) return false;
  switch (lt) { case 9: return shift_cols_lt<9>(); case 10: return shift_cols_lt<10>(); case 11: return shift_cols_lt<11>(); case 12: return shift_cols_lt<12>(); default: return false; }
}
hipError_t NttEngine::plan(u32 log_n, bool inverse, NttPlan** out) {
  u32 key = log_n * 2 + (inverse ? 1 : 0);
  auto it = plans.find(key);
  if (it != plans.end()) { *out = it->second.get(); return hipSuccess; }
  std::unique_ptr<NttPlan> p(new NttPlan());
  p->log_n = log_n;
  p->log_n1 = split_log_n1(log_n);
  p->log_n2 = log_n - p->log_n1;
  u64 wn = gl_root_of_unity(log_n);
  if (inverse) wn = gl_inv(wn);
  auto powers = [&](u64** dst, u64 base, u32 count) -> hipError_t {
    HIPCHK(dev_alloc(dst, count ? count : 1));
    if (count) hipLaunchKernelGGL(powers_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, *dst, base, (u64)1, (u64)1, count);
    return hipGetLastError();
  };
  // inner DFT roots: w_{n2}^k (pass B) and w_{n1}^k (pass A)
  u64 w2 = gl_pow(wn, (u64)1 << p->log_n1), w1 = gl_pow(wn, (u64)1 << p->log_n2);
  // per-round twiddle tables of the radix-8 butterflies (R8Tw): round rho has LO = lt-3-3 rho, E = below << 3 rho
  auto stage_tw = [&](u64** dst, u64 base, u32 lt) -> hipError_t {
    HIPCHK(dev_alloc(dst, (size_t)1 << lt));
    u32 off = 0;
    for (u32 rho = 0; (int)lt - 3 - 3 * (int)rho >= 1; rho++) {
      u32 lo = lt - 3 - 3 * rho, count = 7u << lo;
      hipLaunchKernelGGL(r8_round_twiddles_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, *dst + off, base, lo, 3 * rho);
      HIPCHK(hipGetLastError());
      off += count;
    }
    return hipSuccess;
  };
  // merged tables of the shift-twiddle scheme for the pass sizes that run it (shift_rows_ok / shift_cols_ok)
  auto merged_tw = [&](u64** dst, u64 base, u32 lt) -> hipError_t {
    const u32 lb = lt - 6, count = 64u << lb;
    HIPCHK(dev_alloc(dst, count));
    hipLaunchKernelGGL(merged_twiddles_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, *dst, base, lb);
    return hipGetLastError();
  };
  HIPCHK(stage_tw(&p->tw_b, w2, p->log_n2));
  if (shift_rows_ok(p->log_n2)) HIPCHK(merged_tw(&p->tc_b, w2, p->log_n2));
  if (p->log_n1) {
    HIPCHK(stage_tw(&p->tw_a, w1, p->log_n1));
    if (shift_cols_ok(p->log_n1)) HIPCHK(merged_tw(&p->tc_a, w1, p->log_n1));
    HIPCHK(powers(&p->tw4_lo, wn, 4096));
    HIPCHK(powers(&p->tw4_hi, gl_pow(wn, 4096), log_n > 12 ? (1u << (log_n - 12)) : 1));
    if (log_n <= 22 && !getenv("MP2G_NTT_NOFULL")) {  // full 4-step table (<= 32 MB): one multiply per point instead of two
      HIPCHK(dev_alloc(&p->tw4_full, (size_t)1 << log_n));
      hipLaunchKernelGGL(tw4_full_kernel, dim3((u32)((((u64)1 << log_n) + 255) / 256)), dim3(256), 0, stream, p->tw4_full, wn, p->log_n1, p->log_n2);
      HIPCHK(hipGetLastError());
    }
  }
  p->n_inv = inverse ? gl_inv(((u64)1 << log_n) % GL_P) : 0;
  *out = p.get();
  plans[key] = std::move(p);
  return hipSuccess;
}

hipError_t NttEngine::coset(u32 log_n, u32 logK, u64 shift, CosetTables** out) {
  u64 key = ((u64)log_n << 8 | logK) ^ (shift * 0x9E3779B97F4A7C15ULL);
  auto it = cosets.find(key);
  if (it != cosets.end() && it->second->shift == shift && it->second->log_n == log_n && it->second->logK == logK) {
    *out = it->second.get(); return hipSuccess;
  }
  std::unique_ptr<CosetTables> c(new CosetTables());
  c->log_n = log_n; c->logK = logK; c->shift = shift;
  u32 log_n1 = split_log_n1(log_n), log_n2 = log_n - log_n1;
  u32 K = 1u << logK, n1 = 1u << log_n1, n2 = 1u << log_n2;
  HIPCHK(dev_alloc(&c->lo, (size_t)K * n2));
  HIPCHK(dev_alloc(&c->hi, (size_t)K * n1));
  u32 total = (n1 + n2) * K;
  hipLaunchKernelGGL(coset_tables_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, c->lo, c->hi, shift,
                     gl_root_of_unity(log_n + logK), log_n1, log_n2, K);
  HIPCHK(hipGetLastError());
  if (log_n1 && ((u64)K << log_n) <= ((u64)1 << 23)) {  // <= 64 MB: single-table pre-scale for two-pass sizes
    HIPCHK(dev_alloc(&c->full, (size_t)K << log_n));
    u64 tot = (u64)K << log_n;
    hipLaunchKernelGGL(pre_full_kernel, dim3((u32)((tot + 255) / 256)), dim3(256), 0, stream, c->full, shift,
                       gl_root_of_unity(log_n + logK), log_n, K);
    HIPCHK(hipGetLastError());
  }
  *out = c.get();
  if (cosets.count(key)) generation++;  // a colliding key replaces (frees) tables a captured graph may reference
  cosets[key] = std::move(c);
  return hipSuccess;
}

hipError_t NttEngine::ensure_scratch(size_t words) {
  if (words <= scratch_words) return hipSuccess;
  if (scratch) HIPCHK(hipFree(scratch));
  scratch = nullptr; scratch_words = 0;
  generation++;  // captured graphs that reference the old scratch are stale
  HIPCHK(dev_alloc(&scratch, words));
  scratch_words = words;
  return hipSuccess;
}

template <int LT, int LW> static size_t lds_bytes() {
  int e = (1 << LT) << LW;
  return (size_t)(e + (e >> 4) + 1 + R8Tw<LT>::LDS_WORDS + 1) * sizeof(u64);
}

// hipFuncSetAttribute acts on the current device's copy of the kernel: one flag per device, so a process that
// drives several GPUs raises the LDS limit on each of them
#define MP2G_MAX_DEVICES 64
static bool* attr_flag(bool* flags) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return &flags[dev >= 0 && dev < MP2G_MAX_DEVICES ? dev : 0];
}
// tuning aid: MP2G_NTT_LW11=1|2|3 overrides the tile width (log2 of rows / columns per block) of the T = 2^11 passes
static int lw11_override() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MP2G_NTT_LW11"); v = e ? atoi(e) : 0; }
  return v;
}
// tuning / A-B aid: MP2G_NTT_V1=1 runs the barrier-per-round kernels everywhere
static bool use_v1() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MP2G_NTT_V1"); v = e ? atoi(e) : 0; }
  return v != 0;
}
template <int LT, int LW>
static hipError_t launch_rows_lw(const NttArgs& a, bool nat_two_pass, hipStream_t st) {
  constexpr int NT = NttGeom<LT, LW>::NT;
  size_t lds = lds_bytes<LT, LW>();
  u64 total_rows = (u64)a.batch << a.log_n1;
  if constexpr (ShiftGeom<LT, LW, false>::OK) {
    if (!nat_two_pass && !use_v1() && a.tc) {
      static bool flags[MP2G_MAX_DEVICES];
      bool* attr = attr_flag(flags);
      if (!*attr) { HIPCHK(hipFuncSetAttribute((const void*)ntt_rows_v2_kernel<LT, LW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); *attr = true; }
      hipLaunchKernelGGL((ntt_rows_v2_kernel<LT, LW, true>), dim3((u32)((total_rows + (1u << LW) - 1) >> LW)), dim3(NT), lds, st, a);
      return hipGetLastError();
    }
  }
  if constexpr (WaveGeom<LT, LW, false>::OK) {
    if (!nat_two_pass && !use_v1()) {
      static bool flags[MP2G_MAX_DEVICES];
      bool* attr = attr_flag(flags);
      if (!*attr) { HIPCHK(hipFuncSetAttribute((const void*)ntt_rows_v2_kernel<LT, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); *attr = true; }
      hipLaunchKernelGGL((ntt_rows_v2_kernel<LT, LW>), dim3((u32)((total_rows + (1u << LW) - 1) >> LW)), dim3(NT), lds, st, a);
      return hipGetLastError();
    }
  }
  if (nat_two_pass) {
    static bool flags[MP2G_MAX_DEVICES];
    bool* attr = attr_flag(flags);
    if (!*attr) { HIPCHK(hipFuncSetAttribute((const void*)ntt_rows_nat_kernel<LT, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); *attr = true; }
    hipLaunchKernelGGL((ntt_rows_nat_kernel<LT, LW>), dim3((u32)(total_rows >> LW)), dim3(NT), lds, st, a);
  } else {
    static bool flags[MP2G_MAX_DEVICES];
    bool* attr = attr_flag(flags);
    if (!*attr) { HIPCHK(hipFuncSetAttribute((const void*)ntt_rows_kernel<LT, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); *attr = true; }
    hipLaunchKernelGGL((ntt_rows_kernel<LT, LW>), dim3((u32)((total_rows + (1u << LW) - 1) >> LW)), dim3(NT), lds, st, a);
  }
  return hipGetLastError();
}
template <int LT>
static hipError_t launch_rows(const NttArgs& a, bool nat_two_pass, hipStream_t st) {
  if constexpr (LT == 11) {
    if (lw11_override() == 1) return launch_rows_lw<LT, 1>(a, nat_two_pass, st);
    if (lw11_override() == 3) return launch_rows_lw<LT, 3>(a, nat_two_pass, st);
  }
  return launch_rows_lw<LT, rows_lw<LT>()>(a, nat_two_pass, st);
}
template <int LT, int LW>
static hipError_t launch_cols_lw(const NttArgs& a, u64* dst_dense, hipStream_t st) {
  constexpr int NT = NttGeom<LT, LW>::NT;
  size_t lds = lds_bytes<LT, LW>();
  if constexpr (ShiftGeom<LT, LW, true>::OK) {
    if (!use_v1() && a.tc) {
      static bool flags[MP2G_MAX_DEVICES];
      bool* attr = attr_flag(flags);
      if (!*attr) { HIPCHK(hipFuncSetAttribute((const void*)ntt_cols_v2_kernel<LT, LW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); *attr = true; }
      hipLaunchKernelGGL((ntt_cols_v2_kernel<LT, LW, true>), dim3(a.batch << (a.log_n2 - LW)), dim3(NT), lds, st, a, dst_dense);
      return hipGetLastError();
    }
  }
  if constexpr (WaveGeom<LT, LW, true>::OK) {
    if (!use_v1()) {
      static bool flags[MP2G_MAX_DEVICES];
      bool* attr = attr_flag(flags);
      if (!*attr) { HIPCHK(hipFuncSetAttribute((const void*)ntt_cols_v2_kernel<LT, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); *attr = true; }
      hipLaunchKernelGGL((ntt_cols_v2_kernel<LT, LW>), dim3(a.batch << (a.log_n2 - LW)), dim3(NT), lds, st, a, dst_dense);
      return hipGetLastError();
    }
  }
  static bool flags[MP2G_MAX_DEVICES];
  bool* attr = attr_flag(flags);
  if (!*attr) { HIPCHK(hipFuncSetAttribute((const void*)ntt_cols_kernel<LT, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); *attr = true; }
  hipLaunchKernelGGL((ntt_cols_kernel<LT, LW>), dim3(a.batch << (a.log_n2 - LW)), dim3(NT), lds, st, a, dst_dense);
  return hipGetLastError();
}

template <int LT>
static hipError_t launch_cols(const NttArgs& a, u64* dst_dense, hipStream_t st) {
  if constexpr (LT == 10) {  // tuning aid: MP2G_NTT_LW10=1|2
    static int v = -1;
    if (v < 0) { const char* e = getenv("MP2G_NTT_LW10"); v = e ? atoi(e) : 0; }
    if (v == 1) return launch_cols_lw<LT, 1>(a, dst_dense, st);
    if (v == 2) return launch_cols_lw<LT, 2>(a, dst_dense, st);
  }
  if constexpr (LT == 11) {
    if (lw11_override() == 1) return launch_cols_lw<LT, 1>(a, dst_dense, st);
    if (lw11_override() == 3) return launch_cols_lw<LT, 3>(a, dst_dense, st);
  }
  return launch_cols_lw<LT, cols_lw<LT>()>(a, dst_dense, st);
}
#define ROWS_CASE(N) case N: return launch_rows<N>(a, nat, st);
static hipError_t dispatch_rows(u32 lt, const NttArgs& a, bool nat, hipStream_t st) {
  switch (lt) {
    ROWS_CASE(1) ROWS_CASE(2) ROWS_CASE(3) ROWS_CASE(4) ROWS_CASE(5) ROWS_CASE(6)
    ROWS_CASE(7) ROWS_CASE(8) ROWS_CASE(9) ROWS_CASE(10) ROWS_CASE(11) ROWS_CASE(12) ROWS_CASE(13)
    default: return hipErrorInvalidValue;
  }
}
#define COLS_CASE(N) case N: return launch_cols<N>(a, dense, st);
static hipError_t dispatch_cols(u32 lt, const NttArgs& a, u64* dense, hipStream_t st) {
  switch (lt) {
    COLS_CASE(7) COLS_CASE(8) COLS_CASE(9) COLS_CASE(10) COLS_CASE(11) COLS_CASE(12)
    default: return hipErrorInvalidValue;
  }
}

#ifdef MP2G_EXPERIMENT_NTT_PRIORITY
hipError_t NttEngine::run(const u64* in, u64* out, u32 log_n, u32 polys, u32 logK, u64 in_poly_stride,
                          u64 out_poly_stride, bool inverse, const CosetTables* pre, bool bitrev_out) {
  if (!hi_stream) return run_impl(in, out, log_n, polys, logK, in_poly_stride, out_poly_stride, inverse, pre, bitrev_out);
  hipStream_t base = stream;
  HIPCHK(hipEventRecord(ev_fork, base));
  HIPCHK(hipStreamWaitEvent(hi_stream, ev_fork, 0));
  stream = hi_stream;
  hipError_t e = run_impl(in, out, log_n, polys, logK, in_poly_stride, out_poly_stride, inverse, pre, bitrev_out);
  stream = base;
  if (e != hipSuccess) return e;
  HIPCHK(hipEventRecord(ev_join, hi_stream));
  return hipStreamWaitEvent(base, ev_join, 0);
}
hipError_t NttEngine::run_impl(const u64* in, u64* out, u32 log_n, u32 polys, u32 logK, u64 in_poly_stride,
                               u64 out_poly_stride, bool inverse, const CosetTables* pre, bool bitrev_out) {
#else
hipError_t NttEngine::run(const u64* in, u64* out, u32 log_n, u32 polys, u32 logK, u64 in_poly_stride,
                          u64 out_poly_stride, bool inverse, const CosetTables* pre, bool bitrev_out) {
#endif
  if (log_n == 0 || log_n > 24) return hipErrorInvalidValue;
  NttPlan* p;
  HIPCHK(plan(log_n, inverse, &p));
  NttArgs a{};
  a.in = in; a.out = out;
  a.in_poly_stride = in_poly_stride; a.out_poly_stride = out_poly_stride;
  a.logK = logK; a.log_n = log_n; a.log_n1 = p->log_n1; a.log_n2 = p->log_n2;
  a.batch = polys << logK;
  a.tw4_lo = p->tw4_lo; a.tw4_hi = p->tw4_hi; a.tw4_full = p->tw4_full;
  a.pre_lo = pre ? pre->lo : nullptr; a.pre_hi = pre ? pre->hi : nullptr; a.pre_full = pre ? pre->full : nullptr;
  a.post = p->n_inv;
  a.bitrev_out = bitrev_out ? 1 : 0;
  a.inverse = inverse ? 1 : 0;
  if (p->log_n1 == 0) {
    a.tw = p->tw_b; a.tc = p->tc_b;
    return dispatch_rows(p->log_n2, a, false, stream);
  }
  u64* dense = nullptr;
  if (!bitrev_out) {
    HIPCHK(ensure_scratch((size_t)a.batch << log_n));
    dense = scratch;
  }
  a.tw = p->tw_a; a.tc = p->tc_a;
  HIPCHK(dispatch_cols(p->log_n1, a, dense, stream));
  a.tw = p->tw_b; a.tc = p->tc_b;
  a.pre_lo = a.pre_hi = a.pre_full = nullptr;
  if (!bitrev_out) a.in = dense;
  return dispatch_rows(p->log_n2, a, !bitrev_out, stream);
}

hipError_t NttEngine::scale_powers(u64* data, u32 log_n, u32 batch, u64 base, u64 first) {
  u64 total = (u64)batch << log_n;
  hipLaunchKernelGGL(scale_powers_kernel, dim3((u32)((total + 255) / 256)), dim3(256), 0, stream, data, log_n, batch, base, first);
  return hipGetLastError();
}

NttEngine::~NttEngine() { if (scratch) (void)hipFree(scratch); }

}  // namespace mp2g
