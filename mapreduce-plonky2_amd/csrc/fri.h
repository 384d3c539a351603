// Shapes shared by the FRI kernels (fri.hip) and the batched prover (prover.hip).
#pragma once
#include "gl.cuh"

namespace mp2g {

// plonky2 iop/challenger.rs Challenger, one per proof in flight, resident in HBM
struct ChState {
  u64 state[12];
  u64 in[8];
  u64 out[8];
  u32 n_in, n_out;
};
// one committed oracle (PolynomialBatch) for `B` proofs; *_bstride = words between proofs
// (0 for the preprocessed constants/sigmas oracle shared by every proof of the circuit)
struct OracleRef {
  const u64* coeffs;  // [w][n]
  const u64* values;  // [w][N] polynomial-major, bit-reversed index
  const u64* levels;  // Merkle levels
  u64 coeff_bstride, value_bstride, level_bstride;
  u32 w;
};
struct FriShape {
  u32 log_n, rate_bits, cap_h, n_oracles;
  u32 n_polys;  // sum of w
  u32 zs_oracle, zs_count;
  u32 lookup_count;  // the last lookup_count polynomials of oracle zs_oracle: opened at zeta AND g*zeta, last in both batches
  OracleRef o[8];
};
// FRI batch order (plonk/circuit_data.rs fri_all_polys / fri_next_batch_polys; also the order of the flat openings
// and of the transcript, OpeningSet::to_fri_openings): batch 0 (zeta) = every polynomial in oracle order except the
// lookup polynomials, which come last; batch 1 (g zeta) = the Z polynomials, then the lookup polynomials.
GLHD u32 fri_batch_len(const FriShape& sh, u32 batch) { return batch ? sh.zs_count + sh.lookup_count : sh.n_polys; }
GLHD void fri_batch_poly(const FriShape& sh, u32 batch, u32 j, u32& o, u32& p) {
  const u32 zo = sh.zs_oracle, wz = sh.o[zo].w - sh.lookup_count;
  if (batch) { o = zo; p = j < sh.zs_count ? j : wz + (j - sh.zs_count); return; }
  for (u32 oi = 0; oi < sh.n_oracles; oi++) {
    const u32 w = oi == zo ? wz : sh.o[oi].w;
    if (j < w) { o = oi; p = j; return; }
    j -= w;
  }
  o = zo; p = wz + j;
}
struct FriLayers {
  u32 n_layers;
  u32 arity_bits[8];
  const u64* values[8];  // [B][2][m_i] bit-reversed evaluations of layer i
  const u64* levels[8];
  u64 value_bstride[8], level_bstride[8];
};

hipError_t challenger_init(hipStream_t s, ChState* st, u32 B);
hipError_t challenger_step(hipStream_t s, int variant, ChState* st, u32 B, const u64* obs, u64 obs_bstride, u32 n_obs,
                           u64* out, u64 out_bstride, u32 n_get);
hipError_t fri_openings(hipStream_t s, const FriShape& sh, u32 B, const u64* zeta, u64 zeta_bstride, u64* out);
hipError_t fri_final_poly(hipStream_t s, const FriShape& sh, u32 B, const u64* alpha, u64 alpha_bstride, const u64* zeta,
                          u64 zeta_bstride, u64* comp, u64* quot, u64* final_poly);
hipError_t fri_fold_values(hipStream_t s, u32 B, u32 log_m, u32 ab, const u64* in, u64 in_bstride, u64* out, u64 out_bstride,
                           const u64* beta, u64 beta_bstride, u64 shift);
hipError_t fri_fold_coeffs(hipStream_t s, u32 B, u32 n_in, u32 ab, const u64* in, u64 in_bstride, u64* out, u64 out_bstride,
                           const u64* beta, u64 beta_bstride, bool aos_out);
hipError_t fri_soa_to_aos(hipStream_t s, u32 B, u32 n, const u64* in, u64 in_bstride, u32 n_in, u64* out, u64 out_bstride);
// witness: B * FRI_POW_STRIDE words, proof b's result at witness[b * FRI_POW_STRIDE]
#define FRI_POW_STRIDE 16
hipError_t fri_pow(hipStream_t s, int variant, const ChState* st, u32 B, u32 bits, u64* witness);
hipError_t fri_queries(hipStream_t s, const FriShape& sh, const FriLayers& ly, u32 B, u32 num_queries, const u64* chal,
                       u64 chal_bstride, u64* proof, u64 proof_bstride, u64 q_off, u64 q_words);
hipError_t bind_public_inputs(hipStream_t s, u32 B, u64* wires, u64 wires_bstride, u64 n, u32 row, const u64* pi_hash);
hipError_t copy_rows(hipStream_t s, u32 B, const u64* src, u64 src_bstride, u64* dst, u64 dst_bstride, u32 words);
}  // namespace mp2g
