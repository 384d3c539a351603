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
  OracleRef o[8];
};
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
