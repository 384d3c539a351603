// Launcher of the permutation-argument kernels (zperm.hip).
#pragma once
#include "gl.cuh"
namespace mp2g {
// chunk_q: scratch of B * nc * (num_routed/degree) * n words; out: [B][nc * num_routed/degree][n]
hipError_t zpp_compute(hipStream_t s, u32 B, const u64* wires, u64 wires_bstride, const u64* sigmas, u32 log_n, u32 num_routed,
                       u32 degree, const u64* betas, const u64* gammas, u64 chal_bstride, u32 nc, u64* chunk_q, u64* out,
                       u64 out_bstride);
// q[B][nc][8n] (natural order) = vanishing terms of the permutation argument / Z_H on the coset g<w_8n>;
// W/S/Z are the bit-reversed LDE value matrices of wires, sigmas and Z/partial products; bg holds
// betas[nc] then gammas[nc] per proof. gates: q already holds the alpha-reduced gate constraints of every
// point (gate_constraints_lde), which continue the alpha powers after the permutation terms.
hipError_t quotient_perm_values(hipStream_t s, u32 B, const u64* W, u64 w_bstride, const u64* S, const u64* Z, u64 z_bstride,
                                u32 log_n, u32 num_routed, u32 degree, const u64* bg, u64 bg_bstride, const u64* alphas,
                                u64 al_bstride, u32 nc, bool gates, u64* q);
// flags[b] |= 1 when the permutation product of proof b does not wrap to one (a violated copy constraint);
// chunk_q / zs as produced by zpp_compute
hipError_t zpp_wrap_check(hipStream_t s, u32 B, const u64* chunk_q, const u64* zs, u64 zs_bstride, u32 log_n, u32 chunks, u32 nc,
                          u32* flags);
}  // namespace mp2g
