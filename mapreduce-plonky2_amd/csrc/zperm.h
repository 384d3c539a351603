// Launcher of the permutation-argument kernels (zperm.hip).
#pragma once
#include "gl.cuh"
namespace mp2g {
// chunk_q: scratch of B * nc * (num_routed/degree) * n words; out: [B][nc * num_routed/degree][n]
hipError_t zpp_compute(hipStream_t s, u32 B, const u64* wires, u64 wires_bstride, const u64* sigmas, u32 log_n, u32 num_routed,
                       u32 degree, const u64* betas, const u64* gammas, u64 chal_bstride, u32 nc, u64* chunk_q, u64* out,
                       u64 out_bstride);
}  // namespace mp2g
