// Permutation argument on gfx950: the Z polynomial and its partial products, batched over proofs.
//
// Replaces [dep] plonky2 plonk/prover.rs all_wires_permutation_partial_products /
// wires_permutation_partial_products_and_zs (+ plonk_common.rs quotient_chunk_products,
// partial_products_and_z_gx; cosets.rs get_unique_coset_shifts: k_j = g^j), i.e. what prove() does
// between the wires commitment and the Z commitment (recursion-framework/src/circuit_builder.rs:308).
// Output order is the one prove() commits: Z of every challenge first, then each challenge's
// partial products.
//
// Two launches: (1) one lane per (row, challenge, proof) multiplies the 2*num_routed linear factors
// into num_routed/degree chunk quotients with one batched inversion; (2) one block per
// (challenge, proof) turns the per-row products into running products with an in-LDS scan.
#include "zperm.h"

namespace mp2g {

#define ZP_MAX_CHUNKS 16

// chunk_q[((b*nc + c)*chunks + k)*n + i] = prod_{j in chunk k} (w_j + beta k_j x + gamma) / (w_j + beta sigma_j + gamma)
__global__ void __launch_bounds__(256) zpp_chunk_kernel(const u64* __restrict__ wires, u64 wires_bstride, const u64* __restrict__ sigmas,
                                                        u32 log_n, u32 num_routed, u32 degree, const u64* __restrict__ betas,
                                                        const u64* __restrict__ gammas, u64 chal_bstride, u32 nc, u64* __restrict__ chunk_q) {
  const u32 n = 1u << log_n, i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (i >= n) return;
  const u32 chunks = num_routed / degree;
  const u64 beta = betas[b * chal_bstride + c], gamma = gammas[b * chal_bstride + c];
  const u64 bx = gl_mul(beta, gl_pow(gl_root_of_unity(log_n), i));
  const u64* w = wires + b * wires_bstride + i;
  const u64* sg = sigmas + i;
  u64 num[ZP_MAX_CHUNKS], den[ZP_MAX_CHUNKS];
  u64 kj = 1;  // k_j = g^j
  u32 j = 0;
  for (u32 k = 0; k < chunks; k++) {
    u64 pn = 1, pd = 1;
    for (u32 t = 0; t < degree; t++, j++) {
      u64 wv = w[(u64)j << log_n];
      pn = gl_mul(pn, gl_add(gl_add(wv, gl_mul(bx, kj)), gamma));
      pd = gl_mul(pd, gl_add(gl_add(wv, gl_mul(beta, sg[(u64)j << log_n])), gamma));
      kj = gl_mul(kj, GL_MULT_GEN);
    }
    num[k] = pn; den[k] = pd;
  }
  // Montgomery batch inversion of the chunk denominators (0 -> 0, as inverse_or_zero would)
  u64 pre[ZP_MAX_CHUNKS];
  u64 acc = 1;
  for (u32 k = 0; k < chunks; k++) { pre[k] = acc; acc = gl_mul(acc, den[k] ? den[k] : 1); }
  u64 inv = gl_inv(acc);
  u64* out = chunk_q + ((u64)(b * nc + c) * chunks << log_n) + i;
  for (int k = (int)chunks - 1; k >= 0; k--) {
    u64 dinv = den[k] ? gl_mul(inv, pre[k]) : 0;
    inv = gl_mul(inv, den[k] ? den[k] : 1);
    out[(u64)k << log_n] = gl_mul(num[k], dinv);
  }
}
// Z[i] = prod_{i' < i} prod_k q[k][i'],  pp[k][i] = Z[i] * prod_{k' <= k} q[k'][i]
__global__ void __launch_bounds__(1024) zpp_scan_kernel(const u64* __restrict__ chunk_q, u32 log_n, u32 chunks, u32 nc, u64* __restrict__ out,
                                                        u64 out_bstride) {
  const u32 n = 1u << log_n, c = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const u32 T = n < 1024 ? n : 1024, S = n / T;
  const u32 num_prods = chunks - 1;
  const u64* q = chunk_q + ((u64)(b * nc + c) * chunks << log_n);
  __shared__ u64 sc[1024];
  u64 local = 1;
  if (t < T)
    for (u32 r = 0; r < S; r++)
      for (u32 k = 0; k < chunks; k++) local = gl_mul(local, q[((u64)k << log_n) + t * S + r]);
  sc[t] = local;
  __syncthreads();
  // inclusive multiplicative scan (Hillis-Steele)
  for (u32 d = 1; d < T; d <<= 1) {
    u64 v = (t >= d && t < T) ? sc[t - d] : 1;
    __syncthreads();
    if (t < T) sc[t] = gl_mul(sc[t], v);
    __syncthreads();
  }
  if (t >= T) return;
  u64 z = t ? sc[t - 1] : 1;  // exclusive prefix = Z at the first row of this lane's run
  u64* zrow = out + b * out_bstride + ((u64)c << log_n);
  u64* pp = out + b * out_bstride + ((u64)nc << log_n) + ((u64)c * num_prods << log_n);
  for (u32 r = 0; r < S; r++) {
    u32 i = t * S + r;
    zrow[i] = z;
    u64 acc = z;
    for (u32 k = 0; k < chunks; k++) {
      acc = gl_mul(acc, q[((u64)k << log_n) + i]);
      if (k < num_prods) pp[((u64)k << log_n) + i] = acc;
    }
    z = acc;
  }
}

hipError_t zpp_compute(hipStream_t s, u32 B, const u64* wires, u64 wires_bstride, const u64* sigmas, u32 log_n, u32 num_routed,
                       u32 degree, const u64* betas, const u64* gammas, u64 chal_bstride, u32 nc, u64* chunk_q, u64* out,
                       u64 out_bstride) {
  if (!degree || num_routed % degree || num_routed / degree > ZP_MAX_CHUNKS || num_routed / degree < 1) return hipErrorInvalidValue;
  const u32 n = 1u << log_n, chunks = num_routed / degree;
  hipLaunchKernelGGL(zpp_chunk_kernel, dim3((n + 255) / 256, nc, B), dim3(256), 0, s, wires, wires_bstride, sigmas, log_n, num_routed,
                     degree, betas, gammas, chal_bstride, nc, chunk_q);
  hipLaunchKernelGGL(zpp_scan_kernel, dim3(nc, B), dim3(1024), 0, s, chunk_q, log_n, chunks, nc, out, out_bstride);
  return hipGetLastError();
}
}  // namespace mp2g
