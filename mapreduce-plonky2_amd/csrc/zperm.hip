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
#define ZP_MAX_ROUTED 256
// k_j = g^j (cosets.rs get_unique_coset_shifts) for the block, in LDS
__device__ __forceinline__ void fill_k_is(u64* kis, u32 num_routed) {
  for (u32 j = threadIdx.x; j < num_routed; j += blockDim.x) kis[j] = gl_pow(GL_MULT_GEN, j);
  __syncthreads();
}

// chunk_q[((b*nc + c)*chunks + k)*n + i] = prod_{j in chunk k} (w_j + beta k_j x + gamma) / (w_j + beta sigma_j + gamma)
__global__ void __launch_bounds__(256) zpp_chunk_kernel(const u64* __restrict__ wires, u64 wires_bstride, const u64* __restrict__ sigmas,
                                                        u32 log_n, u32 num_routed, u32 degree, const u64* __restrict__ betas,
                                                        const u64* __restrict__ gammas, u64 chal_bstride, u32 nc, u64* __restrict__ chunk_q) {
  const u32 n = 1u << log_n, i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  __shared__ u64 kis[ZP_MAX_ROUTED];
  fill_k_is(kis, num_routed);
  if (i >= n) return;
  const u32 chunks = num_routed / degree;
  const u64 beta = betas[b * chal_bstride + c], gamma = gammas[b * chal_bstride + c];
  const u64 bx = gl_mul(beta, gl_pow(gl_root_of_unity(log_n), i));
  const u64* w = wires + b * wires_bstride + i;
  const u64* sg = sigmas + i;
  u64 num[ZP_MAX_CHUNKS], den[ZP_MAX_CHUNKS];
  u32 j = 0;
  for (u32 k = 0; k < chunks; k++) {
    // the running products stay weak representatives (gl.cuh); only the denominator, which is tested
    // against zero, is canonicalised
    u64 pn = 1, pd = 1;
    for (u32 t = 0; t < degree; t++, j++) {
      u64 wv = w[(u64)j << log_n];
      const u64 wg = gl_addw(wv, gamma);  // shared by numerator and denominator; it rides in the products' addend slots
      pn = gl_mulw(pn, gl_mul_addw(bx, kis[j], wg));
      pd = gl_mulw(pd, gl_mul_addw(beta, sg[(u64)j << log_n], wg));
    }
    num[k] = pn; den[k] = gl_canon(pd);
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

// ---- quotient values: the permutation terms (+ the gate term left in q by gates.hip) -----------
// plonk/prover.rs compute_quotient_polys + vanishing_poly.rs eval_vanishing_poly_base_batch with no
// gate constraints: terms = [L_0(x)(Z_c(x) - 1)]_c ++ [prev * prod(num) - next * prod(den)] per chunk and
// challenge; q_a(x) = sum_k terms[k] alpha_a^k / Z_H(x) on the coset g<w_N>, N = 8n.
// One lane per LDE column p in MEMORY order (the matrices are stored bit-reversed, so the ~180 loads per
// lane are coalesced 8 B/lane streams); the lane's point is x = g w_N^i with i = bitrev(p), Z(g x) sits in
// column bitrev(i + 8), and only the nc results are scattered to their natural slot q[i]. (A natural-order
// mapping gathers every operand from a different cache line and ran 28 % slower end to end.)
__global__ void __launch_bounds__(256) quotient_perm_kernel(const u64* __restrict__ W, u64 w_bstride, const u64* __restrict__ S,
                                                            const u64* __restrict__ Z, u64 z_bstride, u32 log_n, u32 num_routed,
                                                            u32 degree, const u64* __restrict__ bg, u64 bg_bstride,
                                                            const u64* __restrict__ alphas, u64 al_bstride, u32 nc, bool gates,
                                                            u64* __restrict__ q) {
  const u32 lg = log_n + 3;
  const u64 N = (u64)1 << lg;
  const u32 p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  const u32 i = bitrev32(p, lg);
  __shared__ u64 zh_inv[8];
  __shared__ u64 kis[ZP_MAX_ROUTED];
  fill_k_is(kis, num_routed);
  if (threadIdx.x < 8) {
    u64 gn = gl_pow(GL_MULT_GEN, (u64)1 << log_n);
    zh_inv[threadIdx.x] = gl_inv(gl_sub(gl_mul(gn, gl_pow(gl_root_of_unity(3), threadIdx.x)), 1));
  }
  __syncthreads();
  if (p >= N) return;
  const u32 chunks = num_routed / degree, num_prods = chunks - 1;
  const u64 pn = bitrev32((i + 8) & (u32)(N - 1), lg);
  const u64 x = gl_mul(GL_MULT_GEN, gl_pow(gl_root_of_unity(lg), i));
  const u64* w = W + b * w_bstride + p;
  const u64* sg = S + p;
  const u64* z = Z + b * z_bstride;
  u64 acc[2] = {0, 0}, apow[2] = {1, 1}, al[2];
  for (u32 a = 0; a < nc; a++) al[a] = alphas[b * al_bstride + a];
  auto push = [&](u64 term) {
    for (u32 a = 0; a < nc; a++) {
      acc[a] = gl_mul_add(term, apow[a], acc[a]);  // term: any representative
      apow[a] = gl_mul(apow[a], al[a]);
    }
  };
  // L_0(x) = (x^n - 1) / (n (x - 1)); x^n - 1 = 1 / zh_inv
  const u64 zh = gl_inv(zh_inv[i & 7]);
  const u64 l0 = gl_mul(zh, gl_inv(gl_mul(((u64)1 << log_n) % GL_P, gl_sub(x, 1))));
  for (u32 c = 0; c < nc; c++) push(gl_mul(l0, gl_sub(z[((u64)c << lg) + p], 1)));
  for (u32 c = 0; c < nc; c++) {
    const u64 beta = bg[b * bg_bstride + c], gamma = bg[b * bg_bstride + nc + c];
    const u64 bx = gl_mul(beta, x);
    const u64* pp = z + ((u64)(nc + c * num_prods) << lg);
    u32 j = 0;
    u64 prev = z[((u64)c << lg) + p];
    for (u32 k = 0; k < chunks; k++) {
      u64 num = 1, den = 1;
      for (u32 t = 0; t < degree; t++, j++) {
        u64 wv = w[(u64)j << lg];
        const u64 wg = gl_addw(wv, gamma);  // weak running products; w + gamma rides in the addend slots of beta k_j x / beta sigma_j
        num = gl_mulw(num, gl_mul_addw(bx, kis[j], wg));
        den = gl_mulw(den, gl_mul_addw(beta, sg[(u64)j << lg], wg));
      }
      u64 next = k == chunks - 1 ? z[((u64)c << lg) + pn] : pp[((u64)k << lg) + p];
      push(gl_sub(gl_mul(prev, num), gl_mul(next, den)));
      prev = next;
    }
  }
  // gate constraint terms follow (gates.hip left sum_j alpha^j C_j(x) in this lane's slot)
  if (gates)
    for (u32 a = 0; a < nc; a++) acc[a] = gl_add(acc[a], gl_mul(apow[a], q[(((u64)b * nc + a) << lg) + i]));
  for (u32 a = 0; a < nc; a++) q[(((u64)b * nc + a) << lg) + i] = gl_mul(acc[a], zh_inv[i & 7]);
}
hipError_t quotient_perm_values(hipStream_t s, u32 B, const u64* W, u64 w_bstride, const u64* S, const u64* Z, u64 z_bstride,
                                u32 log_n, u32 num_routed, u32 degree, const u64* bg, u64 bg_bstride, const u64* alphas,
                                u64 al_bstride, u32 nc, bool gates, u64* q) {
  if (nc < 1 || nc > 2 || !degree || num_routed % degree || num_routed > ZP_MAX_ROUTED) return hipErrorInvalidValue;
  const u64 N = (u64)8 << log_n;
  hipLaunchKernelGGL(quotient_perm_kernel, dim3((u32)((N + 255) / 256), B), dim3(256), 0, s, W, w_bstride, S, Z, z_bstride, log_n,
                     num_routed, degree, bg, bg_bstride, alphas, al_bstride, nc, gates, q);
  return hipGetLastError();
}

// Copy constraints hold iff the running product returns to one: Z(g^(n-1)) times the last row's chunk
// quotients must be 1 for every challenge (w.h.p. over beta, gamma). flags[b] |= 1 otherwise.
__global__ void zpp_wrap_check_kernel(const u64* __restrict__ chunk_q, const u64* __restrict__ zs, u64 zs_bstride, u32 log_n, u32 chunks,
                                      u32 nc, u32 B, u32* __restrict__ flags) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * nc) return;
  const u32 b = t / nc, c = t % nc, last = (1u << log_n) - 1;
  u64 acc = zs[b * zs_bstride + ((u64)c << log_n) + last];
  const u64* q = chunk_q + ((u64)(b * nc + c) * chunks << log_n);
  for (u32 k = 0; k < chunks; k++) acc = gl_mul(acc, q[((u64)k << log_n) + last]);
  if (acc != 1) atomicOr(&flags[b], 1u);
}
hipError_t zpp_wrap_check(hipStream_t s, u32 B, const u64* chunk_q, const u64* zs, u64 zs_bstride, u32 log_n, u32 chunks, u32 nc,
                          u32* flags) {
  hipLaunchKernelGGL(zpp_wrap_check_kernel, dim3((B * nc + 63) / 64), dim3(64), 0, s, chunk_q, zs, zs_bstride, log_n, chunks, nc, B, flags);
  return hipGetLastError();
}

hipError_t zpp_compute(hipStream_t s, u32 B, const u64* wires, u64 wires_bstride, const u64* sigmas, u32 log_n, u32 num_routed,
                       u32 degree, const u64* betas, const u64* gammas, u64 chal_bstride, u32 nc, u64* chunk_q, u64* out,
                       u64 out_bstride) {
  if (!degree || num_routed % degree || num_routed / degree > ZP_MAX_CHUNKS || num_routed / degree < 1 || num_routed > ZP_MAX_ROUTED)
    return hipErrorInvalidValue;
  const u32 n = 1u << log_n, chunks = num_routed / degree;
  hipLaunchKernelGGL(zpp_chunk_kernel, dim3((n + 255) / 256, nc, B), dim3(256), 0, s, wires, wires_bstride, sigmas, log_n, num_routed,
                     degree, betas, gammas, chal_bstride, nc, chunk_q);
  hipLaunchKernelGGL(zpp_scan_kernel, dim3(nc, B), dim3(1024), 0, s, chunk_q, log_n, chunks, nc, out, out_bstride);
  return hipGetLastError();
}
}  // namespace mp2g
