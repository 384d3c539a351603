// Device-resident Fiat-Shamir challenger and the FRI prover kernels for gfx950, batched over
// `B` proofs of one circuit shape (the map-reduce proves thousands of same-shaped leaf proofs:
// batching is what turns a launch-latency-bound 2^12-row proof into bandwidth/ALU-bound work).
//
// Replaces [dep] plonky2 iop/challenger.rs, fri/oracle.rs PolynomialBatch::prove_openings
// (alpha-batching + divide_by_linear), fri/prover.rs (fri_committed_trees, fri_proof_of_work,
// fri_prover_query_rounds) and the opening evaluation of plonk/prover.rs, as reached from
// recursion-framework/src/circuit_builder.rs:308 / wrap_circuit.rs:143.
//
// Differences in *how* (results are identical field elements):
//   * the transcript lives in device memory: caps, openings and final polynomials are absorbed
//     where they were produced, challenges are consumed by the next kernel; no host round trip.
//   * FRI layers are folded in the value domain (size-16 inverse DFT of each coset + Horner in
//     beta/x) directly on bit-reversed evaluations, which are already in leaf order; the
//     reference folds coefficients and re-runs a coset FFT per layer. Coefficients are folded
//     alongside (a 16-term Horner) only to emit the final polynomial.
//   * the proof-of-work search returns the smallest valid witness (the reference's rayon
//     find_any returns an arbitrary one).
#include "fri.h"
#include <cstdlib>
#include "poseidon_wave.cuh"

namespace mp2g {

// ---- challenger ------------------------------------------------------------------------------
template <int V>
__device__ void ch_duplex(ChState& c) {
  for (u32 i = 0; i < c.n_in; i++) c.state[i] = c.in[i];
  c.n_in = 0;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = c.state[i];
  perm<V>(s);
#pragma unroll
  for (int i = 0; i < 12; i++) c.state[i] = s[i];
#pragma unroll
  for (int i = 0; i < 8; i++) c.out[i] = s[i];
  c.n_out = 8;
}
// transcript b: observe obs[b*obs_bstride .. +n_obs), then draw n_get challenges into out[b*out_bstride ..]
template <int V>
__global__ void ch_kernel(ChState* st, u32 B, const u64* obs, u64 obs_bstride, u32 n_obs, u64* out, u64 out_bstride, u32 n_get) {
  u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  ChState c = st[b];
  const u64* o = obs + b * obs_bstride;
  for (u32 i = 0; i < n_obs; i++) {
    c.n_out = 0;
    c.in[c.n_in++] = o[i];
    if (c.n_in == 8) ch_duplex<V>(c);
  }
  u64* dst = out + b * out_bstride;
  for (u32 i = 0; i < n_get; i++) {
    if (c.n_in || !c.n_out) ch_duplex<V>(c);
    dst[i] = c.out[--c.n_out];
  }
  st[b] = c;
}
// Lane-cooperative form for Poseidon2 (poseidon_wave.cuh): 16 lanes per transcript, the state limb l in
// lane l, the pending input / squeezed output buffers in lanes 0..7. A transcript is ~110 sequential
// permutations (64 of them absorbing the 257 openings), so permutation LATENCY is what counts here.
__global__ void __launch_bounds__(64) ch_wave_kernel(ChState* st, u32 B, const u64* obs, u64 obs_bstride, u32 n_obs, u64* out,
                                                     u64 out_bstride, u32 n_get) {
  const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 braw = gid >> 4, l = gid & 15;
  const bool active = braw < B;
  const u32 b = active ? braw : B - 1;  // idle groups shadow the last transcript (loads only)
  ChState* c = st + b;
  u64 x = l < 12 ? c->state[l] : 0;
  u64 inb = l < 8 ? c->in[l] : 0, outb = l < 8 ? c->out[l] : 0;
  u32 n_in = c->n_in, n_out = c->n_out;
  const u64* o = obs + b * obs_bstride;
  auto duplex = [&]() {
    if (l < n_in) x = inb;
    n_in = 0;
    x = wp2_perm(x, (int)l);
    outb = x;
    n_out = 8;
  };
  for (u32 i = 0; i < n_obs; i++) {
    u64 e = o[i];
    n_out = 0;
    if (l == n_in) inb = e;
    if (++n_in == 8) duplex();
  }
  u64* dst = out + b * out_bstride;
  for (u32 i = 0; i < n_get; i++) {
    if (n_in || !n_out) duplex();
    --n_out;
    if (active && l == n_out) dst[i] = outb;
  }
  if (!active) return;
  if (l < 12) c->state[l] = x;
  if (l < 8) { c->in[l] = inb; c->out[l] = outb; }
  if (l == 0) { c->n_in = n_in; c->n_out = n_out; }
}
__global__ void ch_init_kernel(ChState* st, u32 B) {
  u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  ChState c;
  for (int i = 0; i < 12; i++) c.state[i] = 0;
  for (int i = 0; i < 8; i++) { c.in[i] = 0; c.out[i] = 0; }
  c.n_in = c.n_out = 0;
  st[b] = c;
}
hipError_t challenger_init(hipStream_t s, ChState* st, u32 B) {
  hipLaunchKernelGGL(ch_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, st, B);
  return hipGetLastError();
}
hipError_t challenger_step(hipStream_t s, int variant, ChState* st, u32 B, const u64* obs, u64 obs_bstride, u32 n_obs,
                           u64* out, u64 out_bstride, u32 n_get) {
  if (variant == MP2G_POSEIDON2)
    hipLaunchKernelGGL(ch_wave_kernel, dim3((B * 16 + 63) / 64), dim3(64), 0, s, st, B, obs, obs_bstride, n_obs, out, out_bstride, n_get);
  else
    hipLaunchKernelGGL((ch_kernel<MP2G_POSEIDON>), dim3((B + 63) / 64), dim3(64), 0, s, st, B, obs, obs_bstride, n_obs, out, out_bstride, n_get);
  return hipGetLastError();
}

// ---- openings: every polynomial at zeta, the Z polynomials also at g*zeta ---------------------
// grid (n_open, B), block 256. out[b][j] = sum_i c_i x^i as [c0,c1].
__global__ void __launch_bounds__(256) openings_kernel(FriShape sh, const u64* zeta /*[B][2]*/, u64 zeta_bstride, u64* out /*[B][n_open][2]*/) {
  const u32 j = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const u32 n = 1u << sh.log_n;
  u32 o, p;
  gl2 x = gl2_make(zeta[b * zeta_bstride], zeta[b * zeta_bstride + 1]);
  if (j >= sh.n_polys) {
    fri_batch_poly(sh, 1, j - sh.n_polys, o, p);
    x = gl2_scale(x, gl_root_of_unity(sh.log_n));
  } else {
    fri_batch_poly(sh, 0, j, o, p);
  }
  const u64* c = sh.o[o].coeffs + b * sh.o[o].coeff_bstride + ((u64)p << sh.log_n);
  // lane t owns i = t + 256k: Horner in x^256, then weight by x^t
  gl2 x256 = gl2_pow(x, 256), acc = gl2_make(0, 0);
  if (t < n) {
    u32 top = ((n - 1 - t) >> 8);
    for (int k = (int)top; k >= 0; k--) {
      acc = gl2_mul(acc, x256);
      acc.a = gl_add(acc.a, c[t + ((u32)k << 8)]);
    }
    acc = gl2_mul(acc, gl2_pow(x, t));
  }
  __shared__ u64 ra[256], rb[256];
  ra[t] = acc.a; rb[t] = acc.b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)t < s) { ra[t] = gl_add(ra[t], ra[t + s]); rb[t] = gl_add(rb[t], rb[t + s]); }
    __syncthreads();
  }
  if (t == 0) {
    u64* d = out + ((u64)b * (sh.n_polys + sh.zs_count + sh.lookup_count) + j) * 2;
    d[0] = ra[0]; d[1] = rb[0];
  }
}

// ---- batch composition: comp[b][batch][c][i] = sum_j alpha^j f_j[i] ----------------------------
__global__ void __launch_bounds__(256) compose_kernel(FriShape sh, const u64* alpha, u64 alpha_bstride, u64* comp) {
  const u32 n = 1u << sh.log_n;
  const u32 i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, batch = blockIdx.z;
  if (i >= n) return;
  gl2 al = gl2_make(alpha[b * alpha_bstride], alpha[b * alpha_bstride + 1]);
  gl2 acc = gl2_make(0, 0);
  // Horner from the back of the batch: runs of consecutive polynomials of one oracle (fri_batch_poly order)
  const u32 zo = sh.zs_oracle, wz = sh.o[zo].w - sh.lookup_count;
  auto run = [&](u32 o, u32 first, u32 count) {
    const u64* base = sh.o[o].coeffs + b * sh.o[o].coeff_bstride + i;
    for (int p = (int)(first + count) - 1; p >= (int)first; p--) {
      acc = gl2_mul(acc, al);
      acc.a = gl_add(acc.a, base[(u64)p << sh.log_n]);
    }
  };
  if (sh.lookup_count) run(zo, wz, sh.lookup_count);
  if (batch == 0) {
    for (int o = (int)sh.n_oracles - 1; o >= 0; o--) run((u32)o, 0, (u32)o == zo ? wz : sh.o[o].w);
  } else {
    run(zo, 0, sh.zs_count);
  }
  u64* d = comp + (((u64)b * 2 + batch) * 2) * n;
  d[i] = acc.a;
  d[n + i] = acc.b;
}

// ---- divide_by_linear: q_i = sum_{k>i} c_k z^(k-i-1) -------------------------------------------
// grid (2, B), block 1024; comp/quot: [B][2 batches][2 comps][n]
__global__ void __launch_bounds__(1024) divide_kernel(u32 log_n, const u64* zeta, u64 zeta_bstride, const u64* comp, u64* quot) {
  const u32 n = 1u << log_n, batch = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const u32 T = n < 1024 ? n : 1024, S = n / T;
  gl2 z = gl2_make(zeta[b * zeta_bstride], zeta[b * zeta_bstride + 1]);
  if (batch == 1) z = gl2_scale(z, gl_root_of_unity(log_n));
  const u64* c0 = comp + (((u64)b * 2 + batch) * 2) * n;
  const u64* c1 = c0 + n;
  u64* q0 = quot + (((u64)b * 2 + batch) * 2) * n;
  u64* q1 = q0 + n;
  __shared__ u64 la[1024], lb[1024];
  // s_i = c_i + z s_{i+1}; chunk t = [tS, (t+1)S): L_t = sum_k c_{tS+k} z^k
  gl2 L = gl2_make(0, 0);
  if (t < T) {
    for (int k = (int)S - 1; k >= 0; k--) {
      L = gl2_mul(L, z);
      L.a = gl_add(L.a, c0[t * S + k]);
      L.b = gl_add(L.b, c1[t * S + k]);
    }
  }
  // carry_t = s_{(t+1)S} = sum_{u>t} L_u Z^(u-t-1), Z = z^S : suffix scan over v_t = L_{t+1}
  la[t] = 0; lb[t] = 0;
  __syncthreads();
  if (t < T && t >= 1) { la[t - 1] = L.a; lb[t - 1] = L.b; }
  __syncthreads();
  gl2 zp = gl2_pow(z, S);
  for (u32 d = 1; d < T; d <<= 1) {
    gl2 add = gl2_make(0, 0);
    if (t + d < T) add = gl2_mul(zp, gl2_make(la[t + d], lb[t + d]));
    __syncthreads();
    if (t < T) { la[t] = gl_add(la[t], add.a); lb[t] = gl_add(lb[t], add.b); }
    __syncthreads();
    zp = gl2_mul(zp, zp);
  }
  if (t < T) {
    gl2 s = gl2_make(la[t], lb[t]);
    for (int k = (int)S - 1; k >= 0; k--) {
      u32 i = t * S + k;
      s = gl2_mul(s, z);
      s.a = gl_add(s.a, c0[i]);
      s.b = gl_add(s.b, c1[i]);
      if (i >= 1) { q0[i - 1] = s.a; q1[i - 1] = s.b; }
    }
    if (t == T - 1) { q0[n - 1] = 0; q1[n - 1] = 0; }  // pad back to a power of two
  }
}
// final[b][c][i] = q_zeta[i] * alpha^zs_count + q_gzeta[i]
__global__ void __launch_bounds__(256) combine_kernel(u32 log_n, u32 zs_count, const u64* alpha, u64 alpha_bstride, const u64* quot, u64* final_poly) {
  const u32 n = 1u << log_n, i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= n) return;
  gl2 sh = gl2_pow(gl2_make(alpha[b * alpha_bstride], alpha[b * alpha_bstride + 1]), zs_count);
  const u64* q = quot + (u64)b * 4 * n;
  gl2 r = gl2_add(gl2_mul(gl2_make(q[i], q[n + i]), sh), gl2_make(q[2 * n + i], q[3 * n + i]));
  final_poly[(u64)b * 2 * n + i] = r.a;
  final_poly[(u64)b * 2 * n + n + i] = r.b;
}

// ---- arity-16 (generally 2^ab) fold in the value domain ---------------------------------------
// in: [B][2][m] bit-reversed evaluations on shift*<w_m>; out: [B][2][m>>ab] bit-reversed on shift^(2^ab)
template <int AB>
__global__ void __launch_bounds__(256) fold_values_kernel(u32 log_m, const u64* in, u64 in_bstride, u64* out, u64 out_bstride,
                                                            const u64* beta, u64 beta_bstride, u64 shift_inv, u64 w_m_inv, u64 arity_inv) {
  constexpr int A = 1 << AB;
  const u32 m = 1u << log_m, chunks = m >> AB;
  const u32 c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (c >= chunks) return;
  const u64* v0 = in + b * in_bstride + ((u64)c << AB);
  const u64* v1 = v0 + m;
  gl2 x[A];
#pragma unroll
  for (int k = 0; k < A; k++) x[k] = gl2_make(v0[k], v1[k]);
  // x[] is the natural-order coset evaluation vector in bit-reversed order: DIT inverse DFT
  const u64 wA_inv = gl_pow(w_m_inv, m >> AB);  // w_A^-1
#pragma unroll
  for (int s = 1; s <= AB; s++) {
    const int mm = 1 << s, h = mm >> 1;
    u64 wstep = wA_inv;
    for (int e = s; e < AB; e++) wstep = gl_sqr(wstep);  // w_{2^s}^-1
    u64 w = 1;
#pragma unroll
    for (int j = 0; j < h; j++) {
#pragma unroll
      for (int k = 0; k < A; k += mm) {
        gl2 t = j == 0 ? x[k + j + h] : gl2_scale(x[k + j + h], w);
        gl2 u = x[k + j];
        x[k + j] = gl2_add(u, t);
        x[k + j + h] = gl2_sub(u, t);
      }
      w = gl_mul(w, wstep);
    }
  }
  // a_j = x[j]/A = x0^j P_j(y);  F = sum_j (beta/x0)^j a_j
  u64 x0_inv = gl_mul(shift_inv, gl_pow(w_m_inv, bitrev32(c, log_m - AB)));
  gl2 g = gl2_scale(gl2_make(beta[b * beta_bstride], beta[b * beta_bstride + 1]), x0_inv);
  gl2 acc = x[A - 1];
#pragma unroll
  for (int j = A - 2; j >= 0; j--) acc = gl2_add(gl2_mul(acc, g), x[j]);
  acc = gl2_scale(acc, arity_inv);
  u64* o = out + b * out_bstride;
  o[c] = acc.a;
  o[chunks + c] = acc.b;
}
// coefficient fold: new[i] = sum_j c[i*A + j] beta^j. aos_out != 0: write [i][2] interleaved (final poly)
__global__ void __launch_bounds__(256) fold_coeffs_kernel(u32 n_out, u32 ab, const u64* in, u64 in_bstride, u32 n_in, u64* out, u64 out_bstride,
                                                            const u64* beta, u64 beta_bstride, int aos_out) {
  const u32 i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= n_out) return;
  const u32 A = 1u << ab;
  const u64* c0 = in + b * in_bstride + ((u64)i << ab);
  const u64* c1 = c0 + n_in;
  gl2 be = gl2_make(beta[b * beta_bstride], beta[b * beta_bstride + 1]);
  gl2 acc = gl2_make(0, 0);
  for (int j = (int)A - 1; j >= 0; j--) acc = gl2_add(gl2_mul(acc, be), gl2_make(c0[j], c1[j]));
  u64* o = out + b * out_bstride;
  if (aos_out) { o[2 * i] = acc.a; o[2 * i + 1] = acc.b; }
  else { o[i] = acc.a; o[n_out + i] = acc.b; }
}
// SoA [2][n] -> AoS [n][2] (no-fold case: final polynomial = the composed polynomial itself)
__global__ void soa_to_aos_kernel(u32 n, const u64* in, u64 in_bstride, u32 n_in, u64* out, u64 out_bstride) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= n) return;
  out[b * out_bstride + 2 * i] = in[b * in_bstride + i];
  out[b * out_bstride + 2 * i + 1] = in[b * in_bstride + n_in + i];
}

// ---- proof of work ----------------------------------------------------------------------------
// witness[b * POW_STRIDE] = min { w : perm(state with w at position n_in)[7] has >= bits leading zeros }.
// Several blocks of 256 lanes per proof (fri_pow sizes the grid) sweep the candidates in order (four blocks share a CU, so one block's poll and
// barrier leave the ALUs to the other three; with one 1024-lane block per CU 37 % of the wave cycles were parked: 3.45 -> 3.08 ms
// per 64 base proofs. Handing the candidates out in chunks from a per-proof counter so that the blocks of finished proofs help the
// stragglers was measured too and lost, 4.6-11.9 ms: the helpers gang up on a proof and overshoot its nonce). Blocks of one proof
// talk through one word: a finder publishes with atomicMin, and once per sweep lane 0 of every
// block reads it back with a returning (no-op) atomicMin -- the per-XCD L2s are not coherent, a
// plain or sc1 load of a word that another XCD updates atomically can stay stale for seconds,
// while an atomic executes at the coherence point. Each proof's word sits in its own 128-B line.
#define POW_THREADS 256
#define POW_STRIDE 16
template <int V>
__global__ void __launch_bounds__(POW_THREADS) pow_kernel(const ChState* st, u32 bits, unsigned long long* witness) {
  const u32 b = blockIdx.y;
  const ChState& c = st[b];
  unsigned long long* wit = witness + (u64)b * POW_STRIDE;
  __shared__ unsigned long long s_best;
  u64 base[12];
#pragma unroll
  for (int i = 0; i < 12; i++) base[i] = c.state[i];
  const u32 pos = c.n_in;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if ((u32)i < pos) base[i] = c.in[i];
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 cand = (u64)blockIdx.x * blockDim.x + threadIdx.x;; cand += stride) {
    if (threadIdx.x == 0) s_best = atomicMin(wit, ~0ull);
    __syncthreads();
    const unsigned long long best = s_best;
    __syncthreads();
    // block-uniform exit: the block's smallest candidate of this sweep is already beaten
    if (cand - threadIdx.x > best) break;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = base[i];
#pragma unroll
    for (int i = 0; i < 8; i++)
      if ((u32)i == pos) s[i] = cand;
    perm<V>(s);
    if (cand < GL_P && (bits == 0 || (s[7] >> (64 - bits)) == 0)) atomicMin(wit, (unsigned long long)cand);
  }
}
__global__ void fill_u64_kernel(u64* p, u64 v, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- query rounds -----------------------------------------------------------------------------
// grid (num_queries, B); writes section q of proof b in the flat layout of include/mp2g.h
__global__ void __launch_bounds__(256) query_kernel(FriShape sh, FriLayers ly, const u64* chal /*[B][num_queries]*/, u64 chal_bstride,
                                                      u64* proof, u64 proof_bstride, u64 q_off, u64 q_words) {
  const u32 q = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const u32 lg = sh.log_n + sh.rate_bits;
  const u64 N = (u64)1 << lg;
  u64 x = chal[b * chal_bstride + q] % N;
  u64* o = proof + b * proof_bstride + q_off + q * q_words;
  const u32 depth = lg - sh.cap_h;
  for (u32 oi = 0; oi < sh.n_oracles; oi++) {
    const OracleRef& r = sh.o[oi];
    const u64* vals = r.values + b * r.value_bstride;
    for (u32 p = t; p < r.w; p += 256) o[p] = vals[(u64)p * N + x];
    o += r.w;
    const u64* lv = r.levels + b * r.level_bstride;
    for (u32 e = t; e < depth * 4; e += 256) {
      u32 l = e >> 2, k = e & 3;
      u64 off = 0;
      for (u32 j = 0; j < l; j++) off += (u64)4 << (lg - j);
      o[e] = lv[off + 4 * ((x >> l) ^ 1) + k];
    }
    o += depth * 4;
  }
  u32 clg = lg;
  for (u32 li = 0; li < ly.n_layers; li++) {
    const u32 ab = ly.arity_bits[li], A = 1u << ab;
    const u64 m = (u64)1 << clg;
    x >>= ab;
    clg -= ab;
    const u64* v = ly.values[li] + b * ly.value_bstride[li];
    for (u32 e = t; e < 2 * A; e += 256) o[e] = v[(e & 1) * m + (x << ab) + (e >> 1)];
    o += 2 * A;
    const u32 d2 = clg - sh.cap_h;
    const u64* lv = ly.levels[li] + b * ly.level_bstride[li];
    for (u32 e = t; e < d2 * 4; e += 256) {
      u32 l = e >> 2, k = e & 3;
      u64 off = 0;
      for (u32 j = 0; j < l; j++) off += (u64)4 << (clg - j);
      o[e] = lv[off + 4 * ((x >> l) ^ 1) + k];
    }
    o += d2 * 4;
  }
}
// copy caps of tree b (last cap_words of its levels) to dst[b*dst_bstride ..]
__global__ void copy_rows_kernel(const u64* src, u64 src_bstride, u64* dst, u64 dst_bstride, u32 words) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i < words) dst[b * dst_bstride + i] = src[b * src_bstride + i];
}

// PublicInputGate's generator: wires[b][c][row] = pi_hash[b][c], c < 4
__global__ void bind_pi_kernel(u64* wires, u64 wires_bstride, u64 n, u32 row, const u64* pi_hash, u32 B) {
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 4 * B) wires[(t >> 2) * wires_bstride + (u64)(t & 3) * n + row] = pi_hash[t];
}

// ---- launchers --------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t fri_openings(hipStream_t s, const FriShape& sh, u32 B, const u64* zeta, u64 zeta_bstride, u64* out) {
  hipLaunchKernelGGL(openings_kernel, dim3(sh.n_polys + sh.zs_count + sh.lookup_count, B), dim3(256), 0, s, sh, zeta, zeta_bstride, out);
  return hipGetLastError();
}
hipError_t fri_final_poly(hipStream_t s, const FriShape& sh, u32 B, const u64* alpha, u64 alpha_bstride, const u64* zeta, u64 zeta_bstride,
                          u64* comp, u64* quot, u64* final_poly) {
  const u32 n = 1u << sh.log_n;
  hipLaunchKernelGGL(compose_kernel, dim3((n + 255) / 256, B, 2), dim3(256), 0, s, sh, alpha, alpha_bstride, comp);
  hipLaunchKernelGGL(divide_kernel, dim3(2, B), dim3(1024), 0, s, sh.log_n, zeta, zeta_bstride, comp, quot);
  hipLaunchKernelGGL(combine_kernel, dim3((n + 255) / 256, B), dim3(256), 0, s, sh.log_n, sh.zs_count + sh.lookup_count, alpha, alpha_bstride, quot,
                     final_poly);
  return hipGetLastError();
}
hipError_t fri_fold_values(hipStream_t s, u32 B, u32 log_m, u32 ab, const u64* in, u64 in_bstride, u64* out, u64 out_bstride,
                           const u64* beta, u64 beta_bstride, u64 shift) {
  const u32 chunks = (1u << log_m) >> ab;
  const u64 shift_inv = gl_inv(shift), w_m_inv = gl_inv(gl_root_of_unity(log_m)), a_inv = gl_inv((u64)1 << ab);
  dim3 g((chunks + 255) / 256, B), bl(256);
#define FV(N) case N: hipLaunchKernelGGL((fold_values_kernel<N>), g, bl, 0, s, log_m, in, in_bstride, out, out_bstride, beta, beta_bstride, shift_inv, w_m_inv, a_inv); break;
  switch (ab) { FV(1) FV(2) FV(3) FV(4) default: return hipErrorInvalidValue; }
#undef FV
  return hipGetLastError();
}
hipError_t fri_fold_coeffs(hipStream_t s, u32 B, u32 n_in, u32 ab, const u64* in, u64 in_bstride, u64* out, u64 out_bstride,
                           const u64* beta, u64 beta_bstride, bool aos_out) {
  const u32 n_out = n_in >> ab;
  hipLaunchKernelGGL(fold_coeffs_kernel, dim3((n_out + 255) / 256, B), dim3(256), 0, s, n_out, ab, in, in_bstride, n_in, out, out_bstride, beta, beta_bstride, aos_out ? 1 : 0);
  return hipGetLastError();
}
hipError_t fri_soa_to_aos(hipStream_t s, u32 B, u32 n, const u64* in, u64 in_bstride, u32 n_in, u64* out, u64 out_bstride) {
  hipLaunchKernelGGL(soa_to_aos_kernel, dim3((n + 255) / 256, B), dim3(256), 0, s, n, in, in_bstride, n_in, out, out_bstride);
  return hipGetLastError();
}
hipError_t fri_pow(hipStream_t s, int variant, const ChState* st, u32 B, u32 bits, u64* witness) {
  hipLaunchKernelGGL(fill_u64_kernel, dim3((B * POW_STRIDE + 63) / 64), dim3(64), 0, s, witness, ~(u64)0, B * POW_STRIDE);
  // blocks per proof. The search returns the SMALLEST witness, so every candidate below it is evaluated whatever the order -- and so is
  // the rest of the sweep it lies in: with G candidates per sweep and proof the expected work is 2^bits + G / 2 permutations. The
  // launch is sized to 2^19 lanes over all proofs of the batch (MP2G_POW_LANES overrides it for A/B runs) instead of 2^20 as in
  // rounds 1-4 (a batch of 32 proofs: G = 2^14 instead of 2^15, 1.125 x 2^16 permutations a proof instead of 1.25 x). Measured on
  // the table build (profiles/r05/variants_ab.txt, two alternating repetitions): 2^20 lanes 841.8 / 839.4 proofs/s, 2^19 845.0 /
  // 842.3, 2^18 840.8 / 838.3, 2^17 824.1 / 823.6 -- narrower sweeps save candidates and pay for it in polls and part-filled
  // launches; the search is 7 % of a build's VALU instructions either way. A lone proof keeps sweeps of 2^18 candidates
  static const u32 lanes = [] { const char* e = getenv("MP2G_POW_LANES"); const long v = e ? atol(e) : 0; return (u32)(v >= 256 ? v : 1 << 19); }();
  u32 blocks = lanes / POW_THREADS / (B ? B : 1);
  if (blocks < 8) blocks = 8;
  if (blocks > 1024) blocks = 1024;
  dim3 g(blocks, B), bl(POW_THREADS);
  if (variant == MP2G_POSEIDON2) hipLaunchKernelGGL((pow_kernel<MP2G_POSEIDON2>), g, bl, 0, s, st, bits, (unsigned long long*)witness);
  else hipLaunchKernelGGL((pow_kernel<MP2G_POSEIDON>), g, bl, 0, s, st, bits, (unsigned long long*)witness);
  return hipGetLastError();
}
hipError_t fri_queries(hipStream_t s, const FriShape& sh, const FriLayers& ly, u32 B, u32 num_queries, const u64* chal, u64 chal_bstride,
                       u64* proof, u64 proof_bstride, u64 q_off, u64 q_words) {
  if (!num_queries) return hipSuccess;
  hipLaunchKernelGGL(query_kernel, dim3(num_queries, B), dim3(256), 0, s, sh, ly, chal, chal_bstride, proof, proof_bstride, q_off, q_words);
  return hipGetLastError();
}
hipError_t bind_public_inputs(hipStream_t s, u32 B, u64* wires, u64 wires_bstride, u64 n, u32 row, const u64* pi_hash) {
  hipLaunchKernelGGL(bind_pi_kernel, dim3((4 * B + 63) / 64), dim3(64), 0, s, wires, wires_bstride, n, row, pi_hash, B);
  return hipGetLastError();
}
hipError_t copy_rows(hipStream_t s, u32 B, const u64* src, u64 src_bstride, u64* dst, u64 dst_bstride, u32 words) {
  if (!words) return hipSuccess;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((words + 255) / 256, B), dim3(256), 0, s, src, src_bstride, dst, dst_bstride, words);
  return hipGetLastError();
}
}  // namespace mp2g
