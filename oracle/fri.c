// TEST INFRASTRUCTURE -- CPU oracle (see gl.h header). Never linked into the product.
//
// Challenger, FRI prover (commit / PoW / query), batch-opening composition, FRI verifier and
// the transcript of prove() around them.
// Follows (absent [dep] sources, restated):
//   plonky2/src/iop/challenger.rs        (overwrite-mode duplex; get_challenge pops from the back)
//   plonky2/src/fri/oracle.rs            (PolynomialBatch::prove_openings: alpha-batching, divide_by_linear)
//   plonky2/src/fri/prover.rs            (fri_committed_trees, fri_proof_of_work, fri_prover_query_rounds)
//   plonky2/src/fri/reduction_strategies.rs (ConstantArityBits(4,5))
//   plonky2/src/fri/verifier.rs          (fri_combine_initial, compute_evaluation, final-poly check)
//   plonky2/src/plonk/prover.rs          (observe order: circuit digest, H(public inputs), wires cap,
//                                          betas/gammas, zs cap, alphas, quotient cap, zeta, openings)
// Called from the reference at recursion-framework/src/circuit_builder.rs:308 and
// universal_verifier_gadget/wrap_circuit.rs:143 (`prove`).
#include "fri.h"
#include "gates.h"
#include <stdlib.h>
#include <stdio.h>
#include <omp.h>

void orc_perm(int variant, gl_t s[12]);
void orc_merkle_build(int variant, const gl_t* leaves, size_t leaf_len, unsigned log_leaves, unsigned cap_h, gl_t* levels);
size_t orc_merkle_levels_len(unsigned log_leaves, unsigned cap_h);
const gl_t* orc_merkle_cap_ptr(const gl_t* levels, unsigned log_leaves, unsigned cap_h);
void orc_merkle_prove(const gl_t* levels, unsigned log_leaves, unsigned cap_h, size_t idx, gl_t* siblings);
int orc_merkle_verify(int variant, const gl_t* leaf, size_t leaf_len, size_t idx, const gl_t* siblings, unsigned n_sib, const gl_t* cap);
void orc_fft(gl_t* a, unsigned log_n, int inverse);
void orc_coset_fft(gl_t* a, unsigned log_n, gl_t shift);
void orc_lde_leaves(const gl_t* coeffs, unsigned log_n, size_t w, unsigned rate_bits, gl_t* leaves);
size_t orc_bitrev(size_t x, unsigned bits);
void orc_partial_products_and_zs(const gl_t* wires, const gl_t* sigmas, unsigned log_n, unsigned num_routed,
                                 unsigned degree, const gl_t* betas, const gl_t* gammas, unsigned nc, gl_t* out);
void orc_quotient_perm(const gl_t* wires_coeffs, const gl_t* sigma_coeffs, const gl_t* zs_coeffs, unsigned log_n,
                       unsigned num_routed, unsigned degree, const gl_t* betas, const gl_t* gammas, const gl_t* alphas,
                       unsigned nc, gl_t* out);
// the gate part of a circuit: descriptors, the constant polynomials (selectors first) and the wire count
typedef struct {
  const orc_gate* gates;
  unsigned n_gates, num_selectors, num_constants, wires_w;
  const gl_t* const_coeffs;  // [num_constants][n]
  const gl_t* pi_hash;       // [4]
  const orc_lookup_ctx* lookups;  // NULL: no lookup argument. Otherwise the zs oracle ends with nc * (num_sldc + 1) lookup polynomials
  const gl_t* deltas;             // [nc][4] lookup challenges (A, B, alpha, delta) per challenge round
} orc_gate_ctx;
// A circuit as prove() / verify() need it beyond the FRI shape: permutation geometry, gate table, lookup tables
typedef struct {
  uint32_t num_routed, degree;
  const orc_gate* gates;
  uint32_t n_gates, num_selectors;
  const orc_lookup* luts;
  uint32_t n_luts;
} orc_circuit;
void orc_quotient_polys(const gl_t* wires_coeffs, const gl_t* sigma_coeffs, const gl_t* zs_coeffs, unsigned log_n,
                        unsigned num_routed, unsigned degree, const gl_t* betas, const gl_t* gammas, const gl_t* alphas,
                        unsigned nc, const orc_gate_ctx* G, gl_t* out);

// ---- challenger ---------------------------------------------------------------------------
void orc_ch_init(orc_challenger* c, int variant) { memset(c, 0, sizeof *c); c->variant = variant; }
static void ch_duplex(orc_challenger* c) {
  memcpy(c->state, c->in, c->n_in * sizeof(gl_t));
  c->n_in = 0;
  orc_perm(c->variant, c->state);
  memcpy(c->out, c->state, 8 * sizeof(gl_t));
  c->n_out = 8;
}
void orc_ch_observe(orc_challenger* c, const gl_t* e, size_t n) {
  for (size_t i = 0; i < n; i++) {
    c->n_out = 0;
    c->in[c->n_in++] = e[i];
    if (c->n_in == 8) ch_duplex(c);
  }
}
gl_t orc_ch_get(orc_challenger* c) {
  if (c->n_in || !c->n_out) ch_duplex(c);
  return c->out[--c->n_out];
}
gl2_t orc_ch_get_ext(orc_challenger* c) {
  gl2_t r;
  r.c[0] = orc_ch_get(c);
  r.c[1] = orc_ch_get(c);
  return r;
}

// ---- helpers ------------------------------------------------------------------------------
// fft of ext-valued vectors = component-wise base fft (roots are in the base field)
static void ext_coset_fft(gl2_t* a, unsigned log_n, gl_t shift) {
  size_t n = (size_t)1 << log_n;
  gl_t* t = malloc(n * sizeof(gl_t));
  for (int c = 0; c < 2; c++) {
    for (size_t i = 0; i < n; i++) t[i] = a[i].c[c];
    orc_coset_fft(t, log_n, shift);
    for (size_t i = 0; i < n; i++) a[i].c[c] = t[i];
  }
  free(t);
}
static gl2_t eval_base_poly_ext(const gl_t* coeffs, size_t n, gl2_t z) {
  gl2_t acc = gl2_from(0);
  for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, z), gl2_from(coeffs[i]));
  return acc;
}
static gl2_t eval_ext_poly(const gl2_t* coeffs, size_t n, gl2_t z) {
  gl2_t acc = gl2_from(0);
  for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, z), coeffs[i]);
  return acc;
}

// Flat proof sizes (u64 words). Layout:
//   caps[n_layers][2^cap][4] | queries[num_queries]{ per oracle: leaf[w] sib[(lg-cap)][4] ;
//   per layer i: evals[2^arity][2] sib[(lg - sum_{j<=i} arity_j - cap)][4] } | final[len][2] | pow
size_t orc_fri_proof_words(const orc_fri_params* P) {
  unsigned lg = P->log_n + P->rate_bits;
  size_t capw = ((size_t)4) << P->cap_height;
  size_t q = 0;
  for (uint32_t o = 0; o < P->n_oracles; o++) q += P->oracle_w[o] + 4 * (lg - P->cap_height);
  unsigned cur = lg;
  unsigned deg = P->log_n;
  for (uint32_t i = 0; i < P->n_layers; i++) {
    cur -= P->arity_bits[i];
    deg -= P->arity_bits[i];
    q += (2u << P->arity_bits[i]) + 4 * (cur - P->cap_height);
  }
  return P->n_layers * capw + P->num_queries * q + (2ull << deg) + 1;
}
static size_t n_lookup(const orc_fri_params* P) { return (size_t)P->zs_count * P->num_lookup_polys; }
// config.num_challenges: one Z polynomial per challenge round, so zs_count when the oracles are plonky2's; the
// PCS-only skeleton (any zs_count, no permutation argument) draws as for two rounds
static unsigned num_challenges(const orc_fri_params* P) { return P->zs_count >= 1 && P->zs_count <= 2 ? P->zs_count : 2; }
size_t orc_n_openings(const orc_fri_params* P) {
  size_t t = P->zs_count + n_lookup(P);
  for (uint32_t o = 0; o < P->n_oracles; o++) t += P->oracle_w[o];
  return t;
}
// FRI batch order ([dep] plonk/circuit_data.rs fri_all_polys / fri_next_batch_polys): at zeta every polynomial in
// oracle order EXCEPT the lookup polynomials, which close the batch (after the quotient chunks); at g*zeta the Z
// polynomials, then the lookup polynomials. The flat `openings` array and the transcript use the same order
// (OpeningSet::to_fri_openings).
static size_t batch_len(const orc_fri_params* P, int batch) {
  if (batch) return P->zs_count + n_lookup(P);
  size_t t = 0;
  for (uint32_t o = 0; o < P->n_oracles; o++) t += P->oracle_w[o];
  return t;
}
static void batch_poly(const orc_fri_params* P, int batch, size_t j, uint32_t* o, uint32_t* p) {
  const uint32_t zo = P->zs_oracle, L = (uint32_t)n_lookup(P), wz = P->oracle_w[zo] - L;
  if (batch) {
    *o = zo;
    *p = j < P->zs_count ? (uint32_t)j : wz + (uint32_t)(j - P->zs_count);
    return;
  }
  for (uint32_t oi = 0; oi < P->n_oracles; oi++) {
    uint32_t w = oi == zo ? wz : P->oracle_w[oi];
    if (j < w) { *o = oi; *p = (uint32_t)j; return; }
    j -= w;
  }
  *o = zo;
  *p = wz + (uint32_t)j;
}

// prove_openings + fri_proof. coeffs[o] = [w_o][n] base coefficients; leaves[o] = [8n][w_o];
// levels[o] = Merkle levels. Challenger state continues the caller's transcript.
// The proof-of-work witness is the smallest one, or the one armed with orc_set_pow_witness (the reference's search
// is a non-deterministic find_any; parity is defined given the witness).
// (thread-local: the oracle proves on several threads in bench.py's CPU legs, and an armed witness must reach the proof of the thread
// that armed it, not whichever proof comes first; test-only -- tests/test_reference_vectors.py)
static _Thread_local gl_t g_pow_override;
static _Thread_local int g_pow_override_armed;
// the NEXT proof made ON THIS THREAD (one proof, then the search is back) takes `witness` as its proof-of-work witness
void orc_set_pow_witness(gl_t witness) { g_pow_override = witness; g_pow_override_armed = 1; }
void orc_fri_prove(const orc_fri_params* P, gl_t* const* coeffs, gl_t* const* leaves, gl_t* const* levels,
                   gl2_t zeta, orc_challenger* ch, gl_t* proof) {
  unsigned k = P->log_n, lg = k + P->rate_bits;
  size_t n = (size_t)1 << k, N = (size_t)1 << lg;
  gl2_t alpha = orc_ch_get_ext(ch);
  gl2_t g_zeta = gl2_scale(zeta, gl_root_of_unity(k));

  gl2_t* final_poly = calloc(N, sizeof(gl2_t));  // lde: zero padded to 8n
  gl2_t* comp = malloc(n * sizeof(gl2_t));
  for (int batch = 0; batch < 2; batch++) {
    gl2_t point = batch == 0 ? zeta : g_zeta;
    // reduce_polys_base: sum_j alpha^j f_j, j in batch order
    for (size_t i = 0; i < n; i++) comp[i] = gl2_from(0);
    gl2_t apow = gl2_from(1);
    size_t count = batch_len(P, batch);
    for (size_t j = 0; j < count; j++) {
      uint32_t o, p;
      batch_poly(P, batch, j, &o, &p);
      const gl_t* f = coeffs[o] + (size_t)p * n;
      for (size_t i = 0; i < n; i++) comp[i] = gl2_add(comp[i], gl2_scale(apow, f[i]));
      apow = gl2_mul(apow, alpha);
    }
    // divide_by_linear(point): q_i = sum_{k>i} c_k point^(k-i-1); pad back with a zero
    // alpha.shift_poly(final): final *= alpha^count ; final += quotient
    gl2_t sh = gl2_pow(alpha, count);
    for (size_t i = 0; i < n; i++) final_poly[i] = gl2_mul(final_poly[i], sh);
    gl2_t acc = gl2_from(0);
    for (size_t i = n; i-- > 1;) {
      acc = gl2_add(gl2_mul(acc, point), comp[i]);
      final_poly[i - 1] = gl2_add(final_poly[i - 1], acc);
    }
  }
  free(comp);

  // lde_final_values = lde_final_poly.coset_fft(g)
  gl2_t* cf = final_poly;  // coefficients, length m (only the first m>>rate are non-zero)
  size_t m = N;
  gl2_t* values = malloc(N * sizeof(gl2_t));
  memcpy(values, cf, N * sizeof(gl2_t));
  ext_coset_fft(values, lg, GL_MULT_GEN);

  // ---- commit phase
  size_t capw = ((size_t)4) << P->cap_height;
  gl_t* out_caps = proof;
  gl_t** layer_levels = calloc(P->n_layers, sizeof(gl_t*));
  gl_t** layer_leaves = calloc(P->n_layers, sizeof(gl_t*));
  unsigned cur_lg = lg;
  gl_t shift = GL_MULT_GEN;
  for (uint32_t li = 0; li < P->n_layers; li++) {
    unsigned ab = P->arity_bits[li];
    size_t arity = (size_t)1 << ab;
    // reverse_index_bits_in_place(values); chunks(arity) flattened are the leaves
    gl_t* lv = malloc(m * 2 * sizeof(gl_t));
    for (size_t i = 0; i < m; i++) {
      size_t j = orc_bitrev(i, cur_lg);
      lv[2 * j] = values[i].c[0];
      lv[2 * j + 1] = values[i].c[1];
    }
    unsigned log_leaves = cur_lg - ab;
    gl_t* lvl = malloc(orc_merkle_levels_len(log_leaves, P->cap_height) * sizeof(gl_t));
    orc_merkle_build(P->variant, lv, 2 * arity, log_leaves, P->cap_height, lvl);
    const gl_t* cap = orc_merkle_cap_ptr(lvl, log_leaves, P->cap_height);
    memcpy(out_caps + li * capw, cap, capw * sizeof(gl_t));
    orc_ch_observe(ch, cap, capw);
    layer_levels[li] = lvl;
    layer_leaves[li] = lv;
    gl2_t beta = orc_ch_get_ext(ch);
    // coeffs.chunks_exact(arity).map(reduce_with_powers(beta))
    size_t m2 = m >> ab;
    for (size_t i = 0; i < m2; i++) {
      gl2_t acc = gl2_from(0);
      for (size_t j = arity; j-- > 0;) acc = gl2_add(gl2_mul(acc, beta), cf[i * arity + j]);
      cf[i] = acc;
    }
    m = m2;
    cur_lg -= ab;
    shift = gl_pow(shift, arity);
    memcpy(values, cf, m * sizeof(gl2_t));
    ext_coset_fft(values, cur_lg, shift);
  }
  size_t final_len = m >> P->rate_bits;
  size_t qwords = 0;
  {
    orc_fri_params Q = *P;
    qwords = (orc_fri_proof_words(&Q) - P->n_layers * capw - 2 * final_len - 1) / P->num_queries;
  }
  gl_t* out_q = proof + P->n_layers * capw;
  gl_t* out_final = out_q + P->num_queries * qwords;
  for (size_t i = 0; i < final_len; i++) { out_final[2 * i] = cf[i].c[0]; out_final[2 * i + 1] = cf[i].c[1]; }
  orc_ch_observe(ch, out_final, 2 * final_len);

  // ---- proof of work: smallest witness whose response has >= pow_bits leading zeros -- unless orc_set_pow_witness() armed another
  // one: the reference searches with rayon's find_any, so ITS proof holds any valid witness, and everything after the PoW (the
  // query indices) follows from it; a proof of the reference is reproduced bit for bit given its witness (tests/test_reference_vectors.py)
  if (g_pow_override_armed) {
    gl_t wit = g_pow_override;
    g_pow_override_armed = 0;
    out_final[2 * final_len] = wit;
    orc_ch_observe(ch, &wit, 1);
    gl_t resp = orc_ch_get(ch);
    if (P->pow_bits && (resp >> (64 - P->pow_bits)) != 0) { fprintf(stderr, "oracle: the given pow witness does not satisfy the transcript\n"); abort(); }
  } else {
    gl_t st[12];
    memcpy(st, ch->state, sizeof st);
    memcpy(st, ch->in, ch->n_in * sizeof(gl_t));
    unsigned pos = ch->n_in;
    // plonky2 searches with rayon (find_any); here chunks of candidates are tried in parallel and the
    // smallest hit of the first chunk that has one is kept (= the smallest witness overall)
    gl_t wit = 0;
    for (gl_t base = 0;; base += 4096) {
      gl_t best = ~(gl_t)0;
#pragma omp parallel for schedule(static) reduction(min : best)
      for (long k = 0; k < 4096; k++) {
        gl_t t[12];
        memcpy(t, st, sizeof t);
        t[pos] = base + (gl_t)k;
        orc_perm(P->variant, t);
        if ((P->pow_bits == 0 || (t[7] >> (64 - P->pow_bits)) == 0) && base + (gl_t)k < best) best = base + (gl_t)k;
      }
      if (best != ~(gl_t)0) { wit = best; break; }
    }
    out_final[2 * final_len] = wit;
    orc_ch_observe(ch, &wit, 1);
    gl_t resp = orc_ch_get(ch);
    if (P->pow_bits && (resp >> (64 - P->pow_bits)) != 0) { fprintf(stderr, "oracle: pow mismatch\n"); abort(); }
  }
  // ---- query rounds
  for (uint32_t q = 0; q < P->num_queries; q++) {
    gl_t* o = out_q + q * qwords;
    size_t x = orc_ch_get(ch) % N;
    for (uint32_t oi = 0; oi < P->n_oracles; oi++) {
      uint32_t wo = P->oracle_w[oi];
      memcpy(o, leaves[oi] + x * wo, wo * sizeof(gl_t));
      o += wo;
      orc_merkle_prove(levels[oi], lg, P->cap_height, x, o);
      o += 4 * (lg - P->cap_height);
    }
    unsigned clg = lg;
    for (uint32_t li = 0; li < P->n_layers; li++) {
      unsigned ab = P->arity_bits[li];
      size_t arity = (size_t)1 << ab;
      x >>= ab;
      clg -= ab;
      memcpy(o, layer_leaves[li] + x * 2 * arity, 2 * arity * sizeof(gl_t));
      o += 2 * arity;
      orc_merkle_prove(layer_levels[li], clg, P->cap_height, x, o);
      o += 4 * (clg - P->cap_height);
    }
  }
  for (uint32_t li = 0; li < P->n_layers; li++) { free(layer_levels[li]); free(layer_leaves[li]); }
  free(layer_levels); free(layer_leaves); free(values); free(final_poly);
}

// The PCS skeleton of plonky2's prove(): commitments, Fiat-Shamir, openings, FRI.
// values[o] = [w_o][n] evaluations over the subgroup (natural order). Oracle 0 is the
// preprocessed constants_sigmas oracle (its cap is inside circuit_digest, never observed);
// oracle 1 wires, 2 zs_partial_products (+ lookup polynomials), 3 quotient chunks.
// Outputs: caps[n_oracles][2^cap][4], openings[n_openings][2] in FRI batch order, proof (flat, see above).
// num_routed > 0: oracle 2 (Z / partial products) is not taken from values[2] but computed from the
// wires (values[1]), the sigma values (last num_routed polynomials of values[0]) and the betas /
// gammas drawn after the wires cap, as prove() does; needs oracle_w[2] = zs_count * (num_routed/degree + num_lookup_polys).
// quotient != 0 (needs num_routed > 0): oracle 3 is not taken from values[3] either but computed as
// compute_quotient_polys does. chal, if not NULL, receives betas[2], gammas[2], alphas[2], zeta[2], deltas[8].
void orc_lookup_polys(const gl_t* wires, unsigned log_n, const gl_t deltas[4], const orc_lookup_ctx* L, gl_t* out);
static void pcs_prove_impl(const orc_fri_params* P, const gl_t* const* values, const gl_t circuit_digest[4],
                           const gl_t pi_hash[4], unsigned quotient, const orc_circuit* CK, gl_t* chal,
                           gl_t* caps, gl_t* openings, gl_t* proof) {
  unsigned k = P->log_n, lg = k + P->rate_bits;
  size_t n = (size_t)1 << k, N = (size_t)1 << lg;
  size_t capw = ((size_t)4) << P->cap_height;
  const unsigned num_routed = CK ? CK->num_routed : 0, degree = CK ? CK->degree : 8, nc = P->zs_count, nch = num_challenges(P);
  const unsigned n_gates = CK ? CK->n_gates : 0;
  orc_lookup_ctx LU = {0};
  const int has_lookup = CK && CK->n_luts && P->num_lookup_polys;
  if (has_lookup) { LU.luts = CK->luts; LU.n_luts = CK->n_luts; orc_lookup_shape(&LU, num_routed, degree); }
  const unsigned num_lookup_selectors = has_lookup ? ORC_LOOKUP_SELECTORS + CK->n_luts : 0;
  gl_t* coeffs[8]; gl_t* leaves[8]; gl_t* levels[8];
  const int timing = getenv("ORC_TIMING") != NULL;  // stage split on stderr (measurement aid for the cpu_baseline leg)
  double t_mark = omp_get_wtime(), t_stage[8] = {0};
#define T_ADD(i) do { double t_now = omp_get_wtime(); t_stage[i] += t_now - t_mark; t_mark = t_now; } while (0)
  orc_challenger ch;
  orc_ch_init(&ch, P->variant);
  orc_ch_observe(&ch, circuit_digest, 4);
  orc_ch_observe(&ch, pi_hash, 4);
  // betas, gammas, then (with lookups) 2 * nc more: deltas = betas ++ gammas ++ extra, 4 per challenge round
  gl_t bg[8] = {0}, al[2] = {0, 0};
  gl_t* zs_vals = NULL;
  for (uint32_t o = 0; o < P->n_oracles; o++) {
    size_t w = P->oracle_w[o];
    const gl_t* src = values[o];
    if (o == 2 && num_routed) {
      zs_vals = malloc(w * n * sizeof(gl_t));
      orc_partial_products_and_zs(values[1], values[0] + (size_t)(P->oracle_w[0] - num_routed) * n, k, num_routed, degree,
                                  bg, bg + nc, nc, zs_vals);
      if (has_lookup)  // compute_all_lookup_polys: per challenge round RE + the partial Sum/LDC polynomials
        for (unsigned c = 0; c < nc; c++)
          orc_lookup_polys(values[1], k, bg + 4 * c, &LU, zs_vals + ((size_t)nc * (num_routed / degree) + (size_t)c * P->num_lookup_polys) * n);
      src = zs_vals;
      T_ADD(0);  // Z / partial products / lookup polynomials
    }
    coeffs[o] = malloc(w * n * sizeof(gl_t));
    if (o == 3 && quotient && num_routed) {
      // PolynomialBatch::from_coeffs: the quotient chunks are produced in coefficient form
      orc_gate_ctx G = {CK->gates, n_gates, CK->num_selectors, P->oracle_w[0] - num_routed, P->oracle_w[1], coeffs[0], pi_hash,
                        has_lookup ? &LU : NULL, bg};
      (void)num_lookup_selectors;
      orc_quotient_polys(coeffs[1], coeffs[0] + (size_t)(P->oracle_w[0] - num_routed) * n, coeffs[2], k, num_routed, degree,
                         bg, bg + nc, al, nc, (n_gates || has_lookup) ? &G : NULL, coeffs[o]);
      T_ADD(1);  // quotient polynomials
    } else {
      memcpy(coeffs[o], src, w * n * sizeof(gl_t));
#pragma omp parallel for schedule(dynamic)
      for (size_t p = 0; p < w; p++) orc_fft(coeffs[o] + p * n, k, 1);
      T_ADD(2);  // iFFTs
    }
    leaves[o] = malloc(w * N * sizeof(gl_t));
    orc_lde_leaves(coeffs[o], k, w, P->rate_bits, leaves[o]);
    T_ADD(3);  // LDE
    levels[o] = malloc(orc_merkle_levels_len(lg, P->cap_height) * sizeof(gl_t));
    orc_merkle_build(P->variant, leaves[o], w, lg, P->cap_height, levels[o]);
    T_ADD(4);  // Merkle trees
    memcpy(caps + o * capw, orc_merkle_cap_ptr(levels[o], lg, P->cap_height), capw * sizeof(gl_t));
    if (o == 0) continue;
    orc_ch_observe(&ch, caps + o * capw, capw);
    // wires cap -> num_challenges betas, then as many gammas (+ 2 per challenge more with lookups); zs cap -> alphas
    if (o == 1) for (uint32_t i = 0; i < (has_lookup ? 4 : 2) * nch; i++) bg[i] = orc_ch_get(&ch);
    else if (o == 2) for (uint32_t i = 0; i < nch; i++) al[i] = orc_ch_get(&ch);
  }
  free(zs_vals);
  gl2_t zeta = orc_ch_get_ext(&ch);
  if (chal) {  // betas at [0..2), gammas at [2..4), alphas at [4..6), zeta, deltas[8] whatever num_challenges is
    memset(chal, 0, (has_lookup ? 16 : 8) * sizeof(gl_t));
    for (uint32_t i = 0; i < nch; i++) { chal[i] = bg[i]; chal[2 + i] = bg[nch + i]; chal[4 + i] = al[i]; }
    chal[6] = zeta.c[0]; chal[7] = zeta.c[1];
    if (has_lookup) memcpy(chal + 8, bg, 8 * sizeof(gl_t));
  }
  gl2_t g_zeta = gl2_scale(zeta, gl_root_of_unity(k));
  size_t oi = 0;
  for (int batch = 0; batch < 2; batch++)
    for (size_t j = 0; j < batch_len(P, batch); j++, oi++) {
      uint32_t o, p;
      batch_poly(P, batch, j, &o, &p);
      gl2_t v = eval_base_poly_ext(coeffs[o] + (size_t)p * n, n, batch ? g_zeta : zeta);
      openings[2 * oi] = v.c[0]; openings[2 * oi + 1] = v.c[1];
    }
  orc_ch_observe(&ch, openings, 2 * oi);
  T_ADD(5);  // openings
  orc_fri_prove(P, coeffs, leaves, levels, zeta, &ch, proof);
  T_ADD(6);  // FRI
  if (timing)
    fprintf(stderr, "oracle prove 2^%u: zs %.3f quotient %.3f ifft %.3f lde %.3f merkle %.3f openings %.3f fri %.3f s (%d threads)\n", k, t_stage[0],
            t_stage[1], t_stage[2], t_stage[3], t_stage[4], t_stage[5], t_stage[6], omp_get_max_threads());
#undef T_ADD
  for (uint32_t o = 0; o < P->n_oracles; o++) { free(coeffs[o]); free(leaves[o]); free(levels[o]); }
}

void orc_pcs_prove(const orc_fri_params* P, const gl_t* const* values, const gl_t circuit_digest[4],
                   const gl_t pi_hash[4], unsigned num_routed, unsigned degree, unsigned quotient, gl_t* bgao,
                   gl_t* caps, gl_t* openings, gl_t* proof) {
  orc_circuit CK = {num_routed, degree, NULL, 0, 0, NULL, 0};
  pcs_prove_impl(P, values, circuit_digest, pi_hash, quotient, &CK, bgao, caps, openings, proof);
}
// prove() of a circuit with gates: as orc_pcs_prove(quotient = 1) with the gate constraints as further
// terms of the vanishing polynomial. The constants are the first oracle_w[0] - num_routed polynomials of
// values[0], selectors first.
void orc_pcs_prove_gates(const orc_fri_params* P, const gl_t* const* values, const gl_t circuit_digest[4],
                         const gl_t pi_hash[4], unsigned num_routed, unsigned degree, const orc_gate* gates,
                         unsigned n_gates, unsigned num_selectors, gl_t* bgao, gl_t* caps, gl_t* openings, gl_t* proof) {
  orc_circuit CK = {num_routed, degree, gates, n_gates, num_selectors, NULL, 0};
  pcs_prove_impl(P, values, circuit_digest, pi_hash, 1, &CK, bgao, caps, openings, proof);
}
// ... and with lookup tables: constants = selectors, 4 + n_luts lookup selectors, gate constants; the zs oracle
// ends with the lookup polynomials; chal receives 16 words (see pcs_prove_impl)
void orc_prove_circuit(const orc_fri_params* P, const gl_t* const* values, const gl_t circuit_digest[4], const gl_t pi_hash[4],
                       const orc_circuit* CK, gl_t* chal, gl_t* caps, gl_t* openings, gl_t* proof) {
  pcs_prove_impl(P, values, circuit_digest, pi_hash, 1, CK, chal, caps, openings, proof);
}

// ---- lookup argument: the RE / Sum / LDC polynomials ---------------------------------------------
// [dep] plonky2 plonk/prover.rs compute_lookup_polys (one challenge round): out[0] = RE, out[1 + s] = partial
// Sum/LDC polynomial s; all zero outside the lookup rows. The table rows run downwards from first_lut_row, the
// LookupGate rows below them; every recurrence steps from row + 1 to row (the gate rows are "upside down" so
// that no constraint needs the next row's wires). wires = [>= routed][n] subgroup values.
void orc_lookup_polys(const gl_t* wires, unsigned log_n, const gl_t deltas[4], const orc_lookup_ctx* L, gl_t* out) {
  size_t n = (size_t)1 << log_n;
  unsigned ns = L->num_sldc;
  memset(out, 0, (size_t)(ns + 1) * n * sizeof(gl_t));
  for (unsigned r = 0; r < L->n_luts; r++) {
    const orc_lookup* lu = &L->luts[r];
    for (size_t row = lu->first_lut_row + 1; row-- > lu->last_lut_row;) {
      gl_t re = out[row + 1];  // RE of the row above (0 at the Noop row that follows the table)
      for (unsigned s = 0; s < L->num_lut_slots; s++) {
        gl_t inp = wires[(size_t)(3 * s) * n + row], outp = wires[(size_t)(3 * s + 1) * n + row];
        re = gl_add(gl_mul(re, deltas[3]), gl_add(inp, gl_mul(deltas[1], outp)));
      }
      out[row] = re;
      for (unsigned p = 0; p < ns; p++) {
        gl_t acc = p == 0 ? out[(size_t)ns * n + row + 1] : out[(size_t)p * n + row];
        for (unsigned s = p * L->lut_degree; s < (p + 1) * L->lut_degree && s < L->num_lut_slots; s++) {
          gl_t inp = wires[(size_t)(3 * s) * n + row], outp = wires[(size_t)(3 * s + 1) * n + row], mul = wires[(size_t)(3 * s + 2) * n + row];
          acc = gl_add(acc, gl_mul(mul, gl_inv(gl_sub(deltas[2], gl_add(inp, gl_mul(deltas[0], outp))))));
        }
        out[(size_t)(p + 1) * n + row] = acc;
      }
    }
    for (size_t row = lu->last_lut_row; row-- > lu->last_lu_row;) {
      for (unsigned p = 0; p < ns; p++) {
        gl_t acc = p == 0 ? out[(size_t)ns * n + row + 1] : out[(size_t)p * n + row];
        for (unsigned s = p * L->lu_degree; s < (p + 1) * L->lu_degree && s < L->num_lu_slots; s++) {
          gl_t inp = wires[(size_t)(2 * s) * n + row], outp = wires[(size_t)(2 * s + 1) * n + row];
          acc = gl_sub(acc, gl_inv(gl_sub(deltas[2], gl_add(inp, gl_mul(deltas[0], outp)))));
        }
        out[(size_t)(p + 1) * n + row] = acc;
      }
    }
  }
}

// ---- verifier -----------------------------------------------------------------------------
// Returns 0 when the proof verifies, otherwise a positive code naming the failed check.
int orc_pcs_verify(const orc_fri_params* P, const gl_t circuit_digest[4], const gl_t pi_hash[4],
                   const gl_t* caps, const gl_t* openings, const gl_t* proof) {
  unsigned k = P->log_n, lg = k + P->rate_bits;
  size_t N = (size_t)1 << lg;
  size_t capw = ((size_t)4) << P->cap_height;
  size_t n_open = orc_n_openings(P), n_zeta = batch_len(P, 0);
  orc_challenger ch;
  orc_ch_init(&ch, P->variant);
  orc_ch_observe(&ch, circuit_digest, 4);
  orc_ch_observe(&ch, pi_hash, 4);
  for (uint32_t o = 1; o < P->n_oracles; o++) {
    orc_ch_observe(&ch, caps + o * capw, capw);
    if (o == 1) for (uint32_t i = 0; i < (P->num_lookup_polys ? 4 : 2) * num_challenges(P); i++) (void)orc_ch_get(&ch);
    else if (o == 2) for (uint32_t i = 0; i < num_challenges(P); i++) (void)orc_ch_get(&ch);
  }
  gl2_t zeta = orc_ch_get_ext(&ch);
  gl2_t g_zeta = gl2_scale(zeta, gl_root_of_unity(k));
  orc_ch_observe(&ch, openings, 2 * n_open);
  gl2_t alpha = orc_ch_get_ext(&ch);
  gl2_t betas[8];
  const gl_t* pcaps = proof;
  for (uint32_t li = 0; li < P->n_layers; li++) {
    orc_ch_observe(&ch, pcaps + li * capw, capw);
    betas[li] = orc_ch_get_ext(&ch);
  }
  unsigned deg = k;
  for (uint32_t li = 0; li < P->n_layers; li++) deg -= P->arity_bits[li];
  size_t final_len = (size_t)1 << deg;
  size_t total = orc_fri_proof_words(P);
  size_t qwords = (total - P->n_layers * capw - 2 * final_len - 1) / P->num_queries;
  const gl_t* pq = proof + P->n_layers * capw;
  const gl_t* pfinal = pq + P->num_queries * qwords;
  orc_ch_observe(&ch, pfinal, 2 * final_len);
  gl_t wit = pfinal[2 * final_len];
  orc_ch_observe(&ch, &wit, 1);
  gl_t resp = orc_ch_get(&ch);
  if (P->pow_bits && (resp >> (64 - P->pow_bits)) != 0) return 1;
  // PrecomputedReducedOpenings: sum_j alpha^j v_j per batch
  gl2_t red[2];
  {
    gl2_t acc = gl2_from(0);
    for (size_t i = n_zeta; i-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), (gl2_t){{openings[2 * i], openings[2 * i + 1]}});
    red[0] = acc;
    acc = gl2_from(0);
    for (size_t i = n_open; i-- > n_zeta;) acc = gl2_add(gl2_mul(acc, alpha), (gl2_t){{openings[2 * i], openings[2 * i + 1]}});
    red[1] = acc;
  }
  gl2_t* fin = malloc(final_len * sizeof(gl2_t));
  for (size_t i = 0; i < final_len; i++) fin[i] = (gl2_t){{pfinal[2 * i], pfinal[2 * i + 1]}};
  int rc = 0;
  for (uint32_t q = 0; q < P->num_queries && !rc; q++) {
    const gl_t* o = pq + q * qwords;
    size_t x = orc_ch_get(&ch) % N;
    // initial trees
    const gl_t* leaf[8];
    for (uint32_t oi = 0; oi < P->n_oracles; oi++) {
      uint32_t wo = P->oracle_w[oi];
      leaf[oi] = o;
      if (!orc_merkle_verify(P->variant, o, wo, x, o + wo, lg - P->cap_height, caps + oi * capw)) { rc = 2; break; }
      o += wo + 4 * (lg - P->cap_height);
    }
    if (rc) break;
    gl_t sx = gl_mul(GL_MULT_GEN, gl_pow(gl_root_of_unity(lg), orc_bitrev(x, lg)));
    // fri_combine_initial
    gl2_t sum = gl2_from(0);
    for (int batch = 0; batch < 2; batch++) {
      gl2_t acc = gl2_from(0);
      size_t count = batch_len(P, batch);
      // alpha.reduce(evals): sum_j alpha^j e_j  (Horner from the back), j in FRI batch order
      for (size_t j = count; j-- > 0;) {
        uint32_t oi, p;
        batch_poly(P, batch, j, &oi, &p);
        acc = gl2_add(gl2_mul(acc, alpha), gl2_from(leaf[oi][p]));
      }
      gl2_t num = gl2_sub(acc, red[batch]);
      gl2_t den = gl2_sub(gl2_from(sx), batch == 0 ? zeta : g_zeta);
      sum = gl2_mul(sum, gl2_pow(alpha, count));
      sum = gl2_add(sum, gl2_mul(num, gl2_inv(den)));
    }
    gl2_t old_eval = sum;
    unsigned clg = lg;
    for (uint32_t li = 0; li < P->n_layers; li++) {
      unsigned ab = P->arity_bits[li];
      size_t arity = (size_t)1 << ab;
      size_t coset = x >> ab, within = x & (arity - 1);
      const gl_t* ev = o;
      if (ev[2 * within] != old_eval.c[0] || ev[2 * within + 1] != old_eval.c[1]) { rc = 3; break; }
      // compute_evaluation: interpolate {(coset_start*g^i, evals_rev[i])} at beta (Lagrange)
      gl_t g = gl_root_of_unity(ab);
      size_t rev_within = orc_bitrev(within, ab);
      gl_t start = gl_mul(sx, gl_pow(g, arity - rev_within));
      gl2_t res = gl2_from(0);
      for (size_t i = 0; i < arity; i++) {
        size_t src = orc_bitrev(i, ab);
        gl2_t yi = (gl2_t){{ev[2 * src], ev[2 * src + 1]}};
        gl_t xi = gl_mul(start, gl_pow(g, i));
        gl2_t numr = gl2_from(1);
        gl_t den = 1;
        for (size_t j = 0; j < arity; j++) {
          if (j == i) continue;
          gl_t xj = gl_mul(start, gl_pow(g, j));
          numr = gl2_mul(numr, gl2_sub(betas[li], gl2_from(xj)));
          den = gl_mul(den, gl_sub(xi, xj));
        }
        res = gl2_add(res, gl2_mul(yi, gl2_scale(numr, gl_inv(den))));
      }
      old_eval = res;
      clg -= ab;
      if (!orc_merkle_verify(P->variant, ev, 2 * arity, coset, ev + 2 * arity, clg - P->cap_height, pcaps + li * capw)) { rc = 4; break; }
      o += 2 * arity + 4 * (clg - P->cap_height);
      sx = gl_pow(sx, arity);
      x = coset;
    }
    if (rc) break;
    gl2_t fe = eval_ext_poly(fin, final_len, gl2_from(sx));
    if (!gl2_eq(fe, old_eval)) rc = 5;
  }
  free(fin);
  return rc;
}

// fri/reduction_strategies.rs ConstantArityBits(arity_bits, final_poly_bits)
uint32_t orc_reduction_arity_bits(uint32_t degree_bits, uint32_t rate_bits, uint32_t cap_height,
                                  uint32_t arity_bits, uint32_t final_poly_bits, uint32_t* out) {
  uint32_t n = 0;
  while (degree_bits > final_poly_bits && degree_bits + rate_bits - arity_bits >= cap_height) {
    out[n++] = arity_bits;
    degree_bits -= arity_bits;
  }
  return n;
}
// value-domain fold of one FRI layer (equivalent to the coefficient fold + re-FFT the reference
// performs): in = bit-reversed evaluations [m][2] on shift*<w_m>, out = natural-order
// evaluations [m>>arity_bits][2] on shift^arity * <w_{m/arity}>.
void orc_fri_fold_values(const gl_t* in_bitrev, unsigned log_m, unsigned arity_bits, const gl_t beta_[2],
                         gl_t shift, gl_t* out) {
  size_t m = (size_t)1 << log_m;
  gl2_t* v = malloc(m * sizeof(gl2_t));
  for (size_t i = 0; i < m; i++) { size_t j = orc_bitrev(i, log_m); v[i] = (gl2_t){{in_bitrev[2 * j], in_bitrev[2 * j + 1]}}; }
  // to coefficients: coset ifft per component
  gl_t* t = malloc(m * sizeof(gl_t));
  gl_t si = gl_inv(shift);
  for (int c = 0; c < 2; c++) {
    for (size_t i = 0; i < m; i++) t[i] = v[i].c[c];
    orc_fft(t, log_m, 1);
    gl_t s = 1;
    for (size_t i = 0; i < m; i++) { v[i].c[c] = gl_mul(t[i], s); s = gl_mul(s, si); }
  }
  free(t);
  gl2_t beta = {{beta_[0], beta_[1]}};
  size_t arity = (size_t)1 << arity_bits, m2 = m >> arity_bits;
  for (size_t i = 0; i < m2; i++) {
    gl2_t acc = gl2_from(0);
    for (size_t j = arity; j-- > 0;) acc = gl2_add(gl2_mul(acc, beta), v[i * arity + j]);
    v[i] = acc;
  }
  ext_coset_fft(v, log_m - arity_bits, gl_pow(shift, arity));
  for (size_t i = 0; i < m2; i++) { out[2 * i] = v[i].c[0]; out[2 * i + 1] = v[i].c[1]; }
  free(v);
}

// ---- permutation argument: Z and partial products ---------------------------------------------
// [dep] plonky2 plonk/prover.rs all_wires_permutation_partial_products /
// wires_permutation_partial_products_and_zs, plonk/vanishing_poly / plonk_common.rs
// quotient_chunk_products + partial_products_and_z_gx, plonk/permutation_argument /
// field/src/cosets.rs get_unique_coset_shifts (k_i = g^i), and the Z-first ordering of prove():
//   zs_partial_products = [Z_0 .. Z_{nc-1}, pp(challenge 0)[0..num_prods), pp(challenge 1) ...].
// wires: [>= num_routed][n] subgroup values (row j = wire column j), sigmas: [num_routed][n] values
// of the sigma polynomials, degree = quotient_degree_factor (8), num_routed % degree == 0.
// out: [nc * (1 + num_prods)][n], num_prods = num_routed/degree - 1.
void orc_partial_products_and_zs(const gl_t* wires, const gl_t* sigmas, unsigned log_n, unsigned num_routed,
                                 unsigned degree, const gl_t* betas, const gl_t* gammas, unsigned nc, gl_t* out) {
  size_t n = (size_t)1 << log_n;
  unsigned chunks = num_routed / degree, num_prods = chunks - 1;
  gl_t* k_is = malloc(num_routed * sizeof(gl_t));
  k_is[0] = 1;
  for (unsigned j = 1; j < num_routed; j++) k_is[j] = gl_mul(k_is[j - 1], GL_MULT_GEN);
  gl_t w = gl_root_of_unity(log_n);
  gl_t* chunk_prod = malloc(n * chunks * sizeof(gl_t));
  for (unsigned c = 0; c < nc; c++) {
    gl_t beta = betas[c], gamma = gammas[c];
    gl_t x = 1;
    for (size_t i = 0; i < n; i++) {
      for (unsigned k = 0; k < chunks; k++) {
        gl_t num = 1, den = 1;
        for (unsigned j = k * degree; j < (k + 1) * degree; j++) {
          gl_t wv = wires[(size_t)j * n + i];
          num = gl_mul(num, gl_add(gl_add(wv, gl_mul(beta, gl_mul(k_is[j], x))), gamma));
          den = gl_mul(den, gl_add(gl_add(wv, gl_mul(beta, sigmas[(size_t)j * n + i])), gamma));
        }
        chunk_prod[i * chunks + k] = gl_mul(num, gl_inv(den));
      }
      x = gl_mul(x, w);
    }
    // running products; Z(x) at the front of the batch, partial products after all Z's
    gl_t z_x = 1;
    gl_t* zrow = out + (size_t)c * n;
    gl_t* pp = out + (size_t)nc * n + (size_t)c * num_prods * n;
    for (size_t i = 0; i < n; i++) {
      zrow[i] = z_x;
      gl_t acc = z_x;
      for (unsigned k = 0; k < chunks; k++) {
        acc = gl_mul(acc, chunk_prod[i * chunks + k]);
        if (k < num_prods) pp[(size_t)k * n + i] = acc;
      }
      z_x = acc;  // Z(g x)
    }
  }
  free(chunk_prod);
  free(k_is);
}

// ---- quotient polynomials for the gate-independent part of the vanishing polynomial ------------
// [dep] plonky2 plonk/prover.rs compute_quotient_polys + plonk/vanishing_poly.rs
// eval_vanishing_poly_base_batch / check_partial_products, with an empty gate set (a circuit whose
// only constraints are the copy constraints): terms = [L_0(x)(Z_c(x)-1) for c] ++ [partial-product
// checks of challenge 0, of challenge 1, ...]; per alpha: sum_k terms[k] alpha^k, divided by Z_H on
// the coset g<w_8n>, coset-iFFT, split into 8 chunks of n coefficients per challenge.
// wires/sigmas/zs given as coefficient vectors [.][n]; zs = [Z_0..Z_{nc-1}, pp(0), pp(1)..].
// out: [nc * 8][n] quotient chunk coefficients (the polynomials PolynomialBatch::from_coeffs commits).
void orc_quotient_perm(const gl_t* wires_coeffs, const gl_t* sigma_coeffs, const gl_t* zs_coeffs, unsigned log_n,
                       unsigned num_routed, unsigned degree, const gl_t* betas, const gl_t* gammas, const gl_t* alphas,
                       unsigned nc, gl_t* out) {
  orc_quotient_polys(wires_coeffs, sigma_coeffs, zs_coeffs, log_n, num_routed, degree, betas, gammas, alphas, nc, NULL, out);
}
// With G != NULL the gate constraints C_j(x) = sum_gates filter_g(x) c_{g,j}(x) follow the permutation
// terms in the alpha-reduction (vanishing_poly.rs eval_vanishing_poly_base_batch: z_1 terms, partial
// product terms, gate constraint terms); wires_coeffs must then hold all G->wires_w wire polynomials.
void orc_quotient_polys(const gl_t* wires_coeffs, const gl_t* sigma_coeffs, const gl_t* zs_coeffs, unsigned log_n,
                        unsigned num_routed, unsigned degree, const gl_t* betas, const gl_t* gammas, const gl_t* alphas,
                        unsigned nc, const orc_gate_ctx* G, gl_t* out) {
  const unsigned rate_bits = 3;
  size_t n = (size_t)1 << log_n, N = n << rate_bits;
  unsigned lg = log_n + rate_bits, chunks = num_routed / degree, num_prods = chunks - 1;
  const orc_lookup_ctx* LU = G ? G->lookups : NULL;
  const unsigned nlp = LU ? LU->num_sldc + 1 : 0;  // lookup polynomials per challenge, after the Z / partial products
  unsigned n_zs = nc * (chunks + nlp);
  const unsigned n_lookup_sel = LU ? ORC_LOOKUP_SELECTORS + LU->n_luts : 0;
  // natural-order LDE values on g<w_N>
  unsigned wires_w = G ? G->wires_w : num_routed;
  gl_t* W = malloc((size_t)wires_w * N * sizeof(gl_t));
  gl_t* S = malloc((size_t)num_routed * N * sizeof(gl_t));
  gl_t* C = NULL;
  gl_t* Z = malloc((size_t)n_zs * N * sizeof(gl_t));
  void orc_lde_values(const gl_t*, unsigned, size_t, unsigned, gl_t*);
  orc_lde_values(wires_coeffs, log_n, wires_w, rate_bits, W);
  orc_lde_values(sigma_coeffs, log_n, num_routed, rate_bits, S);
  if (G) {
    C = malloc((size_t)G->num_constants * N * sizeof(gl_t));
    orc_lde_values(G->const_coeffs, log_n, G->num_constants, rate_bits, C);
  }
  orc_lde_values(zs_coeffs, log_n, n_zs, rate_bits, Z);
  gl_t* k_is = malloc(num_routed * sizeof(gl_t));
  k_is[0] = 1;
  for (unsigned j = 1; j < num_routed; j++) k_is[j] = gl_mul(k_is[j - 1], GL_MULT_GEN);
  gl_t wN = gl_root_of_unity(lg), gn = gl_pow(GL_MULT_GEN, n), w8 = gl_root_of_unity(rate_bits);
  gl_t n_field = (gl_t)n % GL_P;
  gl_t* q = malloc((size_t)nc * N * sizeof(gl_t));
  // plonky2 evaluates the coset points in parallel batches (rayon); OpenMP over the points here
#pragma omp parallel
  {
  gl_t* terms = malloc((nc + (size_t)nc * chunks + (size_t)nc * 64 + ORC_MAX_GATE_CONSTRAINTS) * sizeof(gl_t));
  gl_t lz[16], lzn[16];
  gl_t* lc = G ? malloc((G->num_constants + 1) * sizeof(gl_t)) : NULL;
  gl_t* lw = G ? malloc(wires_w * sizeof(gl_t)) : NULL;
#pragma omp for schedule(static)
  for (size_t i = 0; i < N; i++) {
    gl_t x = gl_mul(GL_MULT_GEN, gl_pow(wN, i));
    gl_t zh = gl_sub(gl_mul(gn, gl_pow(w8, i % 8)), 1);  // x^n - 1
    gl_t l0 = gl_mul(zh, gl_inv(gl_mul(n_field, gl_sub(x, 1))));
    size_t inext = (i + 8) % N;
    size_t t = 0;
    for (unsigned c = 0; c < nc; c++) terms[t++] = gl_mul(l0, gl_sub(Z[(size_t)c * N + i], 1));
    for (unsigned c = 0; c < nc; c++) {
      const gl_t* pp = Z + ((size_t)nc + (size_t)c * num_prods) * N;
      for (unsigned k = 0; k < chunks; k++) {
        gl_t num = 1, den = 1;
        for (unsigned j = k * degree; j < (k + 1) * degree; j++) {
          gl_t wv = W[(size_t)j * N + i];
          num = gl_mul(num, gl_add(gl_add(wv, gl_mul(betas[c], gl_mul(k_is[j], x))), gammas[c]));
          den = gl_mul(den, gl_add(gl_add(wv, gl_mul(betas[c], S[(size_t)j * N + i])), gammas[c]));
        }
        gl_t prev = k == 0 ? Z[(size_t)c * N + i] : pp[(size_t)(k - 1) * N + i];
        gl_t next = k == chunks - 1 ? Z[(size_t)c * N + inext] : pp[(size_t)k * N + i];
        terms[t++] = gl_sub(gl_mul(prev, num), gl_mul(next, den));
      }
    }
    if (G) {
      for (unsigned j = 0; j < G->num_constants; j++) lc[j] = C[(size_t)j * N + i];
      for (unsigned j = 0; j < wires_w; j++) lw[j] = W[(size_t)j * N + i];
      if (LU)  // vanishing_all_lookup_terms: between the partial-product terms and the gate constraints
        for (unsigned c = 0; c < nc; c++) {
          const gl_t* lp = Z + ((size_t)nc * chunks + (size_t)c * nlp) * N;
          for (unsigned q = 0; q < nlp; q++) { lz[q] = lp[(size_t)q * N + i]; lzn[q] = lp[(size_t)q * N + inext]; }
          t += orc_lookup_terms_base(LU, lc + G->num_selectors, lw, lz, lzn, G->deltas + 4 * c, terms + t);
        }
      if (G->n_gates) t += orc_gates_eval_base(G->gates, G->n_gates, G->num_selectors, n_lookup_sel, lc, lw, G->pi_hash, terms + t);
    }
    gl_t zh_inv = gl_inv(zh);
    for (unsigned a = 0; a < nc; a++) {
      gl_t acc = 0;
      for (size_t k = t; k-- > 0;) acc = gl_add(gl_mul(acc, alphas[a]), terms[k]);
      q[(size_t)a * N + i] = gl_mul(acc, zh_inv);
    }
  }
  free(terms); free(lc); free(lw);
  }
  void orc_coset_ifft(gl_t*, unsigned, gl_t);
  for (unsigned a = 0; a < nc; a++) {
    orc_coset_ifft(q + (size_t)a * N, lg, GL_MULT_GEN);
    memcpy(out + (size_t)a * N, q + (size_t)a * N, N * sizeof(gl_t));  // 8 chunks of n, contiguous
  }
  free(q); free(k_is); free(W); free(S); free(Z); free(C);
}
// plonk/verifier.rs: vanishing(zeta) == Z_H(zeta) * sum_i zeta^(n i) t_i(zeta) for every challenge, from
// the opened values only (openings layout of orc_pcs_prove; num_constants = oracle_w[0] - num_routed).
// Returns 0 when the identity holds.
int orc_identity_check_circuit(const orc_fri_params* P, const orc_circuit* CK, const gl_t* openings, gl2_t zeta, const gl_t* betas,
                               const gl_t* gammas, const gl_t* alphas, const gl_t* deltas, const gl_t* pi_hash);
int orc_plonk_identity_check_gates(const orc_fri_params* P, unsigned num_routed, unsigned degree, const gl_t* openings,
                                   gl2_t zeta, const gl_t* betas, const gl_t* gammas, const gl_t* alphas,
                                   const orc_gate* gates, unsigned n_gates, unsigned num_selectors, const gl_t* pi_hash) {
  orc_circuit CK = {num_routed, degree, gates, n_gates, num_selectors, NULL, 0};
  return orc_identity_check_circuit(P, &CK, openings, zeta, betas, gammas, alphas, NULL, pi_hash);
}
int orc_plonk_identity_check(const orc_fri_params* P, unsigned num_routed, unsigned degree, const gl_t* openings,
                             gl2_t zeta, const gl_t* betas, const gl_t* gammas, const gl_t* alphas) {
  return orc_plonk_identity_check_gates(P, num_routed, degree, openings, zeta, betas, gammas, alphas, NULL, 0, 0, NULL);
}
// with gates: the gate constraints evaluated over the extension field on the opened constants / wires
// (plonk/vanishing_poly.rs eval_vanishing_poly) follow the permutation terms and, when the circuit has lookup
// tables, the lookup terms. deltas = [nc][4] (NULL without lookups). Openings in FRI batch order.
int orc_identity_check_circuit(const orc_fri_params* P, const orc_circuit* CK, const gl_t* openings, gl2_t zeta, const gl_t* betas,
                               const gl_t* gammas, const gl_t* alphas, const gl_t* deltas, const gl_t* pi_hash) {
  const unsigned num_routed = CK->num_routed, degree = CK->degree, n_gates = CK->n_gates;
  unsigned k = P->log_n, nc = P->zs_count, chunks = num_routed / degree, num_prods = chunks - 1;
  size_t n = (size_t)1 << k;
  const size_t L = n_lookup(P);
  const int has_lookup = CK->n_luts && L;
  size_t o_sig = P->oracle_w[0] - num_routed, o_w = P->oracle_w[0], o_z = o_w + P->oracle_w[1];
  size_t o_q = o_z + P->oracle_w[2] - L, o_lu = o_q + P->oracle_w[3], o_next = o_lu + L, o_lu_next = o_next + nc;
#define OPEN(i) ((gl2_t){{openings[2 * (i)], openings[2 * (i) + 1]}})
  gl2_t zn = zeta;
  for (unsigned i = 0; i < k; i++) zn = gl2_mul(zn, zn);
  gl2_t zh = gl2_sub(zn, gl2_from(1));
  gl2_t l0 = gl2_mul(zh, gl2_inv(gl2_scale(gl2_sub(zeta, gl2_from(1)), (gl_t)n % GL_P)));
  gl2_t terms[64 + 2 * 64 + ORC_MAX_GATE_CONSTRAINTS];
  size_t t = 0;
  for (unsigned c = 0; c < nc; c++) terms[t++] = gl2_mul(l0, gl2_sub(OPEN(o_z + c), gl2_from(1)));
  gl_t kj = 1;
  gl_t k_is[256];
  for (unsigned j = 0; j < num_routed; j++) { k_is[j] = kj; kj = gl_mul(kj, GL_MULT_GEN); }
  for (unsigned c = 0; c < nc; c++) {
    for (unsigned ch = 0; ch < chunks; ch++) {
      gl2_t num = gl2_from(1), den = gl2_from(1);
      for (unsigned j = ch * degree; j < (ch + 1) * degree; j++) {
        gl2_t wv = OPEN(o_w + j);
        num = gl2_mul(num, gl2_add(gl2_add(wv, gl2_scale(zeta, gl_mul(betas[c], k_is[j]))), gl2_from(gammas[c])));
        den = gl2_mul(den, gl2_add(gl2_add(wv, gl2_scale(OPEN(o_sig + j), betas[c])), gl2_from(gammas[c])));
      }
      gl2_t prev = ch == 0 ? OPEN(o_z + c) : OPEN(o_z + nc + (size_t)c * num_prods + ch - 1);
      gl2_t next = ch == chunks - 1 ? OPEN(o_next + c) : OPEN(o_z + nc + (size_t)c * num_prods + ch);
      terms[t++] = gl2_sub(gl2_mul(prev, num), gl2_mul(next, den));
    }
  }
  gl2_t lc[64], lw[256], ph[4];
  for (size_t j = 0; j < o_sig; j++) lc[j] = OPEN(j);
  for (size_t j = 0; j < P->oracle_w[1]; j++) lw[j] = OPEN(o_w + j);
  if (has_lookup) {
    orc_lookup_ctx LU = {CK->luts, CK->n_luts, 0, 0, 0, 0, 0};
    orc_lookup_shape(&LU, num_routed, degree);
    const unsigned nlp = LU.num_sldc + 1;
    gl2_t lz[16], lzn[16];
    for (unsigned c = 0; c < nc; c++) {
      for (unsigned q = 0; q < nlp; q++) { lz[q] = OPEN(o_lu + (size_t)c * nlp + q); lzn[q] = OPEN(o_lu_next + (size_t)c * nlp + q); }
      t += orc_lookup_terms_ext(&LU, lc + CK->num_selectors, lw, lz, lzn, deltas + 4 * c, terms + t);
    }
  }
  if (n_gates) {
    for (int j = 0; j < 4; j++) ph[j] = gl2_from(pi_hash[j]);
    t += orc_gates_eval_ext(CK->gates, n_gates, CK->num_selectors, has_lookup ? ORC_LOOKUP_SELECTORS + CK->n_luts : 0, lc, lw, ph, terms + t);
  }
  for (unsigned a = 0; a < nc; a++) {
    gl2_t van = gl2_from(0);
    for (size_t i = t; i-- > 0;) van = gl2_add(gl2_scale(van, alphas[a]), terms[i]);
    gl2_t tz = gl2_from(0);
    for (unsigned i = 8; i-- > 0;) tz = gl2_add(gl2_mul(tz, zn), OPEN(o_q + (size_t)a * 8 + i));
    if (!gl2_eq(van, gl2_mul(zh, tz))) return 1 + (int)a;
  }
#undef OPEN
  return 0;
}

// plonk/verifier.rs verify_with_challenges: the challenges re-derived from the transcript (get_challenges), the
// PLONK identity at zeta from the opened values (eval_vanishing_poly with the lookup and gate terms), then the
// FRI verifier. 0 = accept; 10 + a = identity fails for challenge a; 1..5 = FRI codes.
int orc_verify_circuit(const orc_fri_params* P, const gl_t circuit_digest[4], const gl_t pi_hash[4], const orc_circuit* CK,
                       const gl_t* caps, const gl_t* openings, const gl_t* proof) {
  size_t capw = ((size_t)4) << P->cap_height;
  const unsigned nc = num_challenges(P);
  const int has_lookup = CK->n_luts && P->num_lookup_polys;
  orc_challenger ch;
  orc_ch_init(&ch, P->variant);
  orc_ch_observe(&ch, circuit_digest, 4);
  orc_ch_observe(&ch, pi_hash, 4);
  gl_t bg[8] = {0}, al[2] = {0, 0};
  for (uint32_t o = 1; o < P->n_oracles; o++) {
    orc_ch_observe(&ch, caps + o * capw, capw);
    if (o == 1) for (uint32_t i = 0; i < (has_lookup ? 4 : 2) * nc; i++) bg[i] = orc_ch_get(&ch);
    else if (o == 2) for (uint32_t i = 0; i < nc; i++) al[i] = orc_ch_get(&ch);
  }
  gl2_t zeta = orc_ch_get_ext(&ch);
  int rc = orc_identity_check_circuit(P, CK, openings, zeta, bg, bg + nc, al, bg, pi_hash);
  if (rc) return 10 + rc - 1;
  return orc_pcs_verify(P, circuit_digest, pi_hash, caps, openings, proof);
}
int orc_verify_gates(const orc_fri_params* P, const gl_t circuit_digest[4], const gl_t pi_hash[4], unsigned num_routed,
                     unsigned degree, const orc_gate* gates, unsigned n_gates, unsigned num_selectors, const gl_t* caps,
                     const gl_t* openings, const gl_t* proof) {
  orc_circuit CK = {num_routed, degree, gates, n_gates, num_selectors, NULL, 0};
  return orc_verify_circuit(P, circuit_digest, pi_hash, &CK, caps, openings, proof);
}
