// Batched PCS prover: host-side orchestration of the commitment / Fiat-Shamir / opening / FRI
// skeleton of plonky2's prove() (plonk/prover.rs) for B same-shape proofs, plus the granular
// challenger / fold / proof-of-work entry points of include/mp2g.h. Everything between the
// input matrices and the finished proofs stays in HBM on one stream: no host synchronisation.
// Reference call sites: recursion-framework/src/circuit_builder.rs:308 (base proof),
// universal_verifier_gadget/wrap_circuit.rs:143 (wrap proofs), verifiable-db/src/api.rs:207.
#include "ctx.h"
#include "fri.h"
#include "zperm.h"
#include "gates.h"
#include "lookup.h"
#include <cstring>
#include <new>
#include <vector>

using namespace mp2g;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail("%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define NEED(c, msg) do { if (!(c)) return fail("invalid argument: %s", msg); } while (0)

struct mp2g_challenger {
  mp2g_ctx* ctx = nullptr;
  int variant = 0;
  uint32_t count = 0;
  DevBuf st;
};

struct mp2g_prover {
  mp2g_ctx* ctx = nullptr;
  mp2g_fri_params P{};
  uint32_t B = 0;     // proofs per prove() call (mp2g_prover_set_active: 1 .. Bcap)
  uint32_t Bcap = 0;  // proofs the buffers were allocated for
  bool have_pre = false;
  size_t capw = 0, levels_words = 0, proof_words = 0, q_words = 0, q_off = 0, final_off = 0, final_len = 0, n_open = 0;
  DevBuf coeffs[8], values[8], levels[8];
  DevBuf ch, chal, zeta, alpha, betas, comp, quot, final_poly, witness, qchal;
  DevBuf fvals[9], flevels[8], fcoeffs[9];
  // permutation argument computed on the device (mp2g_prover_enable_permutation)
  uint32_t num_routed = 0, degree = 0;
  bool quotient = false;
  DevBuf pre_values, zs_values, chunk_q, bg, alphas, qvals;
  // gate constraints (mp2g_prover_set_gates)
  GateTable gates{};
  // lookup argument (mp2g_prover_set_lookups): tables on the device, per-proof table polynomials
  LookupDev lookups{};
  DevBuf lut_tables, lut_eval;
  // PublicInputGate row whose first four wires the prover fills from d_pi_hash (mp2g_prover_bind_public_inputs)
  int64_t pi_row = -1;
  // witness check (mp2g_prover_enable_witness_check): bit 0 copy constraints, bit 1 gate constraints
  bool wcheck = false;
  DevBuf wflags;
  // stage timing (mp2g_prover_enable_timing): events at the phase boundaries of the last prove
  bool timing = false;
  hipEvent_t ev[MP2G_N_STAGES + 1] = {};
  // hipGraph replay of the whole launch sequence (mp2g_prover_enable_graph)
  bool graph_on = false;
  int plain_calls = 0;  // the first call runs un-captured: it creates the cached twiddle / coset tables
  hipGraphExec_t gexec = nullptr;
  const void* gkey[12] = {};
  u64 ggen = 0;  // NttEngine::generation at capture time
  void drop_graph() {
    if (gexec) (void)hipGraphExecDestroy(gexec);
    gexec = nullptr;
  }
  // working buffers that live in the context's shared scratch (ctx.h): what each asked for; bound to addresses at every prove
  std::vector<std::pair<DevBuf*, size_t>> shared;
  const u64* bound_base = nullptr;
  size_t bound_total = 0;
  bool counted = false;  // this prover is one of ctx->scratch_users (it has bound buffers into the shared scratch)
  // a per-batch working buffer: in the shared scratch (recorded, bound later) or, with sharing off, memory of its own
  hipError_t want(DevBuf& d, size_t bytes) {
    if (!ctx->share_scratch) return d.alloc(bytes);
    for (auto& e : shared)
      if (e.first == &d) { e.second = bytes; bound_base = nullptr; return hipSuccess; }
    shared.emplace_back(&d, bytes);
    bound_base = nullptr;
    return hipSuccess;
  }
  ~mp2g_prover() {
    drop_graph();
    for (hipEvent_t e : ev)
      if (e) (void)hipEventDestroy(e);
  }
};
#define STAGE_MARK(pr, i) do { if ((pr)->timing) CK(hipEventRecord((pr)->ev[i], (pr)->ctx->stream)); } while (0)

namespace {
// frees a temporary prover on every exit path of the convenience entry points
struct ProverGuard {
  mp2g_prover* pr = nullptr;
  ~ProverGuard();
};
}  // namespace

namespace mp2g {
int params_check(const mp2g_fri_params* p) {
  NEED(p, "params");
  NEED(p->variant <= 1, "variant");
  NEED(p->log_n >= 1 && p->log_n <= 20, "1 <= log_n <= 20");
  NEED(p->rate_bits >= 1 && p->rate_bits <= 6, "rate_bits");
  NEED(p->n_oracles >= 1 && p->n_oracles <= 8, "n_oracles");
  NEED(p->n_layers <= 8, "n_layers");
  NEED(p->pow_bits <= 32, "pow_bits <= 32");
  NEED(p->num_queries <= 64, "num_queries <= 64");
  NEED(p->zs_oracle < p->n_oracles && p->zs_count <= p->oracle_w[p->zs_oracle], "zs_oracle/zs_count");
  uint32_t deg = p->log_n, lg = p->log_n + p->rate_bits;
  NEED(p->cap_height <= lg, "cap_height");
  for (uint32_t i = 0; i < p->n_layers; i++) {
    NEED(p->arity_bits[i] >= 1 && p->arity_bits[i] <= 4, "arity_bits in 1..4");
    NEED(deg >= p->arity_bits[i], "arity exceeds degree");
    deg -= p->arity_bits[i];
    lg -= p->arity_bits[i];
    NEED(lg >= p->cap_height, "layer smaller than cap");
  }
  for (uint32_t o = 0; o < p->n_oracles; o++) NEED(p->oracle_w[o] >= 1, "oracle_w >= 1");
  NEED(p->num_lookup_polys <= 16, "num_lookup_polys <= 16");
  NEED((uint64_t)p->zs_count * (1 + p->num_lookup_polys) <= p->oracle_w[p->zs_oracle], "Z + lookup polynomials exceed the zs oracle");
  return 0;
}
}  // namespace mp2g

extern "C" {

uint32_t mp2g_reduction_arity_bits(uint32_t degree_bits, uint32_t rate_bits, uint32_t cap_height, uint32_t arity_bits,
                                   uint32_t final_poly_bits, uint32_t* out) {
  uint32_t n = 0;
  if (arity_bits == 0) return 0;
  // plonky2 asserts degree_bits >= arity_bits inside the loop; stop instead of wrapping the u32
  while (degree_bits > final_poly_bits && degree_bits >= arity_bits && degree_bits + rate_bits >= cap_height + arity_bits && n < 8) {
    out[n++] = arity_bits;
    degree_bits -= arity_bits;
  }
  return n;
}
size_t mp2g_fri_n_openings(const mp2g_fri_params* p) {
  size_t t = p->zs_count + (size_t)p->zs_count * p->num_lookup_polys;
  for (uint32_t o = 0; o < p->n_oracles; o++) t += p->oracle_w[o];
  return t;
}
static size_t query_words(const mp2g_fri_params* p) {
  uint32_t lg = p->log_n + p->rate_bits;
  size_t q = 0;
  for (uint32_t o = 0; o < p->n_oracles; o++) q += p->oracle_w[o] + 4 * (lg - p->cap_height);
  uint32_t cur = lg;
  for (uint32_t i = 0; i < p->n_layers; i++) {
    cur -= p->arity_bits[i];
    q += ((size_t)2 << p->arity_bits[i]) + 4 * (cur - p->cap_height);
  }
  return q;
}
static size_t final_poly_len(const mp2g_fri_params* p) {
  uint32_t deg = p->log_n;
  for (uint32_t i = 0; i < p->n_layers; i++) deg -= p->arity_bits[i];
  return (size_t)1 << deg;
}
size_t mp2g_fri_proof_words(const mp2g_fri_params* p) {
  return p->n_layers * ((size_t)4 << p->cap_height) + p->num_queries * query_words(p) + 2 * final_poly_len(p) + 1;
}

// ---- challenger ------------------------------------------------------------------------------
int mp2g_challenger_create(mp2g_ctx* c, int variant, uint32_t count, mp2g_challenger** out) {
  NEED(c && out && count >= 1, "ctx/out/count");
  NEED(variant == 0 || variant == 1, "variant");
  mp2g_challenger* ch = new (std::nothrow) mp2g_challenger();
  if (!ch) return fail("out of memory");
  ch->ctx = c; ch->variant = variant; ch->count = count;
  hipError_t e = ch->st.alloc(sizeof(ChState) * count);
  if (e == hipSuccess) e = challenger_init(c->stream, (ChState*)ch->st.p, count);
  if (e != hipSuccess) { delete ch; return fail("challenger_create: %s", hipGetErrorString(e)); }
  *out = ch;
  return 0;
}
int mp2g_challenger_observe(mp2g_challenger* ch, const uint64_t* elems, uint32_t n) {
  NEED(ch && (elems || !n), "challenger/elems");
  if (!n) return 0;
  mp2g_ctx* c = ch->ctx;
  DevBuf d;
  CK(d.alloc((size_t)ch->count * n * sizeof(u64)));
  CK(hipMemcpyAsync(d.p, elems, (size_t)ch->count * n * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(challenger_step(c->stream, ch->variant, (ChState*)ch->st.p, ch->count, d.p, n, n, d.p, 0, 0));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}
int mp2g_challenger_get(mp2g_challenger* ch, uint32_t n, uint64_t* out) {
  NEED(ch && (out || !n), "challenger/out");
  if (!n) return 0;
  mp2g_ctx* c = ch->ctx;
  DevBuf d;
  CK(d.alloc((size_t)ch->count * n * sizeof(u64)));
  CK(challenger_step(c->stream, ch->variant, (ChState*)ch->st.p, ch->count, d.p, 0, 0, d.p, n, n));
  CK(hipMemcpyAsync(out, d.p, (size_t)ch->count * n * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}
void mp2g_challenger_free(mp2g_challenger* ch) {
  if (!ch) return;
  (void)hipStreamSynchronize(ch->ctx->stream);
  delete ch;
}

// ---- stand-alone fold / proof of work --------------------------------------------------------
int mp2g_fri_fold(mp2g_ctx* c, const uint64_t* evals, uint32_t log_m, uint32_t arity_bits, const uint64_t beta[2],
                  uint64_t shift, uint64_t* out) {
  NEED(c && evals && beta && out, "ctx/pointers");
  NEED(arity_bits >= 1 && arity_bits <= 4 && log_m >= arity_bits && log_m <= 28, "arity_bits in 1..4, log_m >= arity_bits");
  NEED(shift != 0 && shift < GL_P, "shift");
  const size_t m = (size_t)1 << log_m, m2 = m >> arity_bits;
  // AoS host layout -> SoA device layout
  std::vector<u64> soa(2 * m);
  for (size_t i = 0; i < m; i++) { soa[i] = evals[2 * i]; soa[m + i] = evals[2 * i + 1]; }
  DevBuf din, dout, db;
  CK(din.alloc(2 * m * sizeof(u64)));
  CK(dout.alloc(2 * m2 * sizeof(u64)));
  CK(db.alloc(2 * sizeof(u64)));
  CK(hipMemcpyAsync(din.p, soa.data(), 2 * m * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(hipMemcpyAsync(db.p, beta, 2 * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(fri_fold_values(c->stream, 1, log_m, arity_bits, din.p, 0, dout.p, 0, db.p, 0, shift));
  std::vector<u64> res(2 * m2);
  CK(hipMemcpyAsync(res.data(), dout.p, 2 * m2 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < m2; i++) { out[2 * i] = res[i]; out[2 * i + 1] = res[m2 + i]; }
  return 0;
}
int mp2g_fri_pow(mp2g_ctx* c, int variant, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t* witness) {
  NEED(c && state && witness, "ctx/pointers");
  NEED(variant == 0 || variant == 1, "variant");
  NEED(pos < 8 && bits <= 32, "pos < 8, bits <= 32");
  ChState h{};
  for (int i = 0; i < 12; i++) h.state[i] = state[i];
  for (uint32_t i = 0; i < pos; i++) h.in[i] = state[i];
  h.n_in = pos;
  DevBuf ds, dw;
  CK(ds.alloc(sizeof(ChState)));
  CK(dw.alloc(FRI_POW_STRIDE * sizeof(u64)));
  CK(hipMemcpyAsync(ds.p, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
  CK(fri_pow(c->stream, variant, (const ChState*)ds.p, 1, bits, dw.p));
  CK(hipMemcpyAsync(witness, dw.p, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- batched prover --------------------------------------------------------------------------
static int prover_create_impl(mp2g_ctx* c, const mp2g_fri_params* params, uint32_t batch, bool alloc_oracles, mp2g_prover** out) {
  NEED(c && out, "ctx/out");
  NEED(batch >= 1 && batch <= 4096, "1 <= batch <= 4096");
  int rc = params_check(params);
  if (rc) return rc;
  mp2g_prover* pr = new (std::nothrow) mp2g_prover();
  if (!pr) return fail("out of memory");
  pr->ctx = c; pr->P = *params; pr->B = pr->Bcap = batch;
  const mp2g_fri_params& P = pr->P;
  const size_t n = (size_t)1 << P.log_n, N = n << P.rate_bits, B = batch;
  const uint32_t lg = P.log_n + P.rate_bits;
  pr->capw = (size_t)4 << P.cap_height;
  pr->levels_words = merkle_levels_words(lg, P.cap_height);
  pr->proof_words = mp2g_fri_proof_words(&P);
  pr->q_words = query_words(&P);
  pr->q_off = P.n_layers * pr->capw;
  pr->final_len = final_poly_len(&P);
  pr->final_off = pr->q_off + P.num_queries * pr->q_words;
  pr->n_open = mp2g_fri_n_openings(&P);
  hipError_t e = hipSuccess;
  auto A = [&](DevBuf& d, size_t words) { if (e == hipSuccess) e = d.alloc(words * sizeof(u64)); };       // the prover's own (small, or persistent)
  auto S = [&](DevBuf& d, size_t words) { if (e == hipSuccess) e = pr->want(d, words * sizeof(u64)); };  // per-batch working memory
  for (uint32_t o = 0; o < P.n_oracles && alloc_oracles; o++) {
    if (o == 0) {  // the preprocessed oracle: committed once, read by every prove
      A(pr->coeffs[o], (size_t)P.oracle_w[o] * n); A(pr->values[o], (size_t)P.oracle_w[o] * N); A(pr->levels[o], pr->levels_words);
    } else {
      S(pr->coeffs[o], B * P.oracle_w[o] * n); S(pr->values[o], B * P.oracle_w[o] * N); S(pr->levels[o], B * pr->levels_words);
    }
  }
  if (e == hipSuccess) e = pr->ch.alloc(B * sizeof(ChState));
  A(pr->chal, B * 8); A(pr->zeta, B * 2); A(pr->alpha, B * 2); A(pr->betas, B * 16);
  S(pr->comp, B * 4 * n); S(pr->quot, B * 4 * n); S(pr->final_poly, B * 2 * n);
  A(pr->witness, B * FRI_POW_STRIDE); A(pr->qchal, B * (P.num_queries ? P.num_queries : 1));
  size_t m = N, nc = n;
  uint32_t clg = lg;
  S(pr->fvals[0], B * 2 * m);
  for (uint32_t li = 0; li < P.n_layers; li++) {
    clg -= P.arity_bits[li];
    S(pr->flevels[li], B * merkle_levels_words(clg, P.cap_height));
    m >>= P.arity_bits[li]; nc >>= P.arity_bits[li];
    S(pr->fvals[li + 1], B * 2 * m);
    S(pr->fcoeffs[li + 1], B * 2 * nc);
  }
  if (e != hipSuccess) { delete pr; return fail("prover_create: %s", hipGetErrorString(e)); }
  *out = pr;
  return 0;
}
int mp2g_prover_create(mp2g_ctx* c, const mp2g_fri_params* params, uint32_t batch, mp2g_prover** out) {
  return prover_create_impl(c, params, batch, true, out);
}
}  // extern "C"
ProverGuard::~ProverGuard() { if (pr) mp2g_prover_free(pr); }
extern "C" {
void mp2g_prover_free(mp2g_prover* pr) {
  if (!pr) return;
  mp2g_ctx* c = pr->ctx;
  (void)hipStreamSynchronize(c->stream);
  const bool counted = pr->counted;
  delete pr;
  // the shared scratch goes with the LAST prover that used it: a long-lived context that ran one large one-shot prove (mp2g_pcs_prove,
  // mp2g_fri_prove, a bench leg) does not keep that working set until mp2g_ctx_destroy
  if (counted && c->scratch_users && --c->scratch_users == 0) c->prover_scratch.release();
}
// coefficients in pr->coeffs[o] -> LDE values, Merkle levels (PolynomialBatch::from_coeffs)
static hipError_t commit_oracle_coeffs(mp2g_prover* pr, uint32_t o, uint32_t nb) {
  mp2g_ctx* c = pr->ctx;
  const mp2g_fri_params& P = pr->P;
  const u64 n = (u64)1 << P.log_n, N = n << P.rate_bits;
  const uint32_t w = P.oracle_w[o], lg = P.log_n + P.rate_bits;
  hipError_t e;
  CosetTables* pre;
  e = c->ntt.coset(P.log_n, P.rate_bits, GL_MULT_GEN, &pre);
  if (e != hipSuccess) return e;
  e = c->ntt.run(pr->coeffs[o].p, pr->values[o].p, P.log_n, nb * w, P.rate_bits, n, N, false, pre, true);
  if (e != hipSuccess) return e;
  e = leaf_hash_poly_major(c->stream, P.variant, pr->values[o].p, w, N, N, pr->levels[o].p, nb, (u64)w * N, pr->levels_words);
  if (e != hipSuccess) return e;
  return merkle_reduce(c->stream, P.variant, pr->levels[o].p, lg, P.cap_height, nb, pr->levels_words);
}
// values [nb][w][n] -> coeffs, LDE values, Merkle levels of oracle o (PolynomialBatch::from_values)
static hipError_t commit_oracle(mp2g_prover* pr, uint32_t o, const u64* d_values, uint32_t nb) {
  const mp2g_fri_params& P = pr->P;
  const u64 n = (u64)1 << P.log_n;
  hipError_t e = pr->ctx->ntt.run(d_values, pr->coeffs[o].p, P.log_n, nb * P.oracle_w[o], 0, n, n, true, nullptr, false);
  if (e != hipSuccess) return e;
  return commit_oracle_coeffs(pr, o, nb);
}
int mp2g_prover_set_preprocessed_dev(mp2g_prover* pr, const uint64_t* d_values) {
  NEED(pr && d_values, "prover/values");
  const size_t words = (size_t)pr->P.oracle_w[0] << pr->P.log_n;
  CK(pr->pre_values.alloc(words * sizeof(u64)));  // the sigma values are needed again for Z
  CK(hipMemcpyAsync(pr->pre_values.p, d_values, words * sizeof(u64), hipMemcpyDeviceToDevice, pr->ctx->stream));
  CK(commit_oracle(pr, 0, (const u64*)d_values, 1));
  pr->drop_graph();
  pr->have_pre = true;
  return 0;
}
int mp2g_prover_enable_permutation(mp2g_prover* pr, uint32_t num_routed, uint32_t degree) {
  NEED(pr && pr->have_pre, "call mp2g_prover_set_preprocessed_dev first");
  const mp2g_fri_params& P = pr->P;
  NEED(P.n_oracles >= 3 && P.zs_oracle == 2, "needs wires = oracle 1 and Z/partial products = oracle 2");
  NEED(degree >= 1 && num_routed >= degree && num_routed % degree == 0 && num_routed / degree <= 16, "num_routed/degree");
  NEED(num_routed <= P.oracle_w[0] && num_routed <= P.oracle_w[1], "num_routed exceeds the sigma / wire counts");
  NEED(P.zs_count >= 1 && P.zs_count <= 2, "1 or 2 challenges");
  NEED(P.oracle_w[2] == P.zs_count * (num_routed / degree + P.num_lookup_polys), "oracle_w[2] must be zs_count * (num_routed/degree + num_lookup_polys)");
  const size_t n = (size_t)1 << P.log_n;
  CK(pr->want(pr->zs_values, (size_t)pr->Bcap * P.oracle_w[2] * n * sizeof(u64)));
  CK(pr->want(pr->chunk_q, (size_t)pr->Bcap * P.zs_count * (num_routed / degree) * n * sizeof(u64)));
  CK(pr->bg.alloc((size_t)pr->Bcap * 8 * sizeof(u64)));  // betas, gammas (+ the 2 * num_challenges extra lookup challenges)
  CK(pr->alphas.alloc((size_t)pr->Bcap * 2 * sizeof(u64)));
  pr->num_routed = num_routed; pr->degree = degree;
  pr->drop_graph();
  return 0;
}
int mp2g_prover_enable_quotient(mp2g_prover* pr) {
  NEED(pr && pr->num_routed, "call mp2g_prover_enable_permutation first");
  const mp2g_fri_params& P = pr->P;
  NEED(P.n_oracles == 4 && P.rate_bits == 3, "needs the four plonky2 oracles and rate_bits 3 (quotient degree factor 8)");
  NEED(P.oracle_w[3] == P.zs_count * 8, "oracle_w[3] must be zs_count * 8 quotient chunks");
  NEED(P.log_n + 3 <= 24, "log_n <= 21");
  CK(pr->want(pr->qvals, (size_t)pr->Bcap * P.zs_count * ((size_t)8 << P.log_n) * sizeof(u64)));
  pr->quotient = true;
  pr->drop_graph();
  return 0;
}
int mp2g_prover_set_gates(mp2g_prover* pr, const mp2g_gate* gates, uint32_t n_gates, uint32_t num_selectors) {
  NEED(pr && pr->quotient, "call mp2g_prover_enable_quotient first");
  NEED(n_gates <= MP2G_MAX_GATES, "at most MP2G_MAX_GATES gates");
  GateTable t{};
  t.n_gates = n_gates; t.num_selectors = num_selectors;
  t.num_lookup_selectors = pr->gates.num_lookup_selectors;
  if (n_gates) {
    NEED(gates, "gates");
    for (uint32_t i = 0; i < n_gates; i++) t.g[i] = gates[i];
    const char* msg = gate_table_check(t, pr->P.oracle_w[0] - pr->num_routed, pr->P.oracle_w[1]);
    if (msg) return fail("invalid gate table: %s", msg);
    // filter degree + gate degree <= quotient_degree_factor + 1 (gates/selectors.rs is called with that bound:
    // a degree-9n vanishing polynomial divided by Z_H fits the 8n-point coset)
    for (uint32_t i = 0; i < n_gates; i++) {
      uint32_t fdeg = t.g[i].group_end - t.g[i].group_start - 1 + (num_selectors > 1 ? 1 : 0);
      if (fdeg + gate_degree(t.g[i]) > 9) return fail("invalid gate table: filtered degree of gate %u exceeds quotient degree factor + 1 = 9", i);
    }
  }
  pr->gates = t;
  pr->drop_graph();
  return 0;
}
}  // extern "C"

// fri/oracle.rs prove_openings + fri/prover.rs fri_proof for B transcripts: alpha, batch composition,
// LDE, commit phase, final polynomial, proof of work, query rounds. `sh` names the committed
// oracles, `st` the challengers (openings already observed), pr->zeta the opening points.
static int fri_tail(mp2g_prover* pr, const FriShape& sh, ChState* st, u64* d_proof) {
  mp2g_ctx* c = pr->ctx;
  hipStream_t s = c->stream;
  const mp2g_fri_params& P = pr->P;
  const uint32_t B = pr->B, lg = P.log_n + P.rate_bits, V = P.variant;
  const u64 n = (u64)1 << P.log_n, N = n << P.rate_bits;
  const size_t capw = pr->capw;
  u64* chal = pr->chal.p;
  CK(challenger_step(s, V, st, B, chal, 0, 0, pr->alpha.p, 2, 2));  // alpha
  CK(fri_final_poly(s, sh, B, pr->alpha.p, 2, pr->zeta.p, 2, pr->comp.p, pr->quot.p, pr->final_poly.p));
  CosetTables* pre;
  CK(c->ntt.coset(P.log_n, P.rate_bits, GL_MULT_GEN, &pre));
  CK(c->ntt.run(pr->final_poly.p, pr->fvals[0].p, P.log_n, 2 * B, P.rate_bits, n, N, false, pre, true));

  FriLayers ly{};
  ly.n_layers = P.n_layers;
  u64 m = N, nc = n, shift = GL_MULT_GEN;
  uint32_t clg = lg;
  u64* proof = (u64*)d_proof;
  const u64* cur_coeffs = pr->final_poly.p;
  for (uint32_t li = 0; li < P.n_layers; li++) {
    const uint32_t ab = P.arity_bits[li];
    const uint32_t log_leaves = clg - ab;
    const size_t lw = merkle_levels_words(log_leaves, P.cap_height);
    CK(leaf_hash_ext_soa(s, V, pr->fvals[li].p, pr->fvals[li].p + m, ab, (u64)1 << log_leaves, pr->flevels[li].p, B, 2 * m, lw));
    CK(merkle_reduce(s, V, pr->flevels[li].p, log_leaves, P.cap_height, B, lw));
    CK(copy_rows(s, B, pr->flevels[li].p + lw - capw, lw, proof + li * capw, pr->proof_words, (u32)capw));
    u64* beta = pr->betas.p + 2 * li;
    CK(challenger_step(s, V, st, B, proof + li * capw, pr->proof_words, (u32)capw, beta, 16, 2));
    ly.arity_bits[li] = ab;
    ly.values[li] = pr->fvals[li].p; ly.value_bstride[li] = 2 * m;
    ly.levels[li] = pr->flevels[li].p; ly.level_bstride[li] = lw;
    CK(fri_fold_values(s, B, clg, ab, pr->fvals[li].p, 2 * m, pr->fvals[li + 1].p, 2 * (m >> ab), beta, 16, shift));
    const bool last = li + 1 == P.n_layers;
    if (last) CK(fri_fold_coeffs(s, B, (u32)nc, ab, cur_coeffs, 2 * nc, proof + pr->final_off, pr->proof_words, beta, 16, true));
    else CK(fri_fold_coeffs(s, B, (u32)nc, ab, cur_coeffs, 2 * nc, pr->fcoeffs[li + 1].p, 2 * (nc >> ab), beta, 16, false));
    cur_coeffs = pr->fcoeffs[li + 1].p;
    shift = gl_pow(shift, (u64)1 << ab);
    m >>= ab; nc >>= ab; clg -= ab;
  }
  if (P.n_layers == 0) CK(fri_soa_to_aos(s, B, (u32)n, pr->final_poly.p, 2 * n, (u32)n, proof + pr->final_off, pr->proof_words));
  CK(challenger_step(s, V, st, B, proof + pr->final_off, pr->proof_words, (u32)(2 * pr->final_len), chal, 8, 0));
  STAGE_MARK(pr, 5);  // FRI batch composition + commit phase
  CK(fri_pow(s, V, st, B, P.pow_bits, pr->witness.p));
  STAGE_MARK(pr, 6);  // proof of work
  CK(copy_rows(s, B, pr->witness.p, FRI_POW_STRIDE, proof + pr->final_off + 2 * pr->final_len, pr->proof_words, 1));
  CK(challenger_step(s, V, st, B, pr->witness.p, FRI_POW_STRIDE, 1, chal, 8, 1));  // observe witness, draw pow response
  if (P.num_queries) {
    CK(challenger_step(s, V, st, B, chal, 0, 0, pr->qchal.p, P.num_queries, P.num_queries));
    CK(fri_queries(s, sh, ly, B, P.num_queries, pr->qchal.p, P.num_queries, proof, pr->proof_words, pr->q_off, pr->q_words));
  }
  return 0;
}

extern "C" {

static int prove_impl(mp2g_prover* pr, const uint64_t* const* d_values, const uint64_t* d_circuit_digest,
                      const uint64_t* d_pi_hash, uint64_t* d_caps, uint64_t* d_openings, uint64_t* d_proof);
// The prover's working buffers get their addresses in the context's shared scratch (ctx.h) -- before anything of this prove() is
// queued: a scratch that has to grow first waits for the stream (earlier provers' kernels may still run in the old one).
static int bind_scratch(mp2g_prover* pr) {
  if (pr->shared.empty()) return 0;
  mp2g_ctx* c = pr->ctx;
  if (!pr->counted) { pr->counted = true; c->scratch_users++; }
  size_t total = 0;
  for (auto& e : pr->shared) total += (e.second + 255) & ~(size_t)255;
  if (c->prover_scratch.bytes < total) {
    CK(hipStreamSynchronize(c->stream));
    hipError_t e = c->prover_scratch.alloc(total);
    if (e != hipSuccess) return fail("prover scratch of %zu bytes: %s", total, hipGetErrorString(e));
  }
  if (pr->bound_base == c->prover_scratch.p && pr->bound_total == total) return 0;
  size_t off = 0;
  for (auto& e : pr->shared) {
    e.first->borrow(c->prover_scratch.p + off / sizeof(u64), e.second);
    off += (e.second + 255) & ~(size_t)255;
  }
  pr->bound_base = c->prover_scratch.p;
  pr->bound_total = total;
  pr->drop_graph();  // a captured launch sequence holds the old addresses
  return 0;
}
int mp2g_prover_prove_dev(mp2g_prover* pr, const uint64_t* const* d_values, const uint64_t* d_circuit_digest,
                          const uint64_t* d_pi_hash, uint64_t* d_caps, uint64_t* d_openings, uint64_t* d_proof) {
  NEED(pr && d_values && d_circuit_digest && d_pi_hash && d_caps && d_openings && d_proof, "prover/pointers");
  NEED(pr->have_pre, "call mp2g_prover_set_preprocessed_dev first");
  { int rcb = bind_scratch(pr); if (rcb) return rcb; }
  if (!pr->graph_on || pr->timing || pr->plain_calls == 0) {
    pr->plain_calls++;
    return prove_impl(pr, d_values, d_circuit_digest, d_pi_hash, d_caps, d_openings, d_proof);
  }
  // The launch sequence depends only on the shape and on the buffer addresses: replay it as a graph while
  // the caller keeps handing over the same buffers.
  const void* key[12] = {d_circuit_digest, d_pi_hash, d_caps, d_openings, d_proof};
  for (uint32_t o = 1; o < pr->P.n_oracles && o < 8; o++) key[4 + o] = d_values[o - 1];
  hipStream_t s = pr->ctx->stream;
  if (pr->gexec && memcmp(key, pr->gkey, sizeof key) == 0 && pr->ggen == pr->ctx->ntt.generation) {
    CK(hipGraphLaunch(pr->gexec, s));
    return 0;
  }
  pr->drop_graph();
  hipGraph_t g = nullptr;
  // nothing may be (re)allocated inside a capture: the shared NTT scratch another prover or mp2g_ntt_dev call may
  // have left too small is grown here, before the capture begins (the largest two-pass natural-order transform of
  // prove_impl is the quotient iNTT: zs_count * B transforms of 8n points)
  if (pr->quotient)
    CK(pr->ctx->ntt.ensure_scratch((size_t)pr->B * pr->P.zs_count * ((size_t)8 << pr->P.log_n)));
  for (uint32_t o = 1; o < pr->P.n_oracles; o++) CK(pr->ctx->ntt.ensure_scratch((size_t)pr->B * pr->P.oracle_w[o] << pr->P.log_n));
  const u64 gen0 = pr->ctx->ntt.generation;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  int rc = prove_impl(pr, d_values, d_circuit_digest, d_pi_hash, d_caps, d_openings, d_proof);
  hipError_t e = hipStreamEndCapture(s, &g);
  if (!rc && e == hipSuccess && pr->ctx->ntt.generation != gen0) {
    if (g) (void)hipGraphDestroy(g);
    return fail("graph capture: an NTT buffer was reallocated inside the capture");
  }
  if (rc || e != hipSuccess) {
    if (g) (void)hipGraphDestroy(g);
    return rc ? rc : fail("graph capture: %s", hipGetErrorString(e));
  }
  e = hipGraphInstantiate(&pr->gexec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) { pr->gexec = nullptr; return fail("hipGraphInstantiate: %s", hipGetErrorString(e)); }
  memcpy(pr->gkey, key, sizeof key);
  pr->ggen = gen0;
  CK(hipGraphLaunch(pr->gexec, s));
  return 0;
}
int mp2g_prover_enable_graph(mp2g_prover* pr, int on) {
  NEED(pr, "prover");
  pr->graph_on = on != 0;
  if (!on) pr->drop_graph();
  return 0;
}
static int prove_impl(mp2g_prover* pr, const uint64_t* const* d_values, const uint64_t* d_circuit_digest,
                      const uint64_t* d_pi_hash, uint64_t* d_caps, uint64_t* d_openings, uint64_t* d_proof) {
  mp2g_ctx* c = pr->ctx;
  hipStream_t s = c->stream;
  const mp2g_fri_params& P = pr->P;
  const uint32_t B = pr->B, V = P.variant;
  const u64 n = (u64)1 << P.log_n, N = n << P.rate_bits;
  const size_t capw = pr->capw, LW = pr->levels_words;
  ChState* st = (ChState*)pr->ch.p;
  u64* chal = pr->chal.p;
  const u64 caps_b = P.n_oracles * capw;

  STAGE_MARK(pr, 0);
  if (pr->pi_row >= 0)
    CK(bind_public_inputs(s, B, (u64*)d_values[0], (u64)P.oracle_w[1] * n, n, (u32)pr->pi_row, (const u64*)d_pi_hash));
  if (pr->wcheck) {
    CK(hipMemsetAsync(pr->wflags.p, 0, B * sizeof(u32), s));
    if (pr->gates.n_gates)
      CK(gate_check(s, B, pr->gates, pr->pre_values.p, (const u64*)d_values[0], (u64)P.oracle_w[1] * n, n, (const u64*)d_pi_hash,
                    (u32*)pr->wflags.p));
  }
  CK(challenger_init(s, st, B));
  CK(challenger_step(s, V, st, B, (const u64*)d_circuit_digest, 0, 4, chal, 8, 0));
  CK(challenger_step(s, V, st, B, (const u64*)d_pi_hash, 4, 4, chal, 8, 0));
  CK(copy_rows(s, B, pr->levels[0].p + LW - capw, 0, (u64*)d_caps, caps_b, (u32)capw));
  for (uint32_t o = 1; o < P.n_oracles; o++) {
    const u64* vals = (const u64*)d_values[o - 1];
    if (o == 2 && pr->num_routed) {
      // betas = bg[0..nc), gammas = bg[nc..2nc) of every transcript (drawn after the wires cap:
      // get_n_challenges(num_challenges) twice)
      CK(zpp_compute(s, B, (const u64*)d_values[0], (u64)P.oracle_w[1] * n, pr->pre_values.p + (u64)(P.oracle_w[0] - pr->num_routed) * n,
                     P.log_n, pr->num_routed, pr->degree, pr->bg.p, pr->bg.p + P.zs_count, 8, P.zs_count, pr->chunk_q.p, pr->zs_values.p,
                     (u64)P.oracle_w[2] * n));
      if (pr->lookups.n_luts) {
        // compute_all_lookup_polys: deltas of round c = bg[4c .. 4c+4) (betas ++ gammas ++ the extra challenges)
        CK(lookup_table_polys(s, B, pr->lookups, pr->bg.p, 8, P.zs_count, pr->lut_eval.p));
        CK(lookup_polys(s, B, pr->lookups, (const u64*)d_values[0], (u64)P.oracle_w[1] * n, P.log_n, pr->bg.p, 8, P.zs_count,
                        pr->zs_values.p + (u64)P.zs_count * (pr->num_routed / pr->degree) * n, (u64)P.oracle_w[2] * n, pr->lut_eval.p,
                        pr->wcheck ? (u32*)pr->wflags.p : nullptr));
      }
      vals = pr->zs_values.p;
      if (pr->wcheck)
        CK(zpp_wrap_check(s, B, pr->chunk_q.p, pr->zs_values.p, (u64)P.oracle_w[2] * n, P.log_n, pr->num_routed / pr->degree, P.zs_count,
                          (u32*)pr->wflags.p));
    }
    if (o == 3 && pr->quotient) {
      // compute_quotient_polys (gate constraints first, then folded into the permutation terms): values on the coset, coset iFFT, and the
      // 8n coefficients of each challenge are its 8 degree-n chunks, already laid out as oracle 3's coeffs
      const u32 nc = P.zs_count;
      if (pr->gates.n_gates)
        CK(gate_constraints_lde(s, B, pr->gates, pr->values[0].p, pr->values[1].p, (u64)P.oracle_w[1] * N, P.log_n + 3, pr->alphas.p, 2,
                                nc, (const u64*)d_pi_hash, pr->qvals.p));
      // vanishing_all_lookup_terms sit between the partial-product and the gate terms of the alpha-reduction
      if (pr->lookups.n_luts)
        CK(quotient_lookup_values(s, B, pr->lookups, pr->values[0].p, pr->gates.num_selectors, pr->values[1].p, (u64)P.oracle_w[1] * N,
                                  pr->values[2].p, (u64)P.oracle_w[2] * N, nc * (pr->num_routed / pr->degree), P.log_n, pr->bg.p, 8,
                                  pr->lut_eval.p, pr->alphas.p, 2, nc, pr->qvals.p));
      CK(quotient_perm_values(s, B, pr->values[1].p, (u64)P.oracle_w[1] * N, pr->values[0].p + (u64)(P.oracle_w[0] - pr->num_routed) * N,
                              pr->values[2].p, (u64)P.oracle_w[2] * N, P.log_n, pr->num_routed, pr->degree, pr->bg.p, 8,
                              pr->alphas.p, 2, nc, pr->gates.n_gates != 0, pr->qvals.p));
      CK(c->ntt.run(pr->qvals.p, pr->coeffs[3].p, P.log_n + 3, B * nc, 0, N, N, true, nullptr, false));
      CK(c->ntt.scale_powers(pr->coeffs[3].p, P.log_n + 3, B * nc, gl_inv(GL_MULT_GEN), 1));
      CK(commit_oracle_coeffs(pr, 3, B));
    } else {
      NEED(vals, "d_values[o]");
      CK(commit_oracle(pr, o, vals, B));
    }
    u64* cap_dst = (u64*)d_caps + o * capw;
    CK(copy_rows(s, B, pr->levels[o].p + LW - capw, LW, cap_dst, caps_b, (u32)capw));
    // plonk/prover.rs: wires cap -> num_challenges betas, then as many gammas; zs cap -> num_challenges
    // alphas; all other caps -> 0
    // (the PCS-only skeleton, which accepts any zs_count, draws as for two rounds)
    const uint32_t nch = P.zs_count >= 1 && P.zs_count <= 2 ? P.zs_count : 2;
    uint32_t n_get = o == 1 ? (pr->lookups.n_luts ? 4 : 2) * nch : (o == 2 ? nch : 0);
    u64* dst = chal;
    u64 dst_stride = 8;
    if (pr->num_routed && o == 1) { dst = pr->bg.p; dst_stride = 8; }
    if (pr->num_routed && o == 2) { dst = pr->alphas.p; dst_stride = 2; }
    CK(challenger_step(s, V, st, B, cap_dst, caps_b, (u32)capw, dst, dst_stride, n_get));
    if (o <= 3) STAGE_MARK(pr, o);  // 1: wires committed, 2: Z / partial products, 3: quotient
  }
  if (P.n_oracles < 4) for (uint32_t o = P.n_oracles; o <= 3; o++) STAGE_MARK(pr, o);
  CK(challenger_step(s, V, st, B, chal, 0, 0, pr->zeta.p, 2, 2));  // zeta

  FriShape sh{};
  sh.log_n = P.log_n; sh.rate_bits = P.rate_bits; sh.cap_h = P.cap_height; sh.n_oracles = P.n_oracles;
  sh.zs_oracle = P.zs_oracle; sh.zs_count = P.zs_count; sh.lookup_count = P.zs_count * P.num_lookup_polys;
  for (uint32_t o = 0; o < P.n_oracles; o++) {
    OracleRef& r = sh.o[o];
    r.coeffs = pr->coeffs[o].p; r.values = pr->values[o].p; r.levels = pr->levels[o].p; r.w = P.oracle_w[o];
    r.coeff_bstride = o ? (u64)r.w * n : 0;
    r.value_bstride = o ? (u64)r.w * N : 0;
    r.level_bstride = o ? LW : 0;
    sh.n_polys += r.w;
  }
  CK(fri_openings(s, sh, B, pr->zeta.p, 2, (u64*)d_openings));
  CK(challenger_step(s, V, st, B, (const u64*)d_openings, 2 * pr->n_open, (u32)(2 * pr->n_open), chal, 8, 0));
  STAGE_MARK(pr, 4);  // openings
  int rc = fri_tail(pr, sh, st, (u64*)d_proof);
  if (rc) return rc;
  STAGE_MARK(pr, 7);
  return 0;
}
int mp2g_prover_set_lookups(mp2g_prover* pr, const mp2g_lookup* luts, uint32_t n_luts) {
  NEED(pr && pr->quotient, "call mp2g_prover_enable_quotient first");
  NEED(n_luts <= MP2G_MAX_LUTS, "at most MP2G_MAX_LUTS lookup tables");
  const mp2g_fri_params& P = pr->P;
  LookupDev L{};
  if (!n_luts) {
    NEED(P.num_lookup_polys == 0, "params.num_lookup_polys != 0 needs lookup tables");
    pr->lookups = L;
    pr->gates.num_lookup_selectors = 0;
    pr->drop_graph();
    return 0;
  }
  NEED(luts && pr->gates.n_gates, "luts; call mp2g_prover_set_gates first");
  L.n_luts = n_luts;
  L.num_lu_slots = pr->num_routed / 2; L.num_lut_slots = pr->num_routed / 3;
  L.lu_degree = pr->degree - 1;
  NEED(L.lu_degree >= 1 && L.num_lu_slots <= 40 && L.num_lut_slots >= 1, "slot geometry (num_routed <= 80, degree >= 2)");
  L.num_sldc = (L.num_lu_slots + L.lu_degree - 1) / L.lu_degree;
  L.lut_degree = (L.num_lut_slots + L.num_sldc - 1) / L.num_sldc;
  NEED(P.num_lookup_polys == L.num_sldc + 1, "params.num_lookup_polys must be ceil((num_routed/2) / (degree-1)) + 1");
  NEED(P.oracle_w[1] >= 3 * L.num_lut_slots && P.oracle_w[1] >= 2 * L.num_lu_slots, "wires");
  const uint32_t n_sel = 4 + n_luts, num_constants = P.oracle_w[0] - pr->num_routed;
  NEED(pr->gates.num_selectors + n_sel <= num_constants, "the constants must hold 4 + n_luts lookup selectors after the selectors");
  const uint64_t n = (uint64_t)1 << P.log_n;
  size_t total = 0;
  for (uint32_t r = 0; r < n_luts; r++) {
    const mp2g_lookup& u = luts[r];
    NEED(u.table && u.table_len >= 1 && u.table_len <= 65536, "table");
    NEED(u.last_lu_row < u.last_lut_row && u.last_lut_row <= u.first_lut_row && (uint64_t)u.first_lut_row + 1 < n, "lookup rows");
    NEED((uint64_t)(u.first_lut_row - u.last_lut_row + 1) * L.num_lut_slots >= u.table_len, "the table does not fit its LookupTableGate rows");
    total += (size_t)u.table_len * 2;
  }
  // from here on the old tables are gone: no captured graph and no LookupDev may point into them if a later step fails
  pr->drop_graph();
  pr->lookups = LookupDev{};
  pr->gates.num_lookup_selectors = 0;
  CK(pr->lut_tables.alloc(total * sizeof(uint16_t)));
  CK(pr->lut_eval.alloc((size_t)pr->Bcap * P.zs_count * MP2G_MAX_LUTS * sizeof(u64)));
  size_t off = 0;
  for (uint32_t r = 0; r < n_luts; r++) {
    const mp2g_lookup& u = luts[r];
    L.last_lu_row[r] = u.last_lu_row; L.last_lut_row[r] = u.last_lut_row; L.first_lut_row[r] = u.first_lut_row; L.table_len[r] = u.table_len;
    L.table[r] = (const uint16_t*)pr->lut_tables.p + off;
    CK(hipMemcpyAsync((uint16_t*)pr->lut_tables.p + off, u.table, (size_t)u.table_len * 2 * sizeof(uint16_t), hipMemcpyHostToDevice, pr->ctx->stream));
    off += (size_t)u.table_len * 2;
  }
  CK(hipStreamSynchronize(pr->ctx->stream));  // the caller's table memory may go away
  pr->lookups = L;
  pr->gates.num_lookup_selectors = n_sel;
  // the gate constants move behind the lookup selectors: re-validate the table against the constant count
  const char* msg = gate_table_check(pr->gates, num_constants, P.oracle_w[1]);
  if (msg) { pr->lookups = LookupDev{}; pr->gates.num_lookup_selectors = 0; return fail("invalid gate table with lookups: %s", msg); }
  pr->drop_graph();
  return 0;
}
int mp2g_prover_bind_public_inputs(mp2g_prover* pr, int64_t row) {
  NEED(pr, "prover");
  NEED(row < (int64_t)1 << pr->P.log_n, "row < 2^log_n");
  NEED(pr->P.n_oracles >= 2 && pr->P.oracle_w[1] >= 4, "needs a wires oracle of at least 4 columns");
  pr->pi_row = row < 0 ? -1 : row;
  pr->drop_graph();
  return 0;
}
int mp2g_prover_set_active(mp2g_prover* pr, uint32_t n) {
  NEED(pr && n >= 1 && n <= pr->Bcap, "1 <= n <= the batch the prover was created for");
  if (n != pr->B) pr->drop_graph();
  pr->B = n;
  return 0;
}
int mp2g_prover_enable_witness_check(mp2g_prover* pr, int on) {
  NEED(pr && pr->num_routed, "call mp2g_prover_enable_permutation first");
  if (on && !pr->wflags.p) CK(pr->wflags.alloc((size_t)pr->Bcap * sizeof(u32)));
  pr->wcheck = on != 0;
  pr->drop_graph();
  return 0;
}
int mp2g_prover_witness_check_enabled(const mp2g_prover* pr) { return pr && pr->wcheck ? 1 : 0; }
int mp2g_prover_witness_status(mp2g_prover* pr, uint32_t* flags) {
  NEED(pr && pr->wcheck, "call mp2g_prover_enable_witness_check first");
  std::vector<uint32_t> h(pr->B);
  CK(hipMemcpyAsync(h.data(), pr->wflags.p, pr->B * sizeof(u32), hipMemcpyDeviceToHost, pr->ctx->stream));
  CK(hipStreamSynchronize(pr->ctx->stream));
  if (flags) memcpy(flags, h.data(), pr->B * sizeof(u32));
  for (uint32_t b = 0; b < pr->B; b++)
    if (h[b])
      return fail("invalid witness: proof %u of the batch violates%s%s%s", b, (h[b] & 1) ? " a copy constraint" : "",
                  (h[b] & 2) ? " a gate constraint" : "", (h[b] & 4) ? " the lookup argument" : "");
  return 0;
}
}  // extern "C"
namespace mp2g { int prover_flags_to_host_async(mp2g_prover* pr, uint32_t* h_dst); }
int mp2g::prover_flags_to_host_async(mp2g_prover* pr, uint32_t* h_dst) {
  NEED(pr && pr->wcheck && h_dst, "prover with the witness check on / destination");
  CK(hipMemcpyAsync(h_dst, pr->wflags.p, pr->B * sizeof(u32), hipMemcpyDeviceToHost, pr->ctx->stream));
  return 0;
}
extern "C" {
int mp2g_prover_enable_timing(mp2g_prover* pr, int on) {
  NEED(pr, "prover");
  if (on)
    for (hipEvent_t& e : pr->ev)
      if (!e) CK(hipEventCreate(&e));
  pr->timing = on != 0;
  return 0;
}
int mp2g_prover_stage_ms(mp2g_prover* pr, float out[MP2G_N_STAGES]) {
  NEED(pr && out, "prover/out");
  NEED(pr->timing, "call mp2g_prover_enable_timing first");
  CK(hipStreamSynchronize(pr->ctx->stream));
  for (int i = 0; i < MP2G_N_STAGES; i++) CK(hipEventElapsedTime(&out[i], pr->ev[i], pr->ev[i + 1]));
  return 0;
}

int mp2g_partial_products_and_zs(mp2g_ctx* c, const uint64_t* wires, uint32_t wires_w, const uint64_t* sigmas, uint32_t log_n,
                                 uint32_t num_routed, uint32_t degree, const uint64_t* betas, const uint64_t* gammas, uint32_t nc,
                                 uint64_t* out) {
  NEED(c && wires && sigmas && betas && gammas && out, "ctx/pointers");
  NEED(log_n >= 1 && log_n <= 20 && nc >= 1 && nc <= 4, "log_n/nc");
  NEED(degree >= 1 && num_routed >= degree && num_routed % degree == 0 && num_routed / degree <= 16 && num_routed <= wires_w, "num_routed/degree");
  const size_t n = (size_t)1 << log_n, chunks = num_routed / degree;
  DevBuf dw, ds, dc, dq, dout;
  CK(dw.alloc(wires_w * n * sizeof(u64)));
  CK(ds.alloc(num_routed * n * sizeof(u64)));
  CK(dc.alloc(2 * nc * sizeof(u64)));
  CK(dq.alloc(nc * chunks * n * sizeof(u64)));
  CK(dout.alloc(nc * chunks * n * sizeof(u64)));
  CK(hipMemcpyAsync(dw.p, wires, wires_w * n * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(hipMemcpyAsync(ds.p, sigmas, num_routed * n * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(hipMemcpyAsync(dc.p, betas, nc * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(hipMemcpyAsync(dc.p + nc, gammas, nc * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(zpp_compute(c->stream, 1, dw.p, 0, ds.p, log_n, num_routed, degree, dc.p, dc.p + nc, 0, nc, dq.p, dout.p, 0));
  CK(hipMemcpyAsync(out, dout.p, nc * chunks * n * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}

uint32_t mp2g_gate_num_constraints(const mp2g_gate* g) { return g ? gate_num_constraints(*g) : 0; }
uint32_t mp2g_gate_degree(const mp2g_gate* g) { return g ? gate_degree(*g) : 0; }
int mp2g_eval_gate_constraints(mp2g_ctx* c, const mp2g_gate* gates, uint32_t n_gates, uint32_t num_selectors, const uint64_t* consts,
                               uint32_t num_constants, const uint64_t* wires, uint32_t wires_w, uint64_t npts,
                               const uint64_t pi_hash[4], uint64_t* out) {
  NEED(c && gates && consts && wires && pi_hash && out, "ctx/pointers");
  NEED(n_gates >= 1 && n_gates <= MP2G_MAX_GATES, "1..MP2G_MAX_GATES gates");
  NEED(npts >= 1 && npts <= ((uint64_t)1 << 28), "npts");
  GateTable t{};
  t.n_gates = n_gates; t.num_selectors = num_selectors;
  uint32_t max_j = 0;
  for (uint32_t i = 0; i < n_gates; i++) {
    t.g[i] = gates[i];
    uint32_t k = gate_num_constraints(gates[i]);
    if (k > max_j) max_j = k;
  }
  const char* msg = gate_table_check(t, num_constants, wires_w);
  if (msg) return fail("invalid gate table: %s", msg);
  if (!max_j) return 0;
  DevBuf dc, dw, dp, dout;
  CK(dc.alloc((size_t)num_constants * npts * sizeof(u64)));
  CK(dw.alloc((size_t)wires_w * npts * sizeof(u64)));
  CK(dp.alloc(4 * sizeof(u64)));
  CK(dout.alloc((size_t)max_j * npts * sizeof(u64)));
  CK(hipMemcpyAsync(dc.p, consts, (size_t)num_constants * npts * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(hipMemcpyAsync(dw.p, wires, (size_t)wires_w * npts * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(hipMemcpyAsync(dp.p, pi_hash, 4 * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(gate_constraints_points(c->stream, t, dc.p, dw.p, npts, max_j, dp.p, dout.p));
  CK(hipMemcpyAsync(out, dout.p, (size_t)max_j * npts * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- granular entry points over PolynomialBatch handles ----------------------------------------
int mp2g_batch_eval_ext(const mp2g_batch* b, const uint64_t point[2], uint64_t* out) {
  NEED(b && point && out, "batch/point/out");
  NEED(point[0] < GL_P && point[1] < GL_P, "point canonical");
  mp2g_ctx* c = b->ctx;
  FriShape sh{};
  sh.log_n = b->log_n; sh.rate_bits = b->rate_bits; sh.cap_h = b->cap_h; sh.n_oracles = 1; sh.n_polys = b->w;
  sh.o[0].coeffs = b->coeffs.p; sh.o[0].w = b->w;
  DevBuf dz, dout;
  CK(dz.alloc(2 * sizeof(u64)));
  CK(dout.alloc((size_t)b->w * 2 * sizeof(u64)));
  CK(hipMemcpyAsync(dz.p, point, 2 * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(fri_openings(c->stream, sh, 1, dz.p, 0, dout.p));
  CK(hipMemcpyAsync(out, dout.p, (size_t)b->w * 2 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}

int mp2g_fri_prove(mp2g_ctx* c, const mp2g_fri_params* params, mp2g_batch* const* oracles, const uint64_t zeta[2],
                   mp2g_challenger* ch, uint64_t* proof) {
  NEED(c && params && oracles && zeta && ch && proof, "ctx/pointers");
  NEED(ch->ctx == c && ch->count == 1, "challenger must live on this context with count 1");
  NEED(ch->variant == (int)params->variant, "challenger / params hash variant differ");
  mp2g_prover* pr;
  int rc = prover_create_impl(c, params, 1, false, &pr);
  if (rc) return rc;
  ProverGuard guard;
  guard.pr = pr;
  const mp2g_fri_params& P = pr->P;
  FriShape sh{};
  sh.log_n = P.log_n; sh.rate_bits = P.rate_bits; sh.cap_h = P.cap_height; sh.n_oracles = P.n_oracles;
  sh.zs_oracle = P.zs_oracle; sh.zs_count = P.zs_count; sh.lookup_count = P.zs_count * P.num_lookup_polys;
  for (uint32_t o = 0; o < P.n_oracles; o++) {
    const mp2g_batch* b = oracles[o];
    if (!b || b->ctx != c || b->log_n != P.log_n || b->w != P.oracle_w[o] || b->rate_bits != P.rate_bits ||
        b->cap_h != P.cap_height || b->variant != (int)P.variant) {
      return fail("oracle %u does not match the FRI parameters", o);
    }
    OracleRef& r = sh.o[o];
    r.coeffs = b->coeffs.p; r.values = b->values.p; r.levels = b->levels.p; r.w = b->w;
    sh.n_polys += r.w;
  }
  DevBuf dproof;
  hipError_t e = dproof.alloc(pr->proof_words * sizeof(u64));
  if (e == hipSuccess) e = hipMemcpyAsync(pr->zeta.p, zeta, 2 * sizeof(u64), hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) return fail("fri_prove setup: %s", hipGetErrorString(e));
  rc = bind_scratch(pr);
  if (!rc) rc = fri_tail(pr, sh, (ChState*)ch->st.p, dproof.p);
  if (!rc) {
    e = hipMemcpyAsync(proof, dproof.p, pr->proof_words * sizeof(u64), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail("fri_prove copy-out: %s", hipGetErrorString(e));
  }
  return rc;
}

int mp2g_pcs_prove(mp2g_ctx* c, const mp2g_fri_params* params, const uint64_t* const* values, const uint64_t circuit_digest[4],
                   const uint64_t pi_hash[4], uint64_t* caps, uint64_t* openings, uint64_t* proof) {
  NEED(c && values && circuit_digest && pi_hash && caps && openings && proof, "ctx/pointers");
  mp2g_prover* pr;
  int rc = mp2g_prover_create(c, params, 1, &pr);
  if (rc) return rc;
  ProverGuard guard;
  guard.pr = pr;
  const mp2g_fri_params& P = pr->P;
  const size_t n = (size_t)1 << P.log_n;
  DevBuf dv[8], dd, dp, dcaps, dopen, dproof;
  const uint64_t* dptr[8] = {};
  hipError_t e = hipSuccess;
  for (uint32_t o = 0; o < P.n_oracles && e == hipSuccess; o++) {
    e = dv[o].alloc(P.oracle_w[o] * n * sizeof(u64));
    if (e == hipSuccess) e = hipMemcpyAsync(dv[o].p, values[o], P.oracle_w[o] * n * sizeof(u64), hipMemcpyHostToDevice, c->stream);
    if (o) dptr[o - 1] = dv[o].p;
  }
  if (e == hipSuccess) e = dd.alloc(32);
  if (e == hipSuccess) e = dp.alloc(32);
  if (e == hipSuccess) e = dcaps.alloc(P.n_oracles * pr->capw * sizeof(u64));
  if (e == hipSuccess) e = dopen.alloc(pr->n_open * 2 * sizeof(u64));
  if (e == hipSuccess) e = dproof.alloc(pr->proof_words * sizeof(u64));
  if (e == hipSuccess) e = hipMemcpyAsync(dd.p, circuit_digest, 32, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dp.p, pi_hash, 32, hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) return fail("pcs_prove setup: %s", hipGetErrorString(e));
  rc = mp2g_prover_set_preprocessed_dev(pr, dv[0].p);
  if (!rc) rc = mp2g_prover_prove_dev(pr, dptr, dd.p, dp.p, dcaps.p, dopen.p, dproof.p);
  if (!rc) {
    e = hipMemcpyAsync(caps, dcaps.p, P.n_oracles * pr->capw * sizeof(u64), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(openings, dopen.p, pr->n_open * 2 * sizeof(u64), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(proof, dproof.p, pr->proof_words * sizeof(u64), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail("pcs_prove copy-out: %s", hipGetErrorString(e));
  }
  return rc;
}

}  // extern "C"
