// generate_proof for a batch, host side in C++ behind the C ABI (include/mp2g.h, mp2g_chain_*).
//
// Replaces the bodies of CircuitWithUniversalVerifier::generate_proof (recursion-framework/src/circuit_builder.rs:286-311:
// set the witness, prove the base circuit, hand the proof to the wrap circuit) and WrapCircuit::wrap_proof
// (universal_verifier_gadget/wrap_circuit.rs:122-148: for every wrap step set the previous proof as witness and prove) for
// `batch` nodes of one framework circuit at once, without the host between the steps: per step the circuit's witness program is
// replayed on the device (mp2g_witness_program_run_dev) into the step's wire matrix, its probe gives the public-inputs hash,
// mp2g_prover_prove_dev follows on the same stream, and the next step's witness inputs -- the proof's public inputs, the three
// proof caps, openings and FRI proof words, plonky2's ProofWithPublicInputsTarget order -- are gathered from the prover's outputs
// by four strided device copies. Child proofs that already live on the device (a previous chain's outputs, or tensors received
// from another rank) are copied into the base step's inputs in place (mp2g_chain_patch). One synchronisation at the end.
#include "chain.h"
#include <cstring>
#include <new>

using namespace mp2g;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail("%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define NEED(c, msg) do { if (!(c)) return fail("invalid argument: %s", msg); } while (0)

extern "C" {
int mp2g_chain_create(mp2g_ctx* c, uint32_t n_steps, mp2g_prover* const* provers, mp2g_witness_program* const* programs,
                      const mp2g_fri_params* params, const uint64_t* const* d_circuit_digests, uint32_t capacity, mp2g_chain** out) {
  NEED(c && out && provers && programs && params && d_circuit_digests && n_steps >= 1 && n_steps <= 8 && capacity >= 1, "ctx / steps / capacity");
  mp2g_chain* ch = new (std::nothrow) mp2g_chain();
  if (!ch) return fail("out of memory");
  ch->ctx = c; ch->cap = capacity;
  ch->n_steps = n_steps;
  hipError_t e = hipSuccess;
  for (uint32_t s = 0; s < n_steps; s++) {
    mp2g_chain::Step& st = ch->steps[s];
    if (!provers[s] || !programs[s] || !d_circuit_digests[s]) { delete ch; return fail("invalid argument: step %u", s); }
    st.pr = provers[s]; st.prog = programs[s]; st.d_digest = d_circuit_digests[s]; st.P = params[s];
    const mp2g_fri_params& P = st.P;
    if (P.log_n != st.prog->log_n || P.n_oracles != 4 || P.oracle_w[1] != NUM_WIRES) { delete ch; return fail("invalid argument: step %u: the program and the FRI parameters describe different circuits", s); }
    st.n_in = st.prog->input_sids.size();
    st.n_probe = st.prog->probe.size();
    if (st.n_probe < 4) { delete ch; return fail("invalid argument: step %u: set the program's probe (public-inputs hash, public inputs) first", s); }
    st.cap_words = (size_t)4 << P.cap_height;
    st.n_open = mp2g_fri_n_openings(&P);
    st.proof_words = mp2g_fri_proof_words(&P);
    if (s > 0) {
      const mp2g_chain::Step& pv = ch->steps[s - 1];
      const size_t want = (pv.n_probe - 4) + 3 * pv.cap_words + 2 * pv.n_open + pv.proof_words;
      if (st.n_in != want) { delete ch; return fail("invalid argument: step %u takes %zu inputs, a proof of step %u has %zu words", s, st.n_in, s - 1, want); }
    }
    const size_t B = capacity;
    auto A = [&](DevBuf& d, size_t words) { if (e == hipSuccess) e = d.alloc(words * sizeof(u64)); };
    A(st.in, B * st.n_in); A(st.wires, (B * NUM_WIRES) << P.log_n); A(st.probe, B * st.n_probe); A(st.pi_hash, B * 4);
    A(st.caps, B * P.n_oracles * st.cap_words); A(st.openings, B * st.n_open * 2); A(st.proof, B * st.proof_words);
  }
  if (e == hipSuccess) e = hipHostMalloc((void**)&ch->h_in, (size_t)capacity * ch->steps[0].n_in * sizeof(u64), hipHostMallocDefault);
  if (e == hipSuccess) {
    const mp2g_chain::Step& L = ch->steps[n_steps - 1];
    e = hipHostMalloc((void**)&ch->h_out, (size_t)capacity * (L.P.n_oracles * L.cap_words + 2 * L.n_open + L.proof_words + L.n_probe) * sizeof(u64), hipHostMallocDefault);
  }
  if (e != hipSuccess) { delete ch; return fail("chain_create: %s", hipGetErrorString(e)); }
  *out = ch;
  return 0;
}

int mp2g_chain_run(mp2g_chain* ch, const uint64_t* inputs, uint32_t batch, const mp2g_chain_patch* patches, uint32_t n_patches,
                   uint64_t* caps, uint64_t* openings, uint64_t* proof, uint64_t* public_inputs) {
  NEED(ch && inputs && batch >= 1 && batch <= ch->cap && (patches || !n_patches), "chain / inputs / batch <= capacity");
  // (a caller that mixes the two entry points: an upload of h_in queued by chain_enqueue must have left the buffer BEFORE it is overwritten)
  if (ch->in_ev_set[0]) CK(hipEventSynchronize(ch->in_ev[0]));
  memcpy(ch->h_in, inputs, (size_t)batch * ch->steps[0].n_in * sizeof(u64));
  return chain_run_staged(ch, batch, patches, n_patches, nullptr, caps, openings, proof, public_inputs);
}
}  // extern "C"

// the launch sequence of a batch: upload, patches, `between`, every step (witness replay, prove, hand-over to the next step);
// h_flags: the steps' witness-check flags go to pinned host memory behind their prove() (chain_enqueue)
static int chain_steps(mp2g_chain* ch, uint32_t batch, const u64* h_in, const mp2g_chain_patch* patches, uint32_t n_patches, const ChainHooks* hooks,
                       uint32_t* h_flags, hipEvent_t uploaded) {
  hipStream_t s = ch->ctx->stream;
  mp2g_chain::Step& s0 = ch->steps[0];
  CK(hipMemcpyAsync(s0.in.p, h_in, (size_t)batch * s0.n_in * sizeof(u64), hipMemcpyHostToDevice, s));
  if (uploaded) CK(hipEventRecord(uploaded, s));
  for (uint32_t i = 0; i < n_patches; i++)
    CK(hipMemcpyAsync(s0.in.p + (size_t)patches[i].job * s0.n_in + patches[i].offset, patches[i].d_src, (size_t)patches[i].n_words * sizeof(u64),
                      hipMemcpyDeviceToDevice, s));
  if (hooks && hooks->between) { int rc = hooks->between(hooks->user, ch, s); if (rc) return rc; }
  for (size_t k = 0; k < ch->n_steps; k++) {
    mp2g_chain::Step& st = ch->steps[k];
    if (k > 0) {  // the previous proof becomes this step's witness inputs: public inputs, caps of oracles 1..3, openings, FRI proof words
      mp2g_chain::Step& pv = ch->steps[k - 1];
      const size_t n_pi = pv.n_probe - 4, cw = 3 * pv.cap_words, ow = 2 * pv.n_open, pw = pv.proof_words, pitch = st.n_in * 8;
      CK(hipMemcpy2DAsync(st.in.p, pitch, pv.probe.p + 4, pv.n_probe * 8, n_pi * 8, batch, hipMemcpyDeviceToDevice, s));
      CK(hipMemcpy2DAsync(st.in.p + n_pi, pitch, pv.caps.p + pv.cap_words, pv.P.n_oracles * pv.cap_words * 8, cw * 8, batch, hipMemcpyDeviceToDevice, s));
      CK(hipMemcpy2DAsync(st.in.p + n_pi + cw, pitch, pv.openings.p, ow * 8, ow * 8, batch, hipMemcpyDeviceToDevice, s));
      CK(hipMemcpy2DAsync(st.in.p + n_pi + cw + ow, pitch, pv.proof.p, pw * 8, pw * 8, batch, hipMemcpyDeviceToDevice, s));
    }
    int rc = mp2g_prover_set_active(st.pr, batch);
    if (rc) return rc;
    rc = mp2g_witness_program_run_dev(st.prog, ch->ctx, st.in.p, batch, st.wires.p, st.probe.p);
    if (rc) return rc;
    CK(hipMemcpy2DAsync(st.pi_hash.p, 32, st.probe.p, st.n_probe * 8, 32, batch, hipMemcpyDeviceToDevice, s));
    const uint64_t* vals[3] = {st.wires.p, nullptr, nullptr};
    rc = mp2g_prover_prove_dev(st.pr, vals, st.d_digest, st.pi_hash.p, st.caps.p, st.openings.p, st.proof.p);
    if (rc) return rc;
    if (h_flags && mp2g_prover_witness_check_enabled(st.pr)) {
      rc = prover_flags_to_host_async(st.pr, h_flags + k * ch->cap);
      if (rc) return rc;
    }
  }
  ch->last_batch = batch;
  return 0;
}

int mp2g::chain_input_buffer(mp2g_chain* ch, uint32_t which, u64** out) {
  NEED(ch && out && which < 2, "chain / buffer");
  if (which == 1 && !ch->h_in_alt)
    CK(hipHostMalloc((void**)&ch->h_in_alt, (size_t)ch->cap * ch->steps[0].n_in * sizeof(u64), hipHostMallocDefault));
  if (!ch->in_ev[which]) CK(hipEventCreateWithFlags(&ch->in_ev[which], hipEventDisableTiming));
  if (ch->in_ev_set[which]) CK(hipEventSynchronize(ch->in_ev[which]));  // the buffer's previous upload has left it
  *out = which ? ch->h_in_alt : ch->h_in;
  return 0;
}

int mp2g::chain_enqueue(mp2g_chain* ch, uint32_t batch, uint32_t which, const ChainHooks* hooks, uint32_t* h_flags) {
  NEED(ch && batch >= 1 && batch <= ch->cap && which < 2 && ch->in_ev[which] && (which == 0 || ch->h_in_alt), "chain / batch <= capacity / buffer from chain_input_buffer");
  int rc = chain_steps(ch, batch, which ? ch->h_in_alt : ch->h_in, nullptr, 0, hooks, h_flags, ch->in_ev[which]);
  ch->in_ev_set[which] = true;  // (also after a failure half way: waiting on a recorded event is harmless, on an unrecorded one immediate)
  if (rc) return rc;
  if (hooks && hooks->after) { rc = hooks->after(hooks->user, ch, ch->ctx->stream); if (rc) return rc; }
  return 0;
}

int mp2g::chain_flags_check(const mp2g_chain* ch, uint32_t batch, const uint32_t* h_flags) {
  for (size_t k = 0; k < ch->n_steps; k++) {
    if (!mp2g_prover_witness_check_enabled(ch->steps[k].pr)) continue;
    for (uint32_t b = 0; b < batch; b++) {
      const uint32_t h = h_flags[k * ch->cap + b];
      if (h)
        return fail("invalid witness: proof %u of the batch (step %zu of its chain) violates%s%s%s", b, k, (h & 1) ? " a copy constraint" : "",
                    (h & 2) ? " a gate constraint" : "", (h & 4) ? " the lookup argument" : "");
    }
  }
  return 0;
}

int mp2g::chain_run_staged(mp2g_chain* ch, uint32_t batch, const mp2g_chain_patch* patches, uint32_t n_patches, const ChainHooks* hooks,
                           uint64_t* caps, uint64_t* openings, uint64_t* proof, uint64_t* public_inputs) {
  NEED(ch && batch >= 1 && batch <= ch->cap && (patches || !n_patches), "chain / batch <= capacity");
  hipStream_t s = ch->ctx->stream;
  mp2g_chain::Step& s0 = ch->steps[0];
  for (uint32_t i = 0; i < n_patches; i++)
    NEED(patches[i].job < batch && patches[i].d_src && (size_t)patches[i].offset + patches[i].n_words <= s0.n_in, "patch outside the inputs");
  if (ch->in_ev_set[0]) CK(hipEventSynchronize(ch->in_ev[0]));  // (callers that fill h_in themselves, e.g. the forest's synchronous mode: the queued upload first)
  {
    int rc = chain_steps(ch, batch, ch->h_in, patches, n_patches, hooks, nullptr, nullptr);
    if (rc) return rc;
  }
  mp2g_chain::Step& L = ch->steps[ch->n_steps - 1];
  const size_t n_caps = (size_t)batch * L.P.n_oracles * L.cap_words, n_op = (size_t)batch * L.n_open * 2, n_pf = (size_t)batch * L.proof_words,
               n_pi = L.n_probe - 4;
  u64* hc = ch->h_out;
  u64* ho = hc + n_caps;
  u64* hp = ho + n_op;
  u64* hi = hp + n_pf;
  if (caps) CK(hipMemcpyAsync(hc, L.caps.p, n_caps * 8, hipMemcpyDeviceToHost, s));
  if (openings) CK(hipMemcpyAsync(ho, L.openings.p, n_op * 8, hipMemcpyDeviceToHost, s));
  if (proof) CK(hipMemcpyAsync(hp, L.proof.p, n_pf * 8, hipMemcpyDeviceToHost, s));
  if (public_inputs) CK(hipMemcpy2DAsync(hi, n_pi * 8, L.probe.p + 4, L.n_probe * 8, n_pi * 8, batch, hipMemcpyDeviceToHost, s));
  if (hooks && hooks->after) { int rc = hooks->after(hooks->user, ch, s); if (rc) return rc; }
  CK(hipStreamSynchronize(s));
  if (caps) memcpy(caps, hc, n_caps * 8);
  if (openings) memcpy(openings, ho, n_op * 8);
  if (proof) memcpy(proof, hp, n_pf * 8);
  if (public_inputs) memcpy(public_inputs, hi, (size_t)batch * n_pi * 8);
  // plonky2's prove() panics on a witness that violates a constraint: with the provers' witness check on, so does the chain
  for (size_t k = 0; k < ch->n_steps; k++)
    if (mp2g_prover_witness_check_enabled(ch->steps[k].pr)) {
      int rc = mp2g_prover_witness_status(ch->steps[k].pr, nullptr);
      if (rc) return rc;
    }
  return 0;
}

extern "C" {
int mp2g_chain_step_buffers(const mp2g_chain* ch, uint32_t step, uint64_t** d_wires, uint64_t** d_probe, uint64_t** d_caps, uint64_t** d_openings,
                            uint64_t** d_proof) {
  NEED(ch && step < ch->n_steps, "chain / step");
  const mp2g_chain::Step& st = ch->steps[step];
  if (d_wires) *d_wires = st.wires.p;
  if (d_probe) *d_probe = st.probe.p;
  if (d_caps) *d_caps = st.caps.p;
  if (d_openings) *d_openings = st.openings.p;
  if (d_proof) *d_proof = st.proof.p;
  return 0;
}

int mp2g_chain_device_proof(const mp2g_chain* ch, uint32_t b, const uint64_t* d_parts[4], uint32_t n_words[4]) {
  NEED(ch && d_parts && n_words && b < ch->last_batch, "chain / proof index of the last run");
  const mp2g_chain::Step& L = ch->steps[ch->n_steps - 1];
  d_parts[0] = L.probe.p + (size_t)b * L.n_probe + 4;                      n_words[0] = (uint32_t)(L.n_probe - 4);
  d_parts[1] = L.caps.p + ((size_t)b * L.P.n_oracles + 1) * L.cap_words;   n_words[1] = (uint32_t)(3 * L.cap_words);
  d_parts[2] = L.openings.p + (size_t)b * L.n_open * 2;                    n_words[2] = (uint32_t)(2 * L.n_open);
  d_parts[3] = L.proof.p + (size_t)b * L.proof_words;                      n_words[3] = (uint32_t)L.proof_words;
  return 0;
}

void mp2g_chain_free(mp2g_chain* ch) {
  if (!ch) return;
  (void)hipStreamSynchronize(ch->ctx->stream);
  delete ch;
}
}  // extern "C"
