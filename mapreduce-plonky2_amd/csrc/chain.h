// The chain of csrc/chain.hip (generate_proof = base prove() + wrap steps for a batch) as the other host-side sources see it:
// forest.hip fills the staging buffer itself and queues its own device copies between the upload and the first witness replay.
#pragma once
#include "ctx.h"
#include "witness.h"

struct mp2g_chain {
  mp2g_ctx* ctx = nullptr;
  uint32_t cap = 0, last_batch = 0;
  struct Step {
    mp2g_prover* pr = nullptr;
    mp2g_witness_program* prog = nullptr;
    const u64* d_digest = nullptr;
    mp2g_fri_params P{};
    size_t n_in = 0, n_probe = 0, cap_words = 0, n_open = 0, proof_words = 0;
    mp2g::DevBuf in, wires, probe, pi_hash, caps, openings, proof;
  };
  Step steps[8];  // DevBuf owns device memory and does not move
  uint32_t n_steps = 0;
  // pinned staging for the inputs on the way up and the last step's outputs on the way down (pageable copies would be staged by
  // the runtime in small synchronous pieces)
  u64* h_in = nullptr;
  u64* h_out = nullptr;
  // a caller that keeps the stream full (forest.hip) fills one input buffer while the upload of the other is still queued: the second
  // buffer (made on first use), and per buffer the event its last upload recorded
  u64* h_in_alt = nullptr;
  hipEvent_t in_ev[2] = {nullptr, nullptr};
  bool in_ev_set[2] = {false, false};
  uint32_t in_flip = 0;
  ~mp2g_chain() {
    if (h_in) (void)hipHostFree(h_in);
    if (h_in_alt) (void)hipHostFree(h_in_alt);
    if (h_out) (void)hipHostFree(h_out);
    for (hipEvent_t e : in_ev) if (e) (void)hipEventDestroy(e);
  }
};


namespace mp2g {
// mp2g_chain_run with the inputs already in ch->h_in ([batch][n_in of step 0]): upload, `between(stream)` (device-side patches of
// the inputs: child proofs that live on the device), every step, `after(stream)` (device-side consumers of the last step's outputs),
// one synchronisation, the witness status of every step. host_out: the four host pointers of mp2g_chain_run or all null.
struct ChainHooks {
  void* user = nullptr;
  int (*between)(void* user, mp2g_chain* ch, hipStream_t s) = nullptr;
  int (*after)(void* user, mp2g_chain* ch, hipStream_t s) = nullptr;
};
int chain_run_staged(mp2g_chain* ch, uint32_t batch, const mp2g_chain_patch* patches, uint32_t n_patches, const ChainHooks* hooks,
                     uint64_t* caps, uint64_t* openings, uint64_t* proof, uint64_t* public_inputs);
// The same launch sequence WITHOUT the synchronisation, for a caller that queues the next batch while this one runs: inputs from the
// pinned buffer `which` (0 = h_in, 1 = h_in_alt; chain_input_buffer waits until the buffer's previous upload has left it), and the
// witness-check flags of every step copied in stream order into h_flags [n_steps][cap] (pinned; a step without the check leaves its
// row alone) -- the caller reads them once an event it records behind this call has completed (chain_flags_check).
int chain_input_buffer(mp2g_chain* ch, uint32_t which, u64** out);
int chain_enqueue(mp2g_chain* ch, uint32_t batch, uint32_t which, const ChainHooks* hooks, uint32_t* h_flags);
int chain_flags_check(const mp2g_chain* ch, uint32_t batch, const uint32_t* h_flags);
// prover.hip: the witness-check flags of the prover's last prove, [active batch] words, copied to pinned host memory in stream order
int prover_flags_to_host_async(mp2g_prover* pr, uint32_t* h_dst);
}  // namespace mp2g
