/* Plain-C client of mp2g_chain: CircuitWithUniversalVerifier::generate_proof (recursion-framework/src/circuit_builder.rs:286-311) --
 * witness generation, base prove(), the wrap chain down to the framework's shared 2^12-row shape -- for a batch of nodes of ONE
 * framework circuit, through the C ABI alone: what a Rust / cgo shim of the reference calls per tree level. Reads the circuit's
 * steps (CircuitData + recorded witness program each) and a batch of witness inputs from a file written by
 * tests/test_gpu_c_abi.py, proves on GPU 0 with the witness check on, prints checksums of the final proofs and public inputs.
 *
 * file layout (little endian): u32 n_steps, batch, n_inputs0; per step: u32 hdr[12] = log_n, num_constants, n_gates, num_selectors,
 * pow_bits, num_queries, n_slots, n_inputs, n_consts, n_probe, tape_len_lo, tape_len_hi; n_gates x mp2g_gate (7 u32);
 * u64 circuit_digest[4]; u64 preprocessed[(num_constants + 80) << log_n]; u64 tape[tape_len]; u32 input_sids[n_inputs];
 * u64 const_slots[2 * n_consts]; u32 probe_sids[n_probe]; then u64 inputs[batch][n_inputs0]
 * build: gcc -std=c11 -Wall -Iinclude examples/c_generate_proof.c -Lmapreduce-plonky2_amd -lmp2gpu -o examples/c_generate_proof */
#include "mp2g.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "%s failed: %s\n", #x, mp2g_last_error()); return 1; } } while (0)
#define READ(ptr, count) do { if (fread((ptr), sizeof *(ptr), (count), f) != (size_t)(count)) { fprintf(stderr, "short read\n"); return 1; } } while (0)
#define MAX_STEPS 4
#define NUM_ROUTED 80
#define NUM_WIRES 135

static uint64_t fnv1a(const void* p, size_t len) {
  const uint8_t* b = p;
  uint64_t h = 1469598103934665603ULL;
  for (size_t i = 0; i < len; i++) { h ^= b[i]; h *= 1099511628211ULL; }
  return h;
}

int main(int argc, char** argv) {
  if (argc != 2) { fprintf(stderr, "usage: %s framework_circuit.bin\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  uint32_t top[3];
  READ(top, 3);
  const uint32_t n_steps = top[0], batch = top[1], n_inputs0 = top[2];
  if (n_steps < 1 || n_steps > MAX_STEPS) return 2;
  mp2g_ctx* ctx;
  CHECK(mp2g_ctx_create(0, &ctx));
  mp2g_prover* provers[MAX_STEPS];
  mp2g_witness_program* programs[MAX_STEPS];
  mp2g_fri_params fps[MAX_STEPS];
  const uint64_t* d_digests[MAX_STEPS];
  uint32_t n_probe_last = 0;
  for (uint32_t s = 0; s < n_steps; s++) {
    uint32_t h[12];
    READ(h, 12);
    const uint32_t log_n = h[0], num_constants = h[1], n_gates = h[2], num_selectors = h[3], n_slots = h[6], n_inputs = h[7], n_consts = h[8], n_probe = h[9];
    const size_t tape_len = (size_t)h[10] | ((size_t)h[11] << 32), n = (size_t)1 << log_n, pre_words = (size_t)(num_constants + NUM_ROUTED) * n;
    mp2g_gate gates[MP2G_MAX_GATES];
    if (n_gates > MP2G_MAX_GATES) return 2;
    READ(gates, n_gates);
    uint64_t digest[4];
    READ(digest, 4);
    uint64_t* pre = malloc(pre_words * 8);
    uint64_t* tape = malloc(tape_len * 8 + 8);
    uint32_t* input_sids = malloc((size_t)n_inputs * 4 + 4);
    uint64_t* consts = malloc((size_t)n_consts * 16 + 8);
    uint32_t* probe = malloc((size_t)n_probe * 4 + 4);
    READ(pre, pre_words); READ(tape, tape_len); READ(input_sids, n_inputs); READ(consts, 2 * (size_t)n_consts); READ(probe, n_probe);
    /* standard_recursion_config (mp2-common/src/lib.rs:45-47) */
    mp2g_fri_params* fp = &fps[s];
    memset(fp, 0, sizeof *fp);
    fp->variant = MP2G_POSEIDON2; fp->log_n = log_n; fp->rate_bits = 3; fp->cap_height = 4; fp->pow_bits = h[4]; fp->num_queries = h[5];
    fp->n_layers = mp2g_reduction_arity_bits(log_n, fp->rate_bits, fp->cap_height, 4, 5, fp->arity_bits);
    fp->n_oracles = 4;
    fp->oracle_w[0] = num_constants + NUM_ROUTED; fp->oracle_w[1] = NUM_WIRES; fp->oracle_w[2] = 2 * (NUM_ROUTED / 8); fp->oracle_w[3] = 16;
    fp->zs_oracle = 2; fp->zs_count = 2;
    void *d_pre, *d_cd;
    CHECK(mp2g_dev_alloc(ctx, pre_words * 8, &d_pre));
    CHECK(mp2g_dev_alloc(ctx, 32, &d_cd));
    CHECK(mp2g_h2d(ctx, d_pre, pre, pre_words * 8));
    CHECK(mp2g_h2d(ctx, d_cd, digest, 32));
    d_digests[s] = d_cd;
    /* CircuitData of the step: the prover with its gate table; prove() fails on an unsatisfied witness */
    CHECK(mp2g_prover_create(ctx, fp, batch, &provers[s]));
    CHECK(mp2g_prover_set_preprocessed_dev(provers[s], d_pre));
    CHECK(mp2g_prover_enable_permutation(provers[s], NUM_ROUTED, 8));
    CHECK(mp2g_prover_enable_quotient(provers[s]));
    CHECK(mp2g_prover_set_gates(provers[s], gates, n_gates, num_selectors));
    CHECK(mp2g_prover_enable_witness_check(provers[s], 1));
    /* its witness generator: the builder's recorded program; the probe = public-inputs hash, then the public inputs */
    CHECK(mp2g_witness_program_create(tape, tape_len, n_slots, log_n, input_sids, n_inputs, consts, n_consts, &programs[s]));
    CHECK(mp2g_witness_program_set_probe(programs[s], probe, n_probe));
    n_probe_last = n_probe;
    free(pre); free(tape); free(input_sids); free(consts); free(probe);
  }
  uint64_t* inputs = malloc((size_t)batch * n_inputs0 * 8);
  READ(inputs, (size_t)batch * n_inputs0);
  fclose(f);

  mp2g_chain* chain;
  CHECK(mp2g_chain_create(ctx, n_steps, provers, programs, fps, d_digests, batch, &chain));
  const mp2g_fri_params* lp = &fps[n_steps - 1];
  const size_t capw = (size_t)4 << lp->cap_height, n_open = mp2g_fri_n_openings(lp), pw = mp2g_fri_proof_words(lp), n_pi = n_probe_last - 4;
  uint64_t* caps = malloc((size_t)batch * 4 * capw * 8);
  uint64_t* openings = malloc((size_t)batch * n_open * 16);
  uint64_t* proof = malloc((size_t)batch * pw * 8);
  uint64_t* pis = malloc((size_t)batch * n_pi * 8);
  int rc = mp2g_chain_run(chain, inputs, batch, NULL, 0, caps, openings, proof, pis);  /* generate_proof x batch */
  if (rc) { printf("generate_proof failed: %s\n", mp2g_last_error()); return 3; }
  for (uint32_t b = 0; b < batch; b++)
    printf("node %u: proof_fnv1a=%016llx openings_fnv1a=%016llx caps_fnv1a=%016llx pis_fnv1a=%016llx\n", b,
           (unsigned long long)fnv1a(proof + (size_t)b * pw, pw * 8), (unsigned long long)fnv1a(openings + (size_t)b * n_open * 2, n_open * 16),
           (unsigned long long)fnv1a(caps + (size_t)b * 4 * capw, 4 * capw * 8), (unsigned long long)fnv1a(pis + (size_t)b * n_pi, n_pi * 8));
  mp2g_chain_free(chain);
  for (uint32_t s = 0; s < n_steps; s++) { mp2g_prover_free(provers[s]); mp2g_witness_program_free(programs[s]); }
  mp2g_ctx_destroy(ctx);
  return 0;
}
