/* Plain-C client of the batched prover: the complete prove() of a gate-level circuit through the C ABI alone
 * (what a Rust / cgo shim does after plonky2's witness generation). Reads a circuit + witness file written by
 * tests/test_gpu_c_abi.py, proves it on GPU 0 with the device-side witness check on, serializes the proof in the
 * reference's bincode layout and prints checksums for the test to compare with the Python harness.
 *
 * file layout (little endian): u32 log_n, num_constants, num_routed, wires_w, n_gates, num_selectors, pow_bits,
 * num_queries; n_gates x mp2g_gate (7 u32); u64 pi_hash[4], circuit_digest[4];
 * u64 preprocessed[(num_constants + num_routed) << log_n]; u64 wires[wires_w << log_n]
 * build: gcc -std=c11 -Wall -Iinclude examples/c_prove_circuit.c -Lmapreduce-plonky2_amd -lmp2gpu -o examples/c_prove_circuit */
#include "mp2g.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "%s failed: %s\n", #x, mp2g_last_error()); return 1; } } while (0)
#define READ(ptr, count) do { if (fread((ptr), sizeof *(ptr), (count), f) != (size_t)(count)) { fprintf(stderr, "short read\n"); return 1; } } while (0)

static uint64_t fnv1a(const void* p, size_t len) {
  const uint8_t* b = p;
  uint64_t h = 1469598103934665603ULL;
  for (size_t i = 0; i < len; i++) { h ^= b[i]; h *= 1099511628211ULL; }
  return h;
}

int main(int argc, char** argv) {
  if (argc != 2) { fprintf(stderr, "usage: %s circuit.bin\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  uint32_t hdr[8];
  READ(hdr, 8);
  const uint32_t log_n = hdr[0], num_constants = hdr[1], num_routed = hdr[2], wires_w = hdr[3], n_gates = hdr[4], num_selectors = hdr[5];
  const size_t n = (size_t)1 << log_n;
  mp2g_gate gates[MP2G_MAX_GATES];
  if (n_gates > MP2G_MAX_GATES) return 2;
  READ(gates, n_gates);
  uint64_t pi_hash[4], digest[4];
  READ(pi_hash, 4);
  READ(digest, 4);
  const size_t pre_words = (size_t)(num_constants + num_routed) * n, wire_words = (size_t)wires_w * n;
  uint64_t* pre = malloc(pre_words * 8);
  uint64_t* wires = malloc(wire_words * 8);
  READ(pre, pre_words);
  READ(wires, wire_words);
  fclose(f);

  /* standard_recursion_config shape: 2 challenges, quotient degree factor 8, rate 1/8, cap height 4, arity 16 */
  mp2g_fri_params fp;
  memset(&fp, 0, sizeof fp);
  fp.variant = MP2G_POSEIDON2; fp.log_n = log_n; fp.rate_bits = 3; fp.cap_height = 4; fp.pow_bits = hdr[6]; fp.num_queries = hdr[7];
  fp.n_layers = mp2g_reduction_arity_bits(log_n, fp.rate_bits, fp.cap_height, 4, 5, fp.arity_bits);
  fp.n_oracles = 4;
  fp.oracle_w[0] = num_constants + num_routed; fp.oracle_w[1] = wires_w; fp.oracle_w[2] = 2 * (num_routed / 8); fp.oracle_w[3] = 16;
  fp.zs_oracle = 2; fp.zs_count = 2;
  const size_t capw = (size_t)4 << fp.cap_height, n_open = mp2g_fri_n_openings(&fp), pw = mp2g_fri_proof_words(&fp);

  mp2g_ctx* ctx;
  CHECK(mp2g_ctx_create(0, &ctx));
  void *d_pre, *d_wires, *d_pi, *d_cd, *d_caps, *d_open, *d_proof;
  CHECK(mp2g_dev_alloc(ctx, pre_words * 8, &d_pre));
  CHECK(mp2g_dev_alloc(ctx, wire_words * 8, &d_wires));
  CHECK(mp2g_dev_alloc(ctx, 32, &d_pi));
  CHECK(mp2g_dev_alloc(ctx, 32, &d_cd));
  CHECK(mp2g_dev_alloc(ctx, 4 * capw * 8, &d_caps));
  CHECK(mp2g_dev_alloc(ctx, n_open * 16, &d_open));
  CHECK(mp2g_dev_alloc(ctx, pw * 8, &d_proof));
  CHECK(mp2g_h2d(ctx, d_pre, pre, pre_words * 8));
  CHECK(mp2g_h2d(ctx, d_wires, wires, wire_words * 8));
  CHECK(mp2g_h2d(ctx, d_pi, pi_hash, 32));
  CHECK(mp2g_h2d(ctx, d_cd, digest, 32));

  mp2g_prover* pr;
  CHECK(mp2g_prover_create(ctx, &fp, 1, &pr));
  CHECK(mp2g_prover_set_preprocessed_dev(pr, d_pre));          /* CircuitData: constants + sigmas, committed once */
  CHECK(mp2g_prover_enable_permutation(pr, num_routed, 8));    /* Z / partial products on the device */
  CHECK(mp2g_prover_enable_quotient(pr));                      /* quotient polynomials on the device ... */
  CHECK(mp2g_prover_set_gates(pr, gates, n_gates, num_selectors)); /* ... with the circuit's gate constraints */
  CHECK(mp2g_prover_enable_witness_check(pr, 1));              /* prove() fails on an unsatisfied witness */
  const uint64_t* d_values[3] = {d_wires, NULL, NULL};
  CHECK(mp2g_prover_prove_dev(pr, d_values, d_cd, d_pi, d_caps, d_open, d_proof));
  uint32_t flags = 0;
  int bad = mp2g_prover_witness_status(pr, &flags);
  printf("witness_flags=%u%s%s\n", flags, bad ? " : " : "", bad ? mp2g_last_error() : "");

  uint64_t* caps = malloc(4 * capw * 8);
  uint64_t* openings = malloc(n_open * 16);
  uint64_t* proof = malloc(pw * 8);
  CHECK(mp2g_d2h(ctx, caps, d_caps, 4 * capw * 8));
  CHECK(mp2g_d2h(ctx, openings, d_open, n_open * 16));
  CHECK(mp2g_d2h(ctx, proof, d_proof, pw * 8));
  size_t len = 0;
  CHECK(mp2g_proof_serialize(&fp, num_constants, caps, openings, proof, pi_hash, 4, NULL, &len));
  uint8_t* bytes = malloc(len);
  CHECK(mp2g_proof_serialize(&fp, num_constants, caps, openings, proof, pi_hash, 4, bytes, &len));
  printf("proof_words=%zu bytes=%zu proof_fnv1a=%016llx wire_fnv1a=%016llx\n", pw, len,
         (unsigned long long)fnv1a(proof, pw * 8), (unsigned long long)fnv1a(bytes, len));
  mp2g_prover_free(pr);
  mp2g_ctx_destroy(ctx);
  return bad ? 3 : 0;
}
