/* Plain-C client of libmp2gpu: what a cgo / Rust-FFI binding sees. Commits a small witness matrix,
 * runs the fused PCS pipeline, serializes the proof in the reference's bincode layout and prints a
 * checksum that tests/test_gpu_c_abi.py compares with the Python harness.
 * build: gcc -std=c11 -Wall -Iinclude examples/c_abi_demo.c -Lmapreduce-plonky2_amd -lmp2gpu -o examples/c_abi_demo */
#include "mp2g.h"
#include <stdio.h>
#include <stdlib.h>

#define P 0xFFFFFFFF00000001ULL
static uint64_t sm_state;
static uint64_t splitmix(void) { /* SplitMix64 stream folded into the field, as tests/oracle.py rand_field */
  sm_state += 0x9E3779B97F4A7C15ULL;
  uint64_t z = sm_state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return z >= P ? z - P : z;
}
static uint64_t* field_matrix(uint64_t seed, size_t count) {
  uint64_t* m = malloc(count * sizeof *m);
  sm_state = seed;
  for (size_t i = 0; i < count; i++) m[i] = splitmix();
  return m;
}
#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "%s failed: %s\n", #x, mp2g_last_error()); return 1; } } while (0)

int main(void) {
  mp2g_ctx* ctx;
  CHECK(mp2g_ctx_create(0, &ctx));
  mp2g_fri_params fp = {0};
  fp.variant = MP2G_POSEIDON2; fp.log_n = 6; fp.rate_bits = 3; fp.cap_height = 4; fp.pow_bits = 6; fp.num_queries = 4;
  fp.n_layers = mp2g_reduction_arity_bits(fp.log_n, fp.rate_bits, fp.cap_height, 4, 5, fp.arity_bits);
  fp.n_oracles = 4;
  const uint32_t w[4] = {5, 9, 4, 3};
  const size_t n = (size_t)1 << fp.log_n;
  const uint64_t* values[4];
  for (int o = 0; o < 4; o++) { fp.oracle_w[o] = w[o]; values[o] = field_matrix(100 + o, w[o] * n); }
  fp.zs_oracle = 2; fp.zs_count = 2;
  uint64_t* cd = field_matrix(1, 4);
  uint64_t* ph = field_matrix(2, 4);
  size_t capw = (size_t)4 << fp.cap_height, n_open = mp2g_fri_n_openings(&fp), pw = mp2g_fri_proof_words(&fp);
  uint64_t* caps = malloc(4 * capw * 8);
  uint64_t* openings = malloc(n_open * 2 * 8);
  uint64_t* proof = malloc(pw * 8);
  CHECK(mp2g_pcs_prove(ctx, &fp, values, cd, ph, caps, openings, proof));
  uint64_t pis[3] = {7, 8, 9};
  size_t len = 0;
  CHECK(mp2g_proof_serialize(&fp, 2, caps, openings, proof, pis, 3, NULL, &len));
  uint8_t* bytes = malloc(len);
  CHECK(mp2g_proof_serialize(&fp, 2, caps, openings, proof, pis, 3, bytes, &len));
  uint64_t h = 1469598103934665603ULL; /* FNV-1a over the wire bytes */
  for (size_t i = 0; i < len; i++) { h ^= bytes[i]; h *= 1099511628211ULL; }
  printf("proof_words=%zu bytes=%zu fnv1a=%016llx pow_witness=%llu\n", pw, len, (unsigned long long)h,
         (unsigned long long)proof[pw - 1]);
  mp2g_ctx_destroy(ctx);
  return 0;
}
