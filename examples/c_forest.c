/* Plain-C client of mp2g_forest: a whole tree of framework proofs -- the map / reduce tree of recursion-framework/tests/integration.rs
 * (map circuit at the leaves, 2-to-1 reduce circuit with two universal verifiers above them) -- proved bottom-up through the C ABI
 * alone: circuits described once, nodes registered once (circuit, children, the words of their witness inputs that are not child
 * proofs), then one mp2g_forest_prove call per wave of units; two worker threads inside the library (one mp2g_ctx = HIP stream and
 * one mp2g_chain per circuit each), child proofs in the forest's device pool. What the reference's harness does node by node with
 * RecursiveCircuits::generate_proof (mp2-v1/tests/common/celltree.rs:54-189, rowtree.rs:78-337). Prints a checksum of the root's proof.
 *
 * file layout (little endian), written by tests/test_gpu_c_abi.py: u32 n_circuits, n_workers, capacity, slot_words, pool_slots;
 * per circuit: u32 n_steps, then per step the block of examples/c_generate_proof.c (u32 hdr[12]; gates; u64 digest[4]; preprocessed;
 * tape; input_sids; const_slots; probe_sids), then the forest descriptor u32 n_inputs, n_children, child_offset[4], n_const;
 * per circuit: u32 count; u64 ids[count]; u64 child_ids[count][n_children]; u64 consts[count][n_const];
 * u32 n_waves; per wave: u32 n_units; u32 offsets[n_units + 1]; u64 nodes[offsets[n_units]]; u64 root_id
 * build: gcc -std=c11 -Wall -Iinclude examples/c_forest.c -Lmapreduce-plonky2_amd -lmp2gpu -o examples/c_forest */
#include "mp2g.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "%s failed: %s\n", #x, mp2g_last_error()); return 1; } } while (0)
#define READ(ptr, count) do { if (fread((ptr), sizeof *(ptr), (count), f) != (size_t)(count)) { fprintf(stderr, "short read\n"); return 1; } } while (0)
#define MAX_STEPS 4
#define MAX_CIRCUITS 8
#define MAX_WORKERS 4
#define NUM_ROUTED 80
#define NUM_WIRES 135

typedef struct {
  uint32_t hdr[12];
  mp2g_gate gates[MP2G_MAX_GATES];
  uint64_t digest[4];
  uint64_t *pre, *tape, *consts;
  uint32_t *input_sids, *probe;
} step_t;
typedef struct { uint32_t n_steps; step_t steps[MAX_STEPS]; mp2g_forest_circuit desc; } circuit_t;

static uint64_t fnv1a(const void* p, size_t len) {
  const uint8_t* b = p;
  uint64_t h = 1469598103934665603ULL;
  for (size_t i = 0; i < len; i++) { h ^= b[i]; h *= 1099511628211ULL; }
  return h;
}

/* the chain of one framework circuit (base + wrap steps) on one worker's context */
static int make_chain(mp2g_ctx* ctx, const circuit_t* c, uint32_t capacity, mp2g_chain** out) {
  mp2g_prover* provers[MAX_STEPS];
  mp2g_witness_program* programs[MAX_STEPS];
  mp2g_fri_params fps[MAX_STEPS];
  const uint64_t* d_digests[MAX_STEPS];
  for (uint32_t s = 0; s < c->n_steps; s++) {
    const step_t* st = &c->steps[s];
    const uint32_t log_n = st->hdr[0], num_constants = st->hdr[1], n_gates = st->hdr[2], num_selectors = st->hdr[3];
    const size_t tape_len = (size_t)st->hdr[10] | ((size_t)st->hdr[11] << 32), pre_words = (size_t)(num_constants + NUM_ROUTED) << log_n;
    mp2g_fri_params* fp = &fps[s];
    memset(fp, 0, sizeof *fp);  /* standard_recursion_config (mp2-common/src/lib.rs:45-47) */
    fp->variant = MP2G_POSEIDON2; fp->log_n = log_n; fp->rate_bits = 3; fp->cap_height = 4; fp->pow_bits = st->hdr[4]; fp->num_queries = st->hdr[5];
    fp->n_layers = mp2g_reduction_arity_bits(log_n, fp->rate_bits, fp->cap_height, 4, 5, fp->arity_bits);
    fp->n_oracles = 4;
    fp->oracle_w[0] = num_constants + NUM_ROUTED; fp->oracle_w[1] = NUM_WIRES; fp->oracle_w[2] = 2 * (NUM_ROUTED / 8); fp->oracle_w[3] = 16;
    fp->zs_oracle = 2; fp->zs_count = 2;
    void *d_pre, *d_cd;
    CHECK(mp2g_dev_alloc(ctx, pre_words * 8, &d_pre));
    CHECK(mp2g_dev_alloc(ctx, 32, &d_cd));
    CHECK(mp2g_h2d(ctx, d_pre, st->pre, pre_words * 8));
    CHECK(mp2g_h2d(ctx, d_cd, st->digest, 32));
    d_digests[s] = d_cd;
    CHECK(mp2g_prover_create(ctx, fp, capacity, &provers[s]));
    CHECK(mp2g_prover_set_preprocessed_dev(provers[s], d_pre));
    CHECK(mp2g_prover_enable_permutation(provers[s], NUM_ROUTED, 8));
    CHECK(mp2g_prover_enable_quotient(provers[s]));
    CHECK(mp2g_prover_set_gates(provers[s], st->gates, n_gates, num_selectors));
    CHECK(mp2g_prover_enable_witness_check(provers[s], 1));
    CHECK(mp2g_witness_program_create(st->tape, tape_len, st->hdr[6], log_n, st->input_sids, st->hdr[7], st->consts, st->hdr[8], &programs[s]));
    CHECK(mp2g_witness_program_set_probe(programs[s], st->probe, st->hdr[9]));
  }
  CHECK(mp2g_chain_create(ctx, c->n_steps, provers, programs, fps, d_digests, capacity, out));
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 2) { fprintf(stderr, "usage: %s forest.bin\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  uint32_t top[5];
  READ(top, 5);
  const uint32_t n_circuits = top[0], n_workers = top[1], capacity = top[2], slot_words = top[3], pool_slots = top[4];
  if (n_circuits < 1 || n_circuits > MAX_CIRCUITS || n_workers < 1 || n_workers > MAX_WORKERS) return 2;
  static circuit_t circuits[MAX_CIRCUITS];
  mp2g_forest_circuit descs[MAX_CIRCUITS];
  for (uint32_t c = 0; c < n_circuits; c++) {
    circuit_t* C = &circuits[c];
    READ(&C->n_steps, 1);
    if (C->n_steps < 1 || C->n_steps > MAX_STEPS) return 2;
    for (uint32_t s = 0; s < C->n_steps; s++) {
      step_t* st = &C->steps[s];
      READ(st->hdr, 12);
      const size_t n = (size_t)1 << st->hdr[0], pre_words = (size_t)(st->hdr[1] + NUM_ROUTED) * n, tape_len = (size_t)st->hdr[10] | ((size_t)st->hdr[11] << 32);
      if (st->hdr[2] > MP2G_MAX_GATES) return 2;
      READ(st->gates, st->hdr[2]);
      READ(st->digest, 4);
      st->pre = malloc(pre_words * 8); st->tape = malloc(tape_len * 8 + 8); st->input_sids = malloc((size_t)st->hdr[7] * 4 + 4);
      st->consts = malloc((size_t)st->hdr[8] * 16 + 8); st->probe = malloc((size_t)st->hdr[9] * 4 + 4);
      READ(st->pre, pre_words); READ(st->tape, tape_len); READ(st->input_sids, st->hdr[7]); READ(st->consts, 2 * (size_t)st->hdr[8]); READ(st->probe, st->hdr[9]);
    }
    uint32_t d[7];
    READ(d, 7);
    C->desc.n_inputs = d[0]; C->desc.n_children = d[1];
    for (int k = 0; k < 4; k++) C->desc.child_offset[k] = d[2 + k];
    C->desc.n_const = d[6];
    descs[c] = C->desc;
  }
  /* one context (= HIP stream) per worker, one chain per circuit on each */
  mp2g_ctx* ctxs[MAX_WORKERS];
  mp2g_chain* chains[MAX_WORKERS * MAX_CIRCUITS];
  for (uint32_t w = 0; w < n_workers; w++) {
    CHECK(mp2g_ctx_create(0, &ctxs[w]));
    for (uint32_t c = 0; c < n_circuits; c++)
      if (make_chain(ctxs[w], &circuits[c], capacity, &chains[w * n_circuits + c])) return 1;
  }
  mp2g_forest* forest;
  CHECK(mp2g_forest_create(n_workers, ctxs, n_circuits, descs, chains, slot_words, pool_slots, &forest));
  for (uint32_t c = 0; c < n_circuits; c++) {
    uint32_t count;
    READ(&count, 1);
    const mp2g_forest_circuit* d = &descs[c];
    uint64_t* ids = malloc((size_t)count * 8 + 8);
    uint64_t* kids = malloc((size_t)count * d->n_children * 8 + 8);
    uint64_t* consts = malloc((size_t)count * d->n_const * 8 + 8);
    READ(ids, count); READ(kids, (size_t)count * d->n_children); READ(consts, (size_t)count * d->n_const);
    CHECK(mp2g_forest_add_nodes(forest, c, count, ids, kids, consts, NULL));
    free(ids); free(kids); free(consts);
  }
  uint32_t n_waves;
  READ(&n_waves, 1);
  for (uint32_t wv = 0; wv < n_waves; wv++) {  /* the waves of an update plan: the units of one wave do not depend on each other */
    uint32_t n_units;
    READ(&n_units, 1);
    uint32_t* offs = malloc((size_t)(n_units + 1) * 4);
    READ(offs, n_units + 1);
    uint64_t* nodes = malloc((size_t)offs[n_units] * 8 + 8);
    READ(nodes, offs[n_units]);
    int rc = mp2g_forest_prove(forest, nodes, offs, n_units);
    if (rc) { printf("forest_prove failed: %s\n", mp2g_last_error()); return 3; }
    free(offs); free(nodes);
  }
  uint64_t root;
  READ(&root, 1);
  fclose(f);
  uint32_t n_words = 0;
  CHECK(mp2g_forest_proof(forest, root, NULL, &n_words));
  uint64_t* words = malloc((size_t)n_words * 8);
  CHECK(mp2g_forest_proof(forest, root, words, &n_words));
  printf("proved=%llu root_words=%u root_fnv1a=%016llx\n", (unsigned long long)mp2g_forest_proved(forest), n_words, (unsigned long long)fnv1a(words, (size_t)n_words * 8));
  mp2g_forest_free(forest);
  for (uint32_t i = 0; i < n_workers * n_circuits; i++) mp2g_chain_free(chains[i]);
  for (uint32_t w = 0; w < n_workers; w++) mp2g_ctx_destroy(ctxs[w]);
  return 0;
}
