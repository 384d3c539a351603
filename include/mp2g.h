/* libmp2gpu -- C ABI of the MI355X (gfx950) back end for the Plonky2 prover path of
 * Lagrange-Labs/mapreduce-plonky2 (mp2-v1 / recursion-framework).
 *
 * The reference has no FFI for this path: every proof is produced by plonky2's
 * `CircuitData::prove(pw)` called at recursion-framework/src/circuit_builder.rs:308,
 * recursion-framework/src/universal_verifier_gadget/wrap_circuit.rs:143 and
 * verifiable-db/src/api.rs:207, with the hasher chosen by `type C = Poseidon2GoldilocksConfig`
 * (mp2-common/src/lib.rs:37-42). A Rust host keeps those signatures and replaces the bodies of
 * plonky2's PolynomialBatch::{from_values,from_coeffs}, MerkleTree::new, fri_proof and the
 * Hasher impl with the calls below (binding sketch: INTEGRATION.md).
 *
 * Conventions follow the only in-tree C ABI, gnark-utils (gnark-utils/src/lib.rs:13-52,
 * lib/lib.go:40-47,142-161,213-216): plain pointers and sizes, an int status (0 = ok), the
 * message of the last failure via mp2g_last_error(), callee-owned handles freed by the matching
 * *_free. Field elements are canonical u64 (< 2^64 - 2^32 + 1), little-endian in memory.
 * `variant` selects the permutation: 0 = Poseidon2 (default C), 1 = Poseidon (WrapC,
 * verifiable-db/src/api.rs:148). A context is bound to one GPU and one HIP stream; it is
 * thread-compatible, not thread-safe. Pointers named d_* are device pointers on the context's
 * GPU, all others are host pointers. No entry point falls back to the CPU.
 */
#ifndef MP2G_H
#define MP2G_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mp2g_ctx mp2g_ctx;
typedef struct mp2g_tree mp2g_tree;   /* plonky2 MerkleTree   (hash/merkle_tree.rs)  */
typedef struct mp2g_batch mp2g_batch; /* plonky2 PolynomialBatch (fri/oracle.rs)     */

#define MP2G_POSEIDON2 0
#define MP2G_POSEIDON 1

/* Diagnostic: Poseidon / Poseidon2 permutations queued by the Merkle leaf sponge (every commitment of every prove() on every
 * context) since the library was loaded -- a host-side count. Divided into the leaf kernel's time in a kernel trace it is the
 * sponge's rate inside a proving step (bench.py `roofline_alu`). Replaces nothing in the reference. */
uint64_t mp2g_stat_leaf_permutations(void);
/* ---- library / context ------------------------------------------------------------------ */
const char* mp2g_last_error(void);
int mp2g_device_count(void);
/* A context = one device, one HIP stream, ONE host thread proving at a time. Every prover created on a context takes its
 * per-batch working buffers (LDE values, Merkle levels, quotient values, FRI layers ...) from the context's shared scratch,
 * which is safe because their launches are ordered on that one stream: a host that wants several proofs in flight
 * concurrently creates one context per worker thread. Environment: MP2G_SHARE_SCRATCH=0 gives every prover buffers of
 * its own (about three times the device memory of a table build; the same proofs). Fails when no HIP device is visible:
 * there is no CPU path. */
int mp2g_ctx_create(int device, mp2g_ctx** out);
void mp2g_ctx_destroy(mp2g_ctx* ctx);
int mp2g_ctx_sync(mp2g_ctx* ctx);
/* HIP's current device is a property of the calling THREAD (0 until set). mp2g_ctx_create sets it for the creating thread; a worker thread
 * that drives a context of another device (one process per GPU with several proving threads, rank > 0 of a node) calls this once before
 * its first call: allocations and launches follow the thread's current device. */
int mp2g_ctx_make_current(mp2g_ctx* ctx);
/* free / total bytes of the context's device (hipMemGetInfo): what a host sizes its provers' capacities against */
int mp2g_ctx_mem_info(mp2g_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
void* mp2g_ctx_stream(mp2g_ctx* ctx);                 /* hipStream_t the context launches on */
int mp2g_ctx_set_stream(mp2g_ctx* ctx, void* stream); /* adopt a caller-owned hipStream_t     */
int mp2g_dev_alloc(mp2g_ctx* ctx, size_t bytes, void** d_ptr);
int mp2g_dev_free(mp2g_ctx* ctx, void* d_ptr);
int mp2g_h2d(mp2g_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int mp2g_d2h(mp2g_ctx* ctx, void* dst, const void* d_src, size_t bytes);
/* stream-ordered strided device-to-device copy: `rows` pieces of width_bytes, the i-th from d_src + i * src_pitch to
 * d_dst + i * dst_pitch (how a batch of proofs becomes the next witness program's input vectors without leaving the device) */
int mp2g_d2d_2d(mp2g_ctx* ctx, void* d_dst, size_t dst_pitch, const void* d_src, size_t src_pitch, size_t width_bytes, size_t rows);
/* pinned host staging memory and stream-ordered uploads (a host that feeds witness matrices from
 * its own memory overlaps the PCIe copy of batch k+1 with the proving of batch k) */
int mp2g_host_alloc(mp2g_ctx* ctx, size_t bytes, void** ptr);
int mp2g_host_free(mp2g_ctx* ctx, void* ptr);
int mp2g_h2d_async(mp2g_ctx* ctx, void* d_dst, const void* src, size_t bytes);
/* HIP-event stopwatch on the context's stream (used by bench.py) */
int mp2g_timer_start(mp2g_ctx* ctx);
int mp2g_timer_stop(mp2g_ctx* ctx, float* ms);

/* ---- NTT / LDE: replaces plonky2_field fft.rs fft/ifft/coset_fft and the LDE loop of
 *      PolynomialBatch::from_coeffs ------------------------------------------------------ */
/* `batch` transforms of 2^log_n points, in place. inverse=0: coefficients -> values
 * v[i] = P(shift * w^i) (coset_shift 0 = no shift); inverse=1: values -> coefficients.
 * bitrev_out!=0 leaves the output in bit-reversed index order. */
int mp2g_ntt(mp2g_ctx* ctx, uint64_t* data, uint32_t log_n, uint32_t batch, int inverse,
             uint64_t coset_shift, int bitrev_out);
int mp2g_ntt_dev(mp2g_ctx* ctx, const uint64_t* d_in, uint64_t* d_out, uint32_t log_n, uint32_t batch,
                 int inverse, uint64_t coset_shift, int bitrev_out);
/* coeffs [w][n] -> leaves [n << rate_bits][w]: row i = evaluations at g * w_N^bitrev(i)
 * (transpose + reverse_index_bits of from_coeffs). Host layout of plonky2. */
int mp2g_lde_leaves(mp2g_ctx* ctx, const uint64_t* coeffs, uint32_t log_n, uint32_t w, uint32_t rate_bits,
                    uint64_t* leaves);
/* same values, kept polynomial-major on the device: d_values [w][n << rate_bits], index
 * bit-reversed (column i of that matrix is leaf i). */
int mp2g_lde_dev(mp2g_ctx* ctx, const uint64_t* d_coeffs, uint32_t log_n, uint32_t w, uint32_t rate_bits,
                 uint64_t* d_values);

/* ---- hashing / Merkle: replaces Hasher::{hash_no_pad,hash_or_noop,two_to_one} and
 *      MerkleTree::{new,prove} ------------------------------------------------------------- */
/* out[i][0..out_len) = hash_n_to_m_no_pad(in[i][0..in_len)), out_len = 4 (HashOut) or 5
 * (map-to-curve, mp2-common/src/group_hashing/field_to_curve.rs:41-47) */
int mp2g_hash_no_pad_batch(mp2g_ctx* ctx, int variant, const uint64_t* in, uint32_t in_len, uint32_t count,
                           uint32_t out_len, uint64_t* out);
int mp2g_hash_no_pad_batch_dev(mp2g_ctx* ctx, int variant, const uint64_t* d_in, uint32_t in_len,
                               uint32_t count, uint32_t out_len, uint64_t* d_out);
/* MerkleTree::new(leaves, cap_height); leaves [2^log_leaves][leaf_len] */
int mp2g_merkle_build(mp2g_ctx* ctx, int variant, const uint64_t* leaves, uint32_t leaf_len,
                      uint32_t log_leaves, uint32_t cap_height, mp2g_tree** out);
int mp2g_merkle_cap(const mp2g_tree* tree, uint64_t* cap /* [1<<cap_height][4] */);
/* leaves_out [n_idx][leaf_len] (may be NULL), siblings_out [n_idx][log_leaves-cap_height][4] bottom-up */
int mp2g_merkle_open(const mp2g_tree* tree, const uint32_t* idx, uint32_t n_idx, uint64_t* leaves_out,
                     uint64_t* siblings_out);
void mp2g_merkle_free(mp2g_tree* tree);

/* ---- polynomial commitment: replaces PolynomialBatch::{from_values,from_coeffs} ----------- */
/* values [w][n] over the subgroup (natural order): per-poly iFFT, LDE x 2^rate_bits on the
 * coset g<w_N>, Merkle tree with cap. */
int mp2g_commit_from_values(mp2g_ctx* ctx, int variant, const uint64_t* values, uint32_t log_n, uint32_t w,
                            uint32_t rate_bits, uint32_t cap_height, mp2g_batch** out);
int mp2g_commit_from_values_dev(mp2g_ctx* ctx, int variant, const uint64_t* d_values, uint32_t log_n,
                                uint32_t w, uint32_t rate_bits, uint32_t cap_height, mp2g_batch** out);
int mp2g_commit_from_coeffs_dev(mp2g_ctx* ctx, int variant, const uint64_t* d_coeffs, uint32_t log_n,
                                uint32_t w, uint32_t rate_bits, uint32_t cap_height, mp2g_batch** out);
/* re-run the commitment into an existing batch of the same shape (no allocation: the form the
 * batched prover and bench.py use) */
int mp2g_recommit_from_values_dev(mp2g_ctx* ctx, mp2g_batch* batch, const uint64_t* d_values);
/* MerkleTree::new(leaves, cap_height) again over the LDE values the batch already holds ([dep] plonky2 hash/merkle_tree.rs;
 * called by PolynomialBatch::from_coeffs): parts = 1 the leaf sponges only (digests of the 2^(log_n + rate_bits) leaves),
 * 2 the tree levels above them only, 3 both. Same cap as the commitment made; what bench.py times the leaf kernel with. */
int mp2g_batch_rehash_dev(mp2g_ctx* ctx, mp2g_batch* batch, int parts);
int mp2g_batch_cap(const mp2g_batch* batch, uint64_t* cap);
int mp2g_batch_coeffs(const mp2g_batch* batch, uint64_t* coeffs /* [w][n] */);
int mp2g_batch_open(const mp2g_batch* batch, const uint32_t* idx, uint32_t n_idx,
                    uint64_t* leaves_out /* [n_idx][w] */, uint64_t* siblings_out);
void mp2g_batch_free(mp2g_batch* batch);

/* ---- Fiat-Shamir + FRI: replaces iop/challenger.rs, fri/prover.rs fri_proof and
 *      PolynomialBatch::prove_openings ---------------------------------------------------- */
/* FRI / PCS parameters of one circuit shape. standard_recursion_config
 * (mp2-common/src/lib.rs:45-47): rate_bits 3, cap_height 4, pow_bits 16, num_queries 28,
 * reduction ConstantArityBits(4,5); oracles = constants_sigmas, wires, zs_partial_products,
 * quotient; the first zs_count polynomials of oracle zs_oracle are also opened at g*zeta. */
typedef struct mp2g_fri_params {
  uint32_t variant;
  uint32_t log_n;       /* degree_bits */
  uint32_t rate_bits;
  uint32_t cap_height;
  uint32_t pow_bits;
  uint32_t num_queries;
  uint32_t n_layers;    /* len(reduction_arity_bits) */
  uint32_t arity_bits[8];
  uint32_t n_oracles;
  uint32_t oracle_w[8];
  uint32_t zs_oracle;
  uint32_t zs_count;
  /* lookup argument: polynomials per challenge round (0 = none; ceil((num_routed/2) / 7) + 1 = 7 under
   * standard_recursion_config). The last zs_count * num_lookup_polys polynomials of oracle zs_oracle are the
   * lookup polynomials (RE, then the partial Sum/LDC polynomials, per round); they are opened at zeta and at
   * g*zeta and close both FRI batches (plonk/circuit_data.rs fri_all_polys / fri_next_batch_polys). */
  uint32_t num_lookup_polys;
} mp2g_fri_params;
/* fri/reduction_strategies.rs ConstantArityBits(arity_bits, final_poly_bits); returns the count */
uint32_t mp2g_reduction_arity_bits(uint32_t degree_bits, uint32_t rate_bits, uint32_t cap_height,
                                   uint32_t arity_bits, uint32_t final_poly_bits, uint32_t* out);
/* Flat FriProof layout in u64 words (the fields of plonky2's FriProof in serde order, without
 * bincode length prefixes):
 *   commit_phase_merkle_caps [n_layers][1<<cap][4]
 *   query_round_proofs [num_queries] { initial_trees_proof: per oracle { leaf[w], siblings[lg-cap][4] };
 *                                      steps: per layer { evals[1<<arity][2], siblings[..][4] } }
 *   final_poly [final_len][2]
 *   pow_witness */
size_t mp2g_fri_proof_words(const mp2g_fri_params* p);
/* sum(oracle_w) + zs_count + zs_count * num_lookup_polys extension values, in FRI batch order
 * (OpeningSet::to_fri_openings): at zeta every polynomial in oracle order except the lookup polynomials, which
 * come last (after the quotient chunks); then at g*zeta the Z polynomials and the lookup polynomials. */
size_t mp2g_fri_n_openings(const mp2g_fri_params* p);

/* Device-resident challengers: `count` independent transcripts stepped in lockstep. */
typedef struct mp2g_challenger mp2g_challenger;
int mp2g_challenger_create(mp2g_ctx* ctx, int variant, uint32_t count, mp2g_challenger** out);
/* transcript t observes elems[t*n .. t*n+n) */
int mp2g_challenger_observe(mp2g_challenger* ch, const uint64_t* elems, uint32_t n);
int mp2g_challenger_get(mp2g_challenger* ch, uint32_t n, uint64_t* out /* [count][n] */);
void mp2g_challenger_free(mp2g_challenger* ch);

/* One FRI layer fold in the value domain. evals [1<<log_m][2]: bit-reversed evaluations on
 * shift*<w_m>; out [(1<<log_m)>>arity_bits][2]: bit-reversed evaluations of the folded polynomial
 * on shift^(2^arity_bits). Equals coeffs.chunks(arity).map(reduce_with_powers(beta)) + coset_fft. */
int mp2g_fri_fold(mp2g_ctx* ctx, const uint64_t* evals, uint32_t log_m, uint32_t arity_bits,
                  const uint64_t beta[2], uint64_t shift, uint64_t* out);
/* fri_proof_of_work: smallest witness w such that permute(state with state[pos] = w)[7] has
 * >= bits leading zeros; `state` = sponge state already overwritten with the pending inputs. */
int mp2g_fri_pow(mp2g_ctx* ctx, int variant, const uint64_t state[12], uint32_t pos, uint32_t bits,
                 uint64_t* witness);

/* Openings of one committed batch at an extension point (plonk/prover.rs eval_commitment):
 * out[p] = [c0, c1] of polynomial p evaluated at point. */
int mp2g_batch_eval_ext(const mp2g_batch* batch, const uint64_t point[2], uint64_t* out /* [w][2] */);
/* PolynomialBatch::prove_openings / fri_proof over committed batches: `oracles` in FRI order
 * (constants_sigmas, wires, zs_partial_products, quotient), zeta the opening point, `ch` a count-1
 * challenger that has already observed the openings; on return it has absorbed the FRI transcript.
 * proof: mp2g_fri_proof_words(params) words, host. This is the call a host makes after computing the
 * quotient polynomials itself. */
int mp2g_fri_prove(mp2g_ctx* ctx, const mp2g_fri_params* params, mp2g_batch* const* oracles, const uint64_t zeta[2],
                   mp2g_challenger* ch, uint64_t* proof);

/* Batched PCS prover: the commitment / Fiat-Shamir / opening / FRI skeleton of plonky2's prove()
 * for `batch` same-shape proofs at once, everything resident on the device. */
typedef struct mp2g_prover mp2g_prover;
int mp2g_prover_create(mp2g_ctx* ctx, const mp2g_fri_params* params, uint32_t batch, mp2g_prover** out);
/* commit oracle 0 (constants_sigmas, [w0][n] subgroup values), shared by every proof */
int mp2g_prover_set_preprocessed_dev(mp2g_prover* pr, const uint64_t* d_values);
/* d_values[o-1] = [batch][w_o][n] subgroup values of oracle o = 1..n_oracles-1;
 * d_circuit_digest [4]; d_pi_hash [batch][4]. Outputs (device): d_caps [batch][n_oracles][1<<cap][4],
 * d_openings [batch][n_openings][2], d_proof [batch][proof_words]. Asynchronous on the stream. */
int mp2g_prover_prove_dev(mp2g_prover* pr, const uint64_t* const* d_values, const uint64_t* d_circuit_digest,
                          const uint64_t* d_pi_hash, uint64_t* d_caps, uint64_t* d_openings, uint64_t* d_proof);
void mp2g_prover_free(mp2g_prover* pr);
/* single proof, host pointers: values[o] = [w_o][n] for o = 0..n_oracles-1 */
int mp2g_pcs_prove(mp2g_ctx* ctx, const mp2g_fri_params* params, const uint64_t* const* values,
                   const uint64_t circuit_digest[4], const uint64_t pi_hash[4], uint64_t* caps,
                   uint64_t* openings, uint64_t* proof);

/* ---- Ecgfp5 multiset digest (off-circuit value side) -------------------------------------- */
/* A point is returned as its canonical 5-limb encoding w = y/x (plonky2_ecgfp5 Point::encode;
 * the form the reference's known-answer test holds, sswu_value.rs:88-118) and/or as the 11-limb
 * short-Weierstrass form [x0..x4, y0..y4, is_inf] of mp2-common/src/group_hashing/mod.rs:163-174.
 * Either output pointer may be NULL. */
/* map_to_curve_point (field_to_curve.rs:36-48) of `count` inputs of in_len limbs each */
int mp2g_map_to_curve_batch(mp2g_ctx* ctx, int variant, const uint64_t* in, uint32_t in_len, uint32_t count,
                            uint64_t* out_w /* [count][5] */, uint64_t* out_weierstrass /* [count][11] */);
/* add_curve_point (curve_add.rs:17-22) over `count` encoded points; fails on an invalid encoding */
int mp2g_curve_sum(mp2g_ctx* ctx, const uint64_t* pts_w /* [count][5] */, uint32_t count, uint64_t out_w[5],
                   uint64_t out_weierstrass[11]);
/* the same sum over n_ranges index ranges [start, end) of one point array at once: the accumulated digest of every node of a
 * tree laid out in order (a subtree = one contiguous range) -- what SplitDigestPoint::accumulate (mp2-common/src/digest.rs:38-47)
 * builds node by node up the cells tree / row tree (verifiable-db/src/cells_tree/full_node.rs:26-28,
 * row_tree/full_node.rs:78-82). An empty range gives the neutral point. */
int mp2g_curve_sum_ranges(mp2g_ctx* ctx, const uint64_t* pts_w /* [count][5] */, uint32_t count, const uint32_t* ranges /* [n_ranges][2] */,
                          uint32_t n_ranges, uint64_t* out_w /* [n_ranges][5] */, uint64_t* out_weierstrass /* [n_ranges][11] */);
/* scalar * point for 128-bit scalars given as 4 little-endian u32 limbs (hash_to_int_value,
 * mp2-common/src/poseidon.rs:120-133) */
int mp2g_scalar_mul_batch(mp2g_ctx* ctx, const uint64_t* pts_w /* [count][5] */, const uint32_t* scalars /* [count][4] */,
                          uint32_t count, uint64_t* out_w /* [count][5] */, uint64_t* out_weierstrass);
/* field_hashed_scalar_mul (group_hashing/mod.rs:220-225): HashToInt(H(inputs)) * base */
int mp2g_field_hashed_scalar_mul(mp2g_ctx* ctx, int variant, const uint64_t* inputs, uint32_t n_inputs,
                                 const uint64_t base_w[5], uint64_t out_w[5], uint64_t out_weierstrass[11]);
/* compute_table_row_digest (mp2-v1/src/values_extraction/mod.rs:527-571):
 *   sum over rows of row_id * sum over columns of D(id_c || value_c), row_id =
 *   HashToInt(H(H(unique column values) || n_cols)).
 * col_ids [n_cols]; values [rows][n_cols][8] and unique [rows][n_unique][8]: each U256 as 8
 * big-endian u32 words, most significant first (mp2-common/src/u256.rs:870-877). */
int mp2g_row_digest_batch(mp2g_ctx* ctx, int variant, const uint64_t* col_ids, uint32_t n_cols,
                          const uint32_t* values, const uint32_t* unique, uint32_t n_unique, uint32_t rows,
                          uint64_t out_w[5], uint64_t out_weierstrass[11]);
/* the per-row terms of that sum, row_id * sum over columns of D(id_c || value_c): the individual digest a row-tree node
 * contributes (verifiable-db/src/row_tree/secondary_index_cell.rs:99-139 with every cell individual) */
int mp2g_row_digests(mp2g_ctx* ctx, int variant, const uint64_t* col_ids, uint32_t n_cols, const uint32_t* values,
                     const uint32_t* unique, uint32_t n_unique, uint32_t rows, uint64_t* out_w /* [rows][5] */,
                     uint64_t* out_weierstrass /* [rows][11] */);
/* same with device-resident inputs; d_frac_out [20] receives the sum in fractional coordinates
 * (X:Z:U:T) for further accumulation -- a representative of the point, not a canonical form: compare points through out_w
 * (Point::encode) or out_weierstrass; out_w / out_weierstrass are host pointers (may be NULL) */
int mp2g_row_digest_batch_dev(mp2g_ctx* ctx, int variant, const uint64_t* d_col_ids, uint32_t n_cols,
                              const uint32_t* d_values, const uint32_t* d_unique, uint32_t n_unique, uint32_t rows,
                              uint64_t* d_frac_out, uint64_t out_w[5], uint64_t out_weierstrass[11]);

/* ---- proof wire format between tree levels / GPUs ------------------------------------------ */
/* bincode 1.3 (little-endian fixed-width ints, u64 length prefixes) serialization of plonky2's
 * ProofWithPublicInputs<F, C, 2> exactly as mp2-common/src/proof.rs:84-98 `serialize_proof`
 * produces it, from the prover's flat outputs. `caps` = [n_oracles][1<<cap][4] (oracle 0, the
 * preprocessed one, is not part of a proof and is skipped); `openings` as produced by the prover
 * ([sum w][2] at zeta in oracle order, then [zs_count][2] at g*zeta); the first num_constants
 * polynomials of oracle 0 are `constants`, the rest `plonk_sigmas`; the first zs_count of oracle
 * zs_oracle are `plonk_zs`, then `partial_products`, then (num_lookup_polys > 0) `lookup_zs`, whose values at
 * g*zeta are `lookup_zs_next`.
 * Call with out = NULL to get the size in *out_len. */
int mp2g_proof_serialize(const mp2g_fri_params* params, uint32_t num_constants, const uint64_t* caps,
                         const uint64_t* openings, const uint64_t* fri_proof, const uint64_t* public_inputs,
                         uint32_t n_public_inputs, uint8_t* out, size_t* out_len);
/* inverse of the above; all output arrays sized as the prover's; fails on malformed input */
int mp2g_proof_deserialize(const mp2g_fri_params* params, uint32_t num_constants, const uint8_t* bytes, size_t len,
                           uint64_t* caps, uint64_t* openings, uint64_t* fri_proof, uint64_t* public_inputs,
                           uint32_t n_public_inputs);
/* ProofWithVK::serialize (mp2-common/src/proof.rs:42-52): bincode(proof) followed by the verifier
 * key as a length-prefixed byte blob (plonky2 VerifierOnlyCircuitData::to_bytes: cap HEIGHT as
 * u64, the cap hashes, the circuit digest). vk_cap_len must be a power of two. */
int mp2g_proof_with_vk_serialize(const uint8_t* proof_bytes, size_t proof_len, const uint64_t* vk_cap,
                                 uint32_t vk_cap_len, const uint64_t vk_circuit_digest[4], uint8_t* out, size_t* out_len);
/* ProofWithVK::deserialize (mp2-common/src/proof.rs:54-57), the inverse: the proof's parts as mp2g_proof_deserialize hands
 * them out, then the verifier key -- vk_cap [vk_cap_len][4] (the caller states the cap length it expects: a blob with
 * another cap height is refused) and the circuit digest. What a proof store (mp2-v1/tests/common/proof_storage.rs:139-140
 * `get_proof_exact`) returns is fed to this before the proof enters its parent's witness. */
int mp2g_proof_with_vk_deserialize(const mp2g_fri_params* params, uint32_t num_constants, const uint8_t* bytes, size_t len,
                                   uint64_t* caps, uint64_t* openings, uint64_t* fri_proof, uint64_t* public_inputs,
                                   uint32_t n_public_inputs, uint64_t* vk_cap, uint32_t vk_cap_len,
                                   uint64_t vk_circuit_digest[4]);

/* ---- permutation argument: replaces plonk/prover.rs all_wires_permutation_partial_products ---- */
/* wires [wires_w][n] and sigmas [num_routed][n]: subgroup values (natural order); only the first
 * num_routed wire columns are read. degree = quotient_degree_factor (8 in standard_recursion_config),
 * num_routed % degree == 0. out [nc * num_routed/degree][n] in the order prove() commits:
 * Z of every challenge, then each challenge's num_routed/degree - 1 partial products. */
int mp2g_partial_products_and_zs(mp2g_ctx* ctx, const uint64_t* wires, uint32_t wires_w, const uint64_t* sigmas,
                                 uint32_t log_n, uint32_t num_routed, uint32_t degree, const uint64_t* betas,
                                 const uint64_t* gammas, uint32_t nc, uint64_t* out);
/* Let the batched prover compute oracle 2 itself (d_values[1] of mp2g_prover_prove_dev may then be
 * NULL): the sigmas are the last num_routed polynomials of the preprocessed oracle, betas/gammas
 * the challenges drawn after the wires cap. Call after mp2g_prover_set_preprocessed_dev. */
int mp2g_prover_enable_permutation(mp2g_prover* pr, uint32_t num_routed, uint32_t degree);
/* Also compute oracle 3 (the quotient chunks) on the device, as plonk/prover.rs compute_quotient_polys
 * does: the vanishing terms Z(1) = 1 and the partial-product checks, plus the gate constraints once
 * mp2g_prover_set_gates has given the gate table (without a table: the complete prove() of a circuit whose
 * only constraints are copy constraints). d_values[2] may then be NULL. Needs
 * mp2g_prover_enable_permutation, rate_bits 3 and oracle_w[3] = zs_count * 8. */
int mp2g_prover_enable_quotient(mp2g_prover* pr);

/* PublicInputGate's witness generator on the device: before the wires are committed, every proof's
 * d_pi_hash[b][0..4) is written to wires 0..3 of row `row` (the circuit's PublicInputGate row) of its wire
 * matrix, IN PLACE in d_values[0] of mp2g_prover_prove_dev. A batch of proofs of one circuit that differ only
 * in their public inputs can then share one witness template (the aggregation levels of
 * recursion-framework/tests/integration.rs:138-261). row < 0 turns it off. */
int mp2g_prover_bind_public_inputs(mp2g_prover* pr, int64_t row);

/* plonky2's prove() panics on a witness that violates a constraint (the reference's tests depend on it:
 * recursion-framework/src/framework.rs:694-700). With the check on, every mp2g_prover_prove_dev also
 * evaluates, per proof, the gate constraints on the subgroup (needs mp2g_prover_set_gates) and the
 * wrap-around of the permutation product; mp2g_prover_witness_status synchronises and returns non-zero with
 * a message naming the first offending proof. flags (may be NULL) receives one word per proof: bit 0 a
 * copy constraint, bit 1 a gate constraint, bit 2 the lookup argument (a looked-up pair that is not in its table). The proof itself is still produced (it does not verify). */
int mp2g_prover_enable_witness_check(mp2g_prover* pr, int on);
/* prove only the first n <= batch witnesses from the next call on (every buffer is proof-major, so a prover created for `batch`
 * proofs serves any smaller batch without new allocations: the narrow levels of a tree reuse the wide levels' prover) */
int mp2g_prover_set_active(mp2g_prover* pr, uint32_t n);
int mp2g_prover_witness_status(mp2g_prover* pr, uint32_t* flags);
int mp2g_prover_witness_check_enabled(const mp2g_prover* pr);
/* Replay the prover's launch sequence (several hundred small kernels per call) as a hipGraph: the
 * first call after enabling runs normally (it creates the cached twiddle tables), the second is captured,
 * later calls with the same buffer addresses launch the instantiated graph. A call with different
 * addresses re-captures. Cuts the launch-bound latency of small batches; results are identical. */
int mp2g_prover_enable_graph(mp2g_prover* pr, int on);
/* Stage timing of the batched prover (measurement aid; plonky2 prints the same split through its
 * `timed!` macro inside prove()). With timing on, every mp2g_prover_prove_dev records HIP events on
 * the prover's stream at the phase boundaries; mp2g_prover_stage_ms synchronises and returns the
 * milliseconds of the last call: 0 wires commitment, 1 Z / partial products + commitment, 2 quotient
 * polynomials + commitment, 3 openings, 4 FRI batch composition + commit phase, 5 proof of work,
 * 6 query rounds. */
#define MP2G_N_STAGES 7
int mp2g_prover_enable_timing(mp2g_prover* pr, int on);
int mp2g_prover_stage_ms(mp2g_prover* pr, float out[MP2G_N_STAGES]);

/* ---- gate constraints: the third part of compute_quotient_polys --------------------------------
 * Replaces [dep] plonky2 plonk/vanishing_poly.rs evaluate_gate_constraints_base_batch and the
 * eval_unfiltered_base of the gates below: all 26 entries the reference registers
 * (mp2-common/src/serialization/circuit_data_serialization.rs:236-267). */
enum {
  MP2G_GATE_NOOP = 0,
  MP2G_GATE_CONSTANT = 1,       /* p0 = num_consts */
  MP2G_GATE_PUBLIC_INPUT = 2,
  MP2G_GATE_ARITHMETIC = 3,     /* p0 = num_ops */
  MP2G_GATE_BASE_SUM = 4,       /* p0 = num_limbs, p1 = base (BaseSumGate<B>) */
  MP2G_GATE_ARITHMETIC_EXT = 5, /* p0 = num_ops (D = 2) */
  MP2G_GATE_MUL_EXT = 6,        /* p0 = num_ops */
  MP2G_GATE_POSEIDON2 = 7,
  MP2G_GATE_EXPONENTIATION = 8, /* p0 = num_power_bits */
  MP2G_GATE_REDUCING = 9,       /* p0 = num_coeffs */
  MP2G_GATE_REDUCING_EXT = 10,  /* p0 = num_coeffs */
  MP2G_GATE_RANDOM_ACCESS = 11, /* p0 = bits (<= 6), p1 = num_copies, p2 = num_extra_constants */
  MP2G_GATE_POSEIDON = 12,      /* the original Poseidon permutation gate (wrap circuits) */
  MP2G_GATE_POSEIDON_MDS = 13,
  MP2G_GATE_COSET_INTERPOLATION = 14, /* p0 = subgroup_bits (2..5), p1 = degree (CosetInterpolationGate::degree) */
  /* plonky2-u32 (restated from memory of the published crate, like everything here that is not in the tree) */
  MP2G_GATE_U32_ARITHMETIC = 15,  /* p0 = num_ops */
  MP2G_GATE_U32_RANGE_CHECK = 16, /* p0 = num_input_limbs */
  MP2G_GATE_U32_SUBTRACTION = 17, /* p0 = num_ops */
  MP2G_GATE_U32_ADD_MANY = 18,    /* p0 = num_addends (<= 16), p1 = num_ops */
  MP2G_GATE_COMPARISON = 19,      /* p0 = num_bits, p1 = num_chunks (chunks of at most 4 bits) */
  /* plonky2 gates/lookup.rs, gates/lookup_table.rs: no constraints of their own, the lookup argument of prove()
   * (mp2g_prover_set_lookups) carries them. p0 = num_slots (num_routed / 2, num_routed / 3) */
  MP2G_GATE_LOOKUP = 20,
  MP2G_GATE_LOOKUP_TABLE = 21,
  /* plonky2_crypto u32/gates/{interleave_u32, uninterleave_to_b32, uninterleave_to_u32}.rs (Keccak / SHA in the MPT
   * circuits), restated from memory of the published crate: p0 = num_ops */
  MP2G_GATE_U32_INTERLEAVE = 22,
  MP2G_GATE_UNINTERLEAVE_TO_B32 = 23,
  MP2G_GATE_UNINTERLEAVE_TO_U32 = 24
};
#define MP2G_MAX_GATES 32
#define MP2G_MAX_GATE_CONSTRAINTS 160
/* One entry per gate of CommonCircuitData::gates, in that order (sorted by degree). The selector
 * fields restate SelectorsInfo (gates/selectors.rs): the gate's filter is
 * prod_{r in [group_start, group_end), r != index} (r - s) * (num_selectors > 1 ? (u32::MAX - s) : 1)
 * with s = local_constants[selector_index]. */
typedef struct {
  uint32_t kind, p0, p1, p2;
  uint32_t selector_index, group_start, group_end;
} mp2g_gate;
/* Gate::num_constraints / Gate::degree of a descriptor; 0 for an unknown kind */
uint32_t mp2g_gate_num_constraints(const mp2g_gate* g);
uint32_t mp2g_gate_degree(const mp2g_gate* g);
/* Give the batched prover the gate table: the quotient then carries the gate constraint terms after
 * the permutation terms (eval_vanishing_poly_base_batch order), i.e. the complete prove() of a circuit
 * built from the gates above. Constants are the first oracle_w[0] - num_routed polynomials of the
 * preprocessed oracle (selectors first, then the gate constants); the public-inputs hash is the
 * d_pi_hash of mp2g_prover_prove_dev. Needs mp2g_prover_enable_quotient. n_gates = 0 removes it. */
int mp2g_prover_set_gates(mp2g_prover* pr, const mp2g_gate* gates, uint32_t n_gates, uint32_t num_selectors);
/* One lookup table of the circuit: plonky2's LookupWire (the rows CircuitBuilder::add_all_lookups appended for it:
 * LookupGate rows [last_lu_row, last_lut_row), LookupTableGate rows [last_lut_row, first_lut_row] with the table
 * running DOWN from first_lut_row, then a Noop row) and the table itself (CommonCircuitData::luts). */
typedef struct {
  uint32_t last_lu_row, last_lut_row, first_lut_row;
  uint32_t table_len;
  const uint16_t* table; /* [table_len][2] = (input, output), host pointer */
} mp2g_lookup;
#define MP2G_MAX_LUTS 16
/* The lookup argument of prove() ([dep] plonk/prover.rs compute_lookup_polys, vanishing_poly.rs
 * check_lookup_constraints): after the wires cap the transcript also yields the lookup challenges (deltas = betas,
 * gammas and 2 * num_challenges more), the RE / Sum / LDC polynomials are computed on the device into the tail of
 * oracle 2, and their constraints enter the quotient between the partial-product and the gate terms. The constants
 * of the preprocessed oracle are then: selectors, the 4 + n_luts lookup selectors (gates/selectors.rs
 * selectors_lookup, selector_ends_lookups), gate constants. Needs mp2g_prover_set_gates (with the LookupGate /
 * LookupTableGate entries), params.num_lookup_polys = ceil((num_routed/2) / (degree-1)) + 1 and
 * oracle_w[2] = zs_count * (num_routed/degree + num_lookup_polys). The multiplicity wires are part of the witness
 * (prove()'s set_lookup_wires fills them on the host). n_luts = 0 removes the argument. */
int mp2g_prover_set_lookups(mp2g_prover* pr, const mp2g_lookup* luts, uint32_t n_luts);

/* The filtered constraints C_j = sum_gates filter_g c_{g,j} at npts arbitrary points (host pointers):
 * consts [num_constants][npts], wires [wires_w][npts], out [max_j][npts] with max_j the largest
 * num_constraints of the table. On the subgroup H a satisfied witness gives all zeros -- the check
 * behind plonky2's "invalid witness" panic in prove(). */
int mp2g_eval_gate_constraints(mp2g_ctx* ctx, const mp2g_gate* gates, uint32_t n_gates, uint32_t num_selectors,
                               const uint64_t* consts, uint32_t num_constants, const uint64_t* wires, uint32_t wires_w,
                               uint64_t npts, const uint64_t pi_hash[4], uint64_t* out);

/* ---- witness generation: the witness tape ------------------------------------------------------------------
 * Replaces [dep] plonky2 iop/generator.rs generate_partial_witness -- the first line of prove() at
 * recursion-framework/src/circuit_builder.rs:308 and universal_verifier_gadget/wrap_circuit.rs:143 -- for circuits whose
 * generator graph does not depend on the witness (every circuit of the recursion framework and of the table build: the
 * set of generators and the wires they read and write are fixed when the circuit is built). plonky2 walks a dependency
 * graph of the generators of mp2-common/src/serialization/circuit_data_serialization.rs:186-231 per proof; here the host
 * records them ONCE, in an order in which every value is produced before it is used, as a flat array of u64 words -- the
 * TAPE -- and the library replays it per proof, on host threads (mp2g_witness_program_run) or on the device for a whole batch
 * (mp2g_witness_program_run_dev: one block per proof, the instructions grouped by dependency level). INTEGRATION.md
 * section 6 maps every generator of the reference's registry to its opcode.
 *
 * VALUES live in SLOTS: n_slots u64 cells per proof, indexed by the tape. A slot plays the part of a plonky2 Target's
 * value; copy constraints need no instruction (two wires that are copies of each other are written from the same slot).
 * Before the tape runs, slot const_slots[2 i] holds the canonical field element const_slots[2 i + 1] and slot input_sids[j]
 * holds the proof's j-th input word; every other slot holds 0 until an instruction writes it.
 * WIRES: wire (col, row) of the 135 x 2^log_n matrix prove() commits to (standard_recursion_config: 135 wires, the first
 * 80 routed). The matrix is zero-filled per proof; an instruction writes the wires of ITS generator's gate (listed below),
 * MP2G_OP_WIRE writes a single one.
 *
 * INSTRUCTION = one opcode word followed by its operands. Operand kinds below: `row` a gate row (< 2^log_n); `i`, `col`,
 * counts: small integers as stated; `k...` a canonical field element (< p); `s...` a slot that is READ; `d...` a slot
 * that is WRITTEN; x[n] = n consecutive operand words; an extension element takes two slots (c0, c1).
 *
 * RULES. (1) Host replay executes the tape in order. (2) The device replay re-orders: an instruction's level is one more
 * than the highest level of the slots it reads (inputs and constants are level 0), and a level's instructions run
 * concurrently; it therefore needs single assignment -- no slot written twice, no slot written after an earlier
 * instruction read it. mp2g_witness_program_run_dev refuses a tape that breaks this (the host replay still accepts it).
 * (3) Two instructions must not write the same wire with different values (they may run in either order on the device).
 * (4) MP2G_OP_PAR brackets sections that neither read each other's written slots nor write the same slots or wires.
 *
 * What mp2g_witness_program_create VALIDATES (a tape that fails is refused with a message, nothing is run): every opcode
 * is known, no instruction is truncated, every row < 2^log_n, every column < 135, gate-operation indices and counts are in
 * the ranges given below, constants are canonical, every slot operand, input slot and constant slot is < n_slots,
 * parallel regions do not nest and their section lengths end on instruction boundaries. It does NOT check that the
 * values satisfy the circuit: that is prove()'s witness check (mp2g_prover_enable_witness_check), which fails the proof
 * the way plonky2's prove() panics on an unsatisfied witness. */
enum mp2g_witness_op {
  /* ArithmeticGate (20 operations a row) / ArithmeticBaseGenerator: out = k_c0 m0 m1 + k_c1 addend.
   * operands: row, i (< 20), k_c0, k_c1, s_m0, s_m1, s_addend, d_out.   wires 4i .. 4i+3 = m0, m1, addend, out.
   * k_c0, k_c1 must be the row's two gate constants (the caller's preprocessed constants hold them). */
  MP2G_OP_ARITH = 1,
  /* ArithmeticExtensionGate (10 operations a row) / ArithmeticExtensionGenerator, over the quadratic extension.
   * operands: row, i (< 10), k_c0, k_c1, s_m0[2], s_m1[2], s_addend[2], d_out[2].   wires 8i .. 8i+7 in that order. */
  MP2G_OP_ARITH_EXT = 2,
  /* Poseidon2Gate / Poseidon2Generator: one permutation, with the swap of the first two 4-limb chunks.
   * operands: row, s_in[12], s_swap (0 or 1), d_out[12].   wires 0..11 inputs, 12..23 outputs, 24 swap, 25..28 the swap
   * deltas, 29..64 / 65..86 / 87..134 the S-box inputs of the first full rounds 1..3, the 22 partial rounds, the last 4 full rounds. */
  MP2G_OP_P2 = 3,
  /* BaseSumGate<2> with 63 limbs / BaseSplitGenerator<2>: the bits of a value below 2^63, little endian.
   * operands: row, s_x, d_bit[63].   wire 0 = x, wires 1..63 = bits.  (other bases / limb counts: MP2G_OP_BASE_SPLIT) */
  MP2G_OP_BASE_SUM = 4,
  /* RandomAccessGate (bits = 4, 4 copies a row, 2 extra constants) / RandomAccessGenerator: out = value[index].
   * operands: row, copy (< 4), s_index (< 16), s_value[16], d_out.   wires 18 copy + 0 = index, + 1 = out, + 2..17 = values;
   * the 4 bits of the index at wires 74 + 4 copy .. (not routed).  (wires 72, 73: the row's extra constants, MP2G_OP_WIRE) */
  MP2G_OP_RA = 5,
  /* ReducingGate with 43 coefficients / ReducingGenerator: acc <- acc alpha + coeff_j for j = 0..42, base-field coefficients.
   * operands: row, s_alpha[2], s_old_acc[2], s_coeff[43], d_out[2].   wires 0,1 out; 2,3 alpha; 4,5 old acc; 6..48 coefficients;
   * 49.. the 42 intermediate accumulators. */
  MP2G_OP_REDUCING = 6,
  /* ReducingExtensionGate with 32 coefficients / ReducingExtensionGenerator: the same with extension coefficients.
   * operands: row, s_alpha[2], s_old_acc[2], s_coeff[32][2], d_out[2].   wires 0,1 out; 2,3 alpha; 4,5 old acc; 6..69 coefficients; 70.. accumulators. */
  MP2G_OP_REDUCING_EXT = 7,
  /* CosetInterpolationGate::with_max_degree(bits, 8) / InterpolationGenerator: the polynomial through 2^bits values on the coset
   * shift <w> evaluated at a point.   operands: row, bits (2..5), s_shift, s_value[2^bits][2], s_point[2], d_out[2].
   * wires 0 shift, 1.. values, then point, out, intermediate (eval, prod) pairs, shifted point (the gate's own layout). */
  MP2G_OP_COSET = 8,
  /* one wire from a slot: ConstantGenerator (ConstantGate wire j = the row's constant j), the PublicInputGate's four
   * public-inputs-hash wires, RandomAccessGate's extra constants, CopyGenerator targets that no other instruction writes.
   * operands: row, col (< 135), s_value. */
  MP2G_OP_WIRE = 9,
  /* QuotientGeneratorExtension: q = num / den in the extension (0 when den = 0: also EqualityGenerator's / NonzeroTestGenerator's
   * "inverse or zero" with num = (1, 0)).   operands: s_num[2], s_den[2], d_q[2].   no wires. */
  MP2G_OP_HINT_DIV_EXT = 10,
  /* the two halves of split_le of a full field element: d = s & (2^63 - 1) / d = s >> 63.   operands: s, d.   no wires. */
  MP2G_OP_HINT_LO63 = 11,
  MP2G_OP_HINT_HI = 12,
  /* LowHighGenerator / SplitToU32Generator (bit = 32): low = s & (2^bit - 1), high = s >> bit.
   * operands: s, bit (1..63), d_low, d_high.   no wires. */
  MP2G_OP_HINT_SPLIT = 13,
  /* a parallel region (rule 4): operands: n_sections (<= 4096), length[n_sections] in words; the sections follow back to back.
   * Host replay with fewer proofs than threads runs the sections on spare threads; the device replay ignores the marker. */
  MP2G_OP_PAR = 14,
  /* PoseidonGate / PoseidonGenerator (the original permutation, WrapC): operands and wires as MP2G_OP_P2. */
  MP2G_OP_POSEIDON = 15,
  /* U32ArithmeticGate with `ops` operations a row / U32ArithmeticGenerator: m0 m1 + addend = low + 2^32 high, all u32.
   * operands: row, i (< ops), ops (1..3), s_m0, s_m1, s_addend, d_low, d_high.   wires 6i .. 6i+5 = m0, m1, addend, low, high,
   * (2^32 - 1 - high)^-1 or 0; the 32 two-bit limbs of the 64-bit result at wires 6 ops + 32 i .. */
  MP2G_OP_U32_ARITH = 16,
  /* U32SubtractionGate / U32SubtractionGenerator: x - y - borrow_in = result - 2^32 borrow_out.
   * operands: row, i (< ops), ops (1..6), s_x, s_y, s_borrow_in, d_result, d_borrow_out.   wires 5i .. 5i+4 in that order; the 16
   * two-bit limbs of result at wires 5 ops + 16 i .. */
  MP2G_OP_U32_SUB = 17,
  /* U32AddManyGate(num_addends, ops) / U32AddManyGenerator: sum of the addends + carry_in = result + 2^32 carry_out.
   * operands: row, i (< ops), ops, n (addends, 1..16), s_addend[n], s_carry_in, d_result, d_carry_out.   wires (n + 3) i .. =
   * addends, carry_in, result, carry_out; 18 two-bit limbs (16 of result, 2 of carry_out) at wires (n + 3) ops + 18 i .. */
  MP2G_OP_U32_ADD_MANY = 18,
  /* U32RangeCheckGate(k) / U32RangeCheckGenerator: x < 2^32.   operands: row, i (< k), k (1..7), s_x.
   * wire i = x, its 16 two-bit limbs at wires k + 16 i .. ; no slot is written. */
  MP2G_OP_U32_RANGE_CHECK = 19,
  /* ComparisonGate(num_bits, num_chunks) / ComparisonGenerator: result = (first <= second), both below 2^num_bits.
   * operands: row, num_bits (1..63), num_chunks (1..16, chunks of ceil(num_bits / num_chunks) bits), s_first, s_second, d_result.
   * wires 0 first, 1 second, 2 result, 3 most significant difference, then per chunk: first chunks, second chunks, equality
   * dummies, chunks-equal flags, intermediate values, then the chunk_bits + 1 bits of 2^chunk_bits + most significant difference. */
  MP2G_OP_COMPARISON = 20,
  /* BaseSumGate<B> with n limbs, B = 2^base_bits / BaseSplitGenerator<B>: the base-B digits of x, little endian.
   * operands: row, base_bits (1 or 2), n (1..63, n base_bits <= 63), s_x, d_limb[n].   wire 0 = x, wires 1..n = limbs. */
  MP2G_OP_BASE_SPLIT = 21,
  /* MulExtensionGate (13 operations a row) / MulExtensionGenerator: out = k_c0 m0 m1 over the extension.
   * operands: row, i (< 13), k_c0, s_m0[2], s_m1[2], d_out[2].   wires 6i .. 6i+5 in that order. */
  MP2G_OP_MUL_EXT = 22,
  /* ExponentiationGate(n) / ExponentiationGenerator: out = base^(sum bit_j 2^j).
   * operands: row, n (1..66), s_base, s_bit[n] (little endian), d_out.   wire 0 base, wires 1..n bits, wire n + 1 out, wires n + 2 ..
   * the n intermediate values (most significant bit first). */
  MP2G_OP_EXP = 23,
  MP2G_OP_END = 24 /* one past the last opcode */
};
/* create: the tape is copied. input_sids [n_inputs]: the slots the caller provides per proof, in the order of the proof's input
 * words (a framework circuit: the circuit-set digest, then per verified child its verifier data, public inputs, caps, openings, FRI
 * proof words and set-membership path, then the circuit's own inputs -- recursion.py universal_inputs); const_slots [n_consts][2]:
 * (slot, value) pairs. */
typedef struct mp2g_witness_program mp2g_witness_program;
int mp2g_witness_program_create(const uint64_t* tape, size_t tape_len, uint32_t n_slots, uint32_t log_n, const uint32_t* input_sids,
                                uint32_t n_inputs, const uint64_t* const_slots, uint32_t n_consts, mp2g_witness_program** out);
uint32_t mp2g_witness_program_num_inputs(const mp2g_witness_program* p);
/* The same replay ON THE DEVICE, for a batch of proofs of one circuit, stream ordered on ctx's stream and without a host copy of
 * anything: one block per proof walks the program's dependency levels (csrc/witness_dev.hip). d_inputs [batch][n_inputs] and
 * d_wires [batch][135][2^log_n] (zero-filled here, then written: what mp2g_prover_prove_dev takes as d_values[0]) are device
 * pointers; d_probe_out [batch][n_probe] receives the slots registered with set_probe (the public-inputs hash and the public inputs,
 * which prove() and the parent's witness need), set once before the first device run. Bit-identical to the host replay. */
uint32_t mp2g_witness_program_num_levels(const mp2g_witness_program* p);
int mp2g_witness_program_set_probe(mp2g_witness_program* p, const uint32_t* probe_sids, uint32_t n_probe);
int mp2g_witness_program_run_dev(mp2g_witness_program* p, mp2g_ctx* ctx, const uint64_t* d_inputs, uint32_t batch, uint64_t* d_wires,
                                 uint64_t* d_probe_out);
/* inputs [batch][n_inputs] canonical field elements -> wires [batch][135][2^log_n] (host memory, the layout of
 * mp2g_prover_prove_dev's d_values[0]); `threads` host threads (0 = all), one proof per thread at a time.
 * probe_sids (may be NULL): slots whose final values are also returned, probe_out [batch][n_probe] -- the circuit's
 * public inputs and their hash, which prove() needs next to the wires. */
int mp2g_witness_program_run(const mp2g_witness_program* p, const uint64_t* inputs, uint32_t batch, uint32_t threads, uint64_t* wires,
                             const uint32_t* probe_sids, uint32_t n_probe, uint64_t* probe_out);
/* The same with one contiguous row of 135 wires per gate row: rows [batch][2^log_n][135]. This is what the host writes fastest (a
 * Poseidon2Gate row is 135 consecutive words instead of 135 cache lines: -25 % per witness); mp2g_wires_from_rows_dev turns the uploaded
 * rows into the prover's [batch][135][2^log_n] on the device. */
int mp2g_witness_program_run_rows(const mp2g_witness_program* p, const uint64_t* inputs, uint32_t batch, uint32_t threads, uint64_t* rows,
                                  const uint32_t* probe_sids, uint32_t n_probe, uint64_t* probe_out);
int mp2g_wires_from_rows_dev(mp2g_ctx* ctx, const uint64_t* d_rows, uint64_t* d_wires, uint32_t log_n, uint32_t num_wires, uint32_t batch);
void mp2g_witness_program_free(mp2g_witness_program* p);

/* ---- generate_proof for a batch of nodes of one framework circuit, on the device (csrc/chain.hip) ------------------------------
 * Replaces the bodies of CircuitWithUniversalVerifier::generate_proof (recursion-framework/src/circuit_builder.rs:286-311) and
 * WrapCircuit::wrap_proof (universal_verifier_gadget/wrap_circuit.rs:122-148): step 0 = the base circuit, steps 1.. = its wrap
 * circuits, each given as its prover (mp2g_prover with the circuit's preprocessed polynomials, gate table, permutation / quotient /
 * witness check switched on), its witness program (probe set: public-inputs hash, then the public inputs), its FRI parameters and
 * its circuit digest (device pointer, 4 words). `capacity` = the widest batch; every narrower one runs in the same buffers.
 * mp2g_chain_run: inputs [batch][n_inputs of step 0] (host) = the base circuit's witness inputs (circuit-set digest, per child the
 * verifier data, the proof, the membership proof; the circuit's own inputs); patches copy child proofs that already live on the
 * device into them in place. Per step: witness replay, prove(), gather of the next step's inputs -- all queued on the context's
 * stream, one synchronisation at the end. Outputs (host, may be NULL): the LAST step's caps [batch][4][cap words], openings
 * [batch][n_openings][2], FRI proof [batch][proof_words], public inputs [batch][n_pi]. A witness that violates a constraint makes
 * the call fail when the provers' witness check is on, as plonky2's prove() panics. */
typedef struct mp2g_chain mp2g_chain;
typedef struct { uint32_t job; uint32_t offset; uint32_t n_words; uint32_t pad_; const uint64_t* d_src; } mp2g_chain_patch;
int mp2g_chain_create(mp2g_ctx* ctx, uint32_t n_steps, mp2g_prover* const* provers, mp2g_witness_program* const* programs,
                      const mp2g_fri_params* params, const uint64_t* const* d_circuit_digests, uint32_t capacity, mp2g_chain** out);
int mp2g_chain_run(mp2g_chain* chain, const uint64_t* inputs, uint32_t batch, const mp2g_chain_patch* patches, uint32_t n_patches,
                   uint64_t* caps, uint64_t* openings, uint64_t* proof, uint64_t* public_inputs);
/* the device buffers of a step after a run (a checker downloads the wires it re-proves; any pointer may be NULL) */
int mp2g_chain_step_buffers(const mp2g_chain* chain, uint32_t step, uint64_t** d_wires, uint64_t** d_probe, uint64_t** d_caps,
                            uint64_t** d_openings, uint64_t** d_proof);
/* proof b of the last run where the chain left it: address and length (words) of its public inputs, its three proof caps, its openings
 * and its FRI proof words -- what a parent's mp2g_chain_patch entries (or a send to another rank) take; valid until the next run */
int mp2g_chain_device_proof(const mp2g_chain* chain, uint32_t b, const uint64_t* d_parts[4], uint32_t n_words[4]);
void mp2g_chain_free(mp2g_chain* chain);

/* ---- a forest of framework proofs: the native scheduler of a tree build (csrc/forest.hip) -------------------------------------
 * Replaces the harness loops that prove every node of a tree children-before-parents, one RecursiveCircuits::generate_proof per
 * node over its children's proofs (mp2-v1/tests/common/celltree.rs:54-189, rowtree.rs:78-337; recursion-framework/src/framework.rs)
 * for a host that hands whole blocks of rows to one GPU. Circuits are described once: the words of the base circuit's witness
 * inputs, how many child proofs a node takes and where each child's proof words lie in the inputs (a child's range = public inputs,
 * the three proof caps, openings, FRI words: what mp2g_chain_device_proof lists), and n_const = the remaining words (circuit-set
 * digest, the children's verifier data and membership proofs, the circuit's own inputs) in input order with the child ranges cut
 * out. chains[worker * n_circuits + circuit]: one mp2g_chain per circuit on every worker's context (NULL = this worker never gets
 * that circuit). Final proofs live in a device pool of `pool_slots` slots of `slot_words` words; a node's slot is returned when its
 * parent is proved unless the node was registered with keep != 0, or on mp2g_forest_release.
 * mp2g_forest_prove: units[u] = unit_nodes[unit_offsets[u] .. unit_offsets[u+1]) are proved by the workers in parallel, each unit
 * level by level (levels counted inside the unit), every level's nodes of one circuit in batches of the chain's capacity. The units
 * of ONE call must not depend on each other (the items of one wave of an update plan do not); children outside a unit must have
 * been proved by an earlier call. A witness that violates a constraint fails the call as plonky2's prove() panics. */
typedef struct mp2g_forest mp2g_forest;
typedef struct {
  uint32_t n_inputs;         /* witness inputs of the base circuit (chain step 0), in words */
  uint32_t n_children;       /* child proofs of a node, <= 4 */
  uint32_t child_offset[4];  /* word offset of each child's proof inside the inputs, increasing */
  uint32_t n_const;          /* the node's own words: everything outside the children's ranges */
} mp2g_forest_circuit;
int mp2g_forest_create(uint32_t n_workers, mp2g_ctx* const* ctxs, uint32_t n_circuits, const mp2g_forest_circuit* circuits,
                       mp2g_chain* const* chains, uint32_t slot_words, uint32_t pool_slots, mp2g_forest** out);
int mp2g_forest_add_nodes(mp2g_forest* f, uint32_t circuit, uint32_t count, const uint64_t* ids, const uint64_t* child_ids /* [count][n_children] */,
                          const uint64_t* consts /* [count][n_const] */, const uint8_t* keep /* [count] or NULL */);
/* How mp2g_forest_prove_plan cuts one wave of Ready items into units (a host that drives mp2g_forest_prove itself can use the same
 * cut): the items in order and whole, units of about group_nodes plan nodes (item i counts item_sizes[i]), at least as many units as
 * workers, shrinking towards the end of the wave (half of what is left per worker, never below group_nodes / 6) so that the last
 * units do not run beside idle workers. unit u = items [unit_first_item[u], unit_first_item[u + 1]); unit_first_item has room for
 * n_items + 1 entries. Pure host arithmetic: needs no GPU. */
int mp2g_forest_group_units(const uint32_t* item_sizes, uint32_t n_items, uint32_t n_workers, uint32_t group_nodes,
                            uint32_t* unit_first_item, uint32_t* n_units);
/* (one mp2g_forest_prove / mp2g_forest_prove_plan call at a time per forest: the call starts the forest's worker threads itself.
 * Inside it a worker keeps up to two batches queued behind the running one; a node counts as proved, and its children's pool
 * slots return, when its batch has been CONFIRMED -- witness flags read, one batch late; a failure rolls the queued batches back.
 * A worker that finds the pool empty waits for slots and fails only when every worker of the call waits. A node is PUBLISHED --
 * visible to mp2g_forest_proof and usable as a child by another unit -- only when its batch is confirmed; inside its own unit it
 * is usable as soon as it is queued (same stream). The units of one call must therefore not name each other's nodes as children
 * (the waves of an update plan never do). After a failed call the nodes of the confirmed batches stay proved: resubmit the unit
 * without them -- a unit that names a proved node is refused.) */
int mp2g_forest_prove(mp2g_forest* f, const uint64_t* unit_nodes, const uint32_t* unit_offsets /* [n_units + 1] */, uint32_t n_units);
/* the harness loop over an update plan (declared below), inside the library: drain the Ready items of a wave, group them into units of
 * about group_nodes plan nodes (never fewer units than workers), prove, mark done, until the plan is finished. A plan node k stands for
 * forest node k and its n_satellites satellite nodes ((j + 1) << satellite_shift) | k (a row and that row's cells-tree nodes).
 * waves / items_per_wave[max_waves] (may be NULL) receive the number of waves and the items of each. */
struct mp2g_update_plan;
int mp2g_forest_prove_plan(mp2g_forest* f, struct mp2g_update_plan* plan, uint32_t group_nodes, uint32_t n_satellites, uint32_t satellite_shift,
                           uint32_t* waves, uint32_t* items_per_wave, uint32_t max_waves);
/* a proved node's final proof: its words in a parent's input order (public inputs, caps of oracles 1..3, openings, FRI words) on the
 * host (words may be NULL to ask for the length) or where they lie on the device (valid until the slot is returned) */
int mp2g_forest_proof(mp2g_forest* f, uint64_t id, uint64_t* words, uint32_t* n_words);
int mp2g_forest_device_proof(mp2g_forest* f, uint64_t id, const uint64_t** d_words, uint32_t* n_words);
int mp2g_forest_release(mp2g_forest* f, uint64_t id);
uint64_t mp2g_forest_proved(const mp2g_forest* f);
uint32_t mp2g_forest_free_slots(mp2g_forest* f);
void mp2g_forest_free(mp2g_forest* f);

/* ---- work plan: the reference's only scheduler --------------------------------------
 * Host-side, no GPU involved. Replaces ryhope/src/storage/updatetree.rs: UpdateTree (:19-242, arena
 * of nodes, node 0 = root, children ordered by arena index) and UpdatePlan (:422-541). Keys are u64
 * (row / cell / block keys); a path runs root -> node. An item handed out by the plan is the unit one
 * GPU proves locally: a single node (subtree_size == 1) or a spun-off subtree whose root proof is the
 * only thing returned. */
typedef struct mp2g_update_tree mp2g_update_tree;
typedef struct mp2g_update_plan mp2g_update_plan;
/* UpdateTree::from_paths (:80-92): paths concatenated in `keys`, path i has path_lens[i] keys.
 * n_paths == 0 gives the empty tree. Fails where the reference panics: an empty path, a key that
 * occurs twice at different positions, a path whose first key is not the root. */
int mp2g_update_tree_from_paths(const uint64_t* keys, const uint32_t* path_lens, uint32_t n_paths, int64_t epoch,
                                mp2g_update_tree** out);
/* UpdateTree::from_map (:296-331): the hierarchy described by a map key -> NodeContext {left, right} (arrays of n
 * entries; has_left / has_right say whether the option is Some), walked pre-order from root. Child keys missing
 * from the map are skipped; is_path_end = NodeContext::is_leaf. A key reached twice is the reference's panic. */
int mp2g_update_tree_from_map(const uint64_t* keys, const uint64_t* left, const uint64_t* right, const uint8_t* has_left,
                              const uint8_t* has_right, uint32_t n, uint64_t root, int64_t epoch, mp2g_update_tree** out);
/* UpdateTree::extend_with_path (:145-151) */
int mp2g_update_tree_extend_with_path(mp2g_update_tree* t, const uint64_t* path, uint32_t len);
uint32_t mp2g_update_tree_size(const mp2g_update_tree* t);
int64_t mp2g_update_tree_epoch(const mp2g_update_tree* t);
int mp2g_update_tree_contains_key(const mp2g_update_tree* t, uint64_t k);
/* arena dump: keys[size], parents[size] (-1 for a root), is_path_end[size]; any pointer may be NULL */
int mp2g_update_tree_nodes(const mp2g_update_tree* t, uint64_t* keys, int32_t* parents, uint8_t* is_path_end);
/* UpdateTree::subtree_size_i (:171-179) by key */
int mp2g_update_tree_subtree_size(const mp2g_update_tree* t, uint64_t k, uint32_t* out);
void mp2g_update_tree_free(mp2g_update_tree* t);
/* into_workplan (subtree_size 1) / into_batched_workplan (:154-163); the tree is consumed (the
 * handle stays valid only for mp2g_update_tree_free, which the plan performs). */
int mp2g_update_plan_create(mp2g_update_tree* t, uint32_t subtree_size, mp2g_update_plan** out);
#define MP2G_PLAN_FINISHED 0 /* Iterator::next() == None           */
#define MP2G_PLAN_READY 1    /* Some(Next::Ready(item))            */
#define MP2G_PLAN_NOT_YET 2  /* Some(Next::NotYet)                 */
/* Iterator::next (:517-531). On READY *k is the item's key; with subtree_size == 1 the item is
 * WorkplanItem::Node and *is_path_end is set (*subtree = NULL); otherwise it is
 * WorkplanItem::Subtree and *subtree receives the spun-off tree (caller frees). Returns one of
 * the MP2G_PLAN_* states, or -1 on invalid arguments. */
int mp2g_update_plan_next(mp2g_update_plan* p, uint64_t* k, int* is_path_end, mp2g_update_tree** subtree);
/* UpdatePlan::done (:449-467); unknown key -> error (RyhopeError::KeyNotFound) */
int mp2g_update_plan_done(mp2g_update_plan* p, uint64_t k);
int mp2g_update_plan_completed(const mp2g_update_plan* p);
void mp2g_update_plan_free(mp2g_update_plan* p);

#ifdef __cplusplus
}
#endif
#endif
