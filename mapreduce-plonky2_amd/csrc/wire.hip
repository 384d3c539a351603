// Proof wire format (host code): bincode 1.3 layout of plonky2's ProofWithPublicInputs / the
// reference's ProofWithVK, as moved between tree levels (mp2-common/src/proof.rs:42-57,84-98;
// verifiable-db/src/cells_tree/api.rs:188-232 deserializes children and serializes the parent).
// serde field order [dep]: Proof{wires_cap, plonk_zs_partial_products_cap, quotient_polys_cap,
// openings{constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products,
// quotient_polys, lookup_zs, lookup_zs_next}, opening_proof{commit_phase_merkle_caps,
// query_round_proofs[{initial_trees_proof{evals_proofs[(Vec<F>, MerkleProof)]}, steps[{evals,
// merkle_proof}]}], final_poly, pow_witness}}, public_inputs.
#include "ctx.h"
#include <cstring>

using namespace mp2g;
#define NEED(c, msg) do { if (!(c)) return fail("invalid argument: %s", msg); } while (0)

namespace {
struct Writer {
  uint8_t* out;
  size_t pos = 0;
  void u64s(const uint64_t* p, size_t n) {
    if (out) memcpy(out + pos, p, n * 8);
    pos += n * 8;
  }
  void len(uint64_t n) { u64s(&n, 1); }
  void bytes(const uint8_t* p, size_t n) {
    if (out) memcpy(out + pos, p, n);
    pos += n;
  }
};
struct Reader {
  const uint8_t* in;
  size_t n, pos = 0;
  bool ok = true;
  void u64s(uint64_t* p, size_t k) {
    if (!ok || pos + k * 8 > n) { ok = false; return; }
    memcpy(p, in + pos, k * 8);
    for (size_t i = 0; i < k; i++) if (p[i] >= GL_P) ok = false;  // canonical field elements only
    pos += k * 8;
  }
  void expect_len(uint64_t want) {
    uint64_t v = 0;
    if (!ok || pos + 8 > n) { ok = false; return; }
    memcpy(&v, in + pos, 8);
    pos += 8;
    if (v != want) ok = false;
  }
};
struct Layout {
  uint32_t lg, depth, n_layers;
  size_t capw, cap_n, final_len, q_words;
};
Layout layout(const mp2g_fri_params* p) {
  Layout l;
  l.lg = p->log_n + p->rate_bits;
  l.depth = l.lg - p->cap_height;
  l.n_layers = p->n_layers;
  l.cap_n = (size_t)1 << p->cap_height;
  l.capw = 4 * l.cap_n;
  uint32_t deg = p->log_n;
  for (uint32_t i = 0; i < p->n_layers; i++) deg -= p->arity_bits[i];
  l.final_len = (size_t)1 << deg;
  l.q_words = (mp2g_fri_proof_words(p) - p->n_layers * l.capw - 2 * l.final_len - 1) / (p->num_queries ? p->num_queries : 1);
  return l;
}
// walks the structure once; T is Writer (serialize) or Reader (deserialize)
template <class T, class U64P>
void walk(T& io, const mp2g_fri_params* p, uint32_t num_constants, U64P caps, U64P openings, U64P fri, U64P pis, uint32_t n_pis,
          void (*vec_len)(T&, uint64_t)) {
  Layout l = layout(p);
  for (uint32_t o = 1; o < p->n_oracles; o++) {  // wires, zs_partial_products, quotient caps
    vec_len(io, l.cap_n);
    io.u64s(caps + o * l.capw, l.capw);
  }
  // OpeningSet
  size_t off = 0;
  auto ext_vec = [&](size_t count) { vec_len(io, count); io.u64s(openings + 2 * off, 2 * count); off += count; };
  size_t n_zeta = 0;
  for (uint32_t o = 0; o < p->n_oracles; o++) n_zeta += p->oracle_w[o];
  // FRI batch order in `openings`: consts+sigmas | wires | zs + partial products | quotient | lookup | zs_next | lookup_next
  // serde order: constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys, lookup_zs, lookup_zs_next
  const size_t L = (size_t)p->zs_count * p->num_lookup_polys;
  size_t o0 = 0, o1 = p->oracle_w[0], o2 = o1 + (p->n_oracles > 1 ? p->oracle_w[1] : 0);
  size_t o3 = o2 + (p->n_oracles > 2 ? p->oracle_w[2] - L : 0);
  size_t o_lu = o3 + (p->n_oracles > 3 ? p->oracle_w[3] : 0);
  (void)o0;
  off = 0; ext_vec(num_constants);
  ext_vec(p->oracle_w[0] - num_constants);
  off = o1; ext_vec(p->n_oracles > 1 ? p->oracle_w[1] : 0);
  off = o2; ext_vec(p->zs_count);
  off = n_zeta; ext_vec(p->zs_count);
  off = o2 + p->zs_count; ext_vec((p->n_oracles > 2 ? p->oracle_w[2] - L : 0) - p->zs_count);
  off = o3; ext_vec(p->n_oracles > 3 ? p->oracle_w[3] : 0);
  off = o_lu; ext_vec(L);                  // lookup_zs
  off = n_zeta + p->zs_count; ext_vec(L);  // lookup_zs_next
  // FriProof
  vec_len(io, l.n_layers);
  for (uint32_t i = 0; i < l.n_layers; i++) { vec_len(io, l.cap_n); io.u64s(fri + i * l.capw, l.capw); }
  vec_len(io, p->num_queries);
  U64P q = fri + l.n_layers * l.capw;
  for (uint32_t r = 0; r < p->num_queries; r++) {
    U64P o = q + r * l.q_words;
    vec_len(io, p->n_oracles);
    for (uint32_t oi = 0; oi < p->n_oracles; oi++) {
      vec_len(io, p->oracle_w[oi]); io.u64s(o, p->oracle_w[oi]); o += p->oracle_w[oi];
      vec_len(io, l.depth); io.u64s(o, 4 * l.depth); o += 4 * l.depth;
    }
    vec_len(io, l.n_layers);
    uint32_t clg = l.lg;
    for (uint32_t i = 0; i < l.n_layers; i++) {
      size_t a = (size_t)1 << p->arity_bits[i];
      clg -= p->arity_bits[i];
      vec_len(io, a); io.u64s(o, 2 * a); o += 2 * a;
      uint32_t d = clg - p->cap_height;
      vec_len(io, d); io.u64s(o, 4 * d); o += 4 * d;
    }
  }
  U64P fin = q + (size_t)p->num_queries * l.q_words;
  vec_len(io, l.final_len); io.u64s(fin, 2 * l.final_len);
  io.u64s(fin + 2 * l.final_len, 1);  // pow_witness
  vec_len(io, n_pis); io.u64s(pis, n_pis);
}
void wlen(Writer& w, uint64_t n) { w.len(n); }
void rlen(Reader& r, uint64_t n) { r.expect_len(n); }
int shape_check(const mp2g_fri_params* p, uint32_t num_constants) {
  int rc = params_check(p);  // cap_height / arity / layer bounds: the layout below subtracts them unchecked
  if (rc) return rc;
  NEED(num_constants <= p->oracle_w[0], "num_constants <= oracle_w[0]");
  NEED(p->zs_oracle == 2 || p->zs_count == 0, "wire format expects the Z polynomials in oracle 2");
  NEED(p->n_oracles <= 4, "wire format has four oracles");
  NEED(p->n_oracles <= 2 || (uint64_t)p->zs_count * (1 + p->num_lookup_polys) <= p->oracle_w[2], "zs_count / num_lookup_polys");
  NEED(p->num_lookup_polys == 0 || p->n_oracles == 4, "lookup polynomials need the four plonky2 oracles");
  return 0;
}
}  // namespace

extern "C" {
int mp2g_proof_serialize(const mp2g_fri_params* p, uint32_t num_constants, const uint64_t* caps, const uint64_t* openings,
                         const uint64_t* fri_proof, const uint64_t* public_inputs, uint32_t n_pis, uint8_t* out, size_t* out_len) {
  int rc = shape_check(p, num_constants);
  if (rc) return rc;
  NEED(out_len, "out_len");
  NEED(!out || (caps && openings && fri_proof && (public_inputs || !n_pis)), "pointers");
  Writer w{out};
  walk<Writer, const uint64_t*>(w, p, num_constants, caps, openings, fri_proof, public_inputs, n_pis, wlen);
  *out_len = w.pos;
  return 0;
}
int mp2g_proof_deserialize(const mp2g_fri_params* p, uint32_t num_constants, const uint8_t* bytes, size_t len, uint64_t* caps,
                           uint64_t* openings, uint64_t* fri_proof, uint64_t* public_inputs, uint32_t n_pis) {
  int rc = shape_check(p, num_constants);
  if (rc) return rc;
  NEED(bytes && caps && openings && fri_proof && (public_inputs || !n_pis), "pointers");
  Reader r{bytes, len};
  walk<Reader, uint64_t*>(r, p, num_constants, caps, openings, fri_proof, public_inputs, n_pis, rlen);
  if (!r.ok || r.pos != len) return fail("malformed proof bytes (shape mismatch, non-canonical element or trailing data)");
  return 0;
}
int mp2g_proof_with_vk_serialize(const uint8_t* proof_bytes, size_t proof_len, const uint64_t* vk_cap, uint32_t vk_cap_len,
                                 const uint64_t vk_circuit_digest[4], uint8_t* out, size_t* out_len) {
  NEED(out_len && (!out || (proof_bytes && vk_cap && vk_circuit_digest)), "pointers");
  uint32_t height = 0;
  while (((uint32_t)1 << height) < vk_cap_len && height < 31) height++;
  NEED(vk_cap_len >= 1 && ((uint32_t)1 << height) == vk_cap_len, "vk_cap_len must be a power of two");
  Writer w{out};
  w.bytes(proof_bytes, proof_len);
  // [dep] VerifierOnlyCircuitData::to_bytes = write_usize(cap height), the cap's hashes (no length), the digest;
  // serialize_with = serialize wraps it as a byte string (u64 length first)
  uint64_t blob = 8 + (uint64_t)vk_cap_len * 32 + 32;
  w.len(blob);
  w.len(height);
  w.u64s(vk_cap, (size_t)vk_cap_len * 4);
  w.u64s(vk_circuit_digest, 4);
  *out_len = w.pos;
  return 0;
}
int mp2g_proof_with_vk_deserialize(const mp2g_fri_params* p, uint32_t num_constants, const uint8_t* bytes, size_t len, uint64_t* caps,
                                   uint64_t* openings, uint64_t* fri_proof, uint64_t* public_inputs, uint32_t n_pis, uint64_t* vk_cap,
                                   uint32_t vk_cap_len, uint64_t vk_circuit_digest[4]) {
  // ProofWithVK::deserialize (mp2-common/src/proof.rs:54-57): the proof, then the verifier key as a byte string whose
  // content is VerifierOnlyCircuitData::to_bytes (cap height, the cap's hashes, the circuit digest)
  int rc = shape_check(p, num_constants);
  if (rc) return rc;
  NEED(bytes && caps && openings && fri_proof && (public_inputs || !n_pis) && vk_cap && vk_circuit_digest, "pointers");
  uint32_t height = 0;
  while (((uint32_t)1 << height) < vk_cap_len && height < 31) height++;
  NEED(vk_cap_len >= 1 && ((uint32_t)1 << height) == vk_cap_len, "vk_cap_len must be a power of two");
  Reader r{bytes, len};
  walk<Reader, uint64_t*>(r, p, num_constants, caps, openings, fri_proof, public_inputs, n_pis, rlen);
  r.expect_len(8 + (uint64_t)vk_cap_len * 32 + 32);
  r.expect_len(height);
  r.u64s(vk_cap, (size_t)vk_cap_len * 4);
  r.u64s(vk_circuit_digest, 4);
  if (!r.ok || r.pos != len) return fail("malformed ProofWithVK bytes (shape mismatch, non-canonical element, another cap height or trailing data)");
  return 0;
}
}  // extern "C"
