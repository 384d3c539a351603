// TEST INFRASTRUCTURE -- CPU oracle (see gl.h header). Never linked into the product.
//
// Poseidon-12 / Poseidon2-12 permutations over Goldilocks, plonky2's sponge helpers and
// MerkleTree with cap.
// Follows (absent [dep] sources, restated from the published algorithms):
//   plonky2/src/hash/poseidon.rs           (full/partial round structure, naive form)
//   plonky2/src/hash/poseidon_goldilocks.rs (MDS circ/diag; test vectors checked in tests/)
//   poseidon2_plonky2 poseidon2_hash.rs / poseidon2_goldilock.rs (HorizenLabs t=12 instance)
//   plonky2/src/hash/hashing.rs            (hash_n_to_m_no_pad, compress)
//   plonky2/src/plonk/config.rs            (Hasher::{hash_no_pad,hash_pad,hash_or_noop,two_to_one})
//   plonky2/src/hash/merkle_tree.rs, merkle_proofs.rs
// In-tree restatements followed exactly: mp2-common/src/hash.rs:24-45 and
//   mp2-common/src/poseidon.rs:151-170 (overwrite-mode sponge, squeeze from the front of the rate).
#include "gl.h"
#include "constants.h"
#include <stdlib.h>

#define W 12
#define RATE 8

// variant: 0 = Poseidon2 (default C of the reference, mp2-common/src/lib.rs:40), 1 = Poseidon
void orc_poseidon_perm(gl_t s[W]) {
  for (int r = 0; r < 30; r++) {
    for (int i = 0; i < W; i++) s[i] = gl_add(s[i], POSEIDON_RC[12 * r + i]);
    if (r < 4 || r >= 26) {
      for (int i = 0; i < W; i++) s[i] = gl_pow7(s[i]);
    } else {
      s[0] = gl_pow7(s[0]);
    }
    gl_t t[W];
    for (int row = 0; row < W; row++) {
      unsigned __int128 acc = 0;  // 12 * 2^64 * 41 < 2^128
      for (int i = 0; i < W; i++) acc += (unsigned __int128)s[(i + row) % W] * POSEIDON_MDS_CIRC[i];
      acc += (unsigned __int128)s[row] * POSEIDON_MDS_DIAG[row];
      t[row] = gl_reduce128(acc);
    }
    memcpy(s, t, sizeof t);
  }
}

static void p2_ext(gl_t s[W]) {
  // circ(2*M4, M4, M4), M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
  static const unsigned M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
  gl_t t[W];
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < 4; i++) {
      unsigned __int128 acc = 0;
      for (int j = 0; j < 4; j++) acc += (unsigned __int128)s[4 * c + j] * M4[i][j];
      t[4 * c + i] = gl_reduce128(acc);
    }
  for (int i = 0; i < 4; i++) {
    gl_t sum = gl_add(gl_add(t[i], t[4 + i]), t[8 + i]);
    for (int c = 0; c < 3; c++) s[4 * c + i] = gl_add(t[4 * c + i], sum);
  }
}
void orc_poseidon2_perm(gl_t s[W]) {
  p2_ext(s);
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < W; i++) s[i] = gl_pow7(gl_add(s[i], POSEIDON2_RC_EXT[12 * r + i]));
    p2_ext(s);
  }
  for (int r = 0; r < 22; r++) {
    s[0] = gl_pow7(gl_add(s[0], POSEIDON2_RC_INT[r]));
    gl_t sum = 0;
    for (int i = 0; i < W; i++) sum = gl_add(sum, s[i]);
    for (int i = 0; i < W; i++) s[i] = gl_add(gl_mul(s[i], POSEIDON2_DIAG_M1[i]), sum);
  }
  for (int r = 4; r < 8; r++) {
    for (int i = 0; i < W; i++) s[i] = gl_pow7(gl_add(s[i], POSEIDON2_RC_EXT[12 * r + i]));
    p2_ext(s);
  }
}
void orc_perm(int variant, gl_t s[W]) {
  if (variant == 0) orc_poseidon2_perm(s); else orc_poseidon_perm(s);
}

// hashing.rs hash_n_to_m_no_pad: overwrite-mode absorb, squeeze rate-first.
void orc_hash_n_to_m_no_pad(int variant, const gl_t* in, size_t n, gl_t* out, size_t m) {
  gl_t s[W] = {0};
  for (size_t i = 0; i < n; i += RATE) {
    size_t k = n - i < RATE ? n - i : RATE;
    memcpy(s, in + i, k * sizeof(gl_t));
    orc_perm(variant, s);
  }
  size_t o = 0;
  for (;;) {
    for (int i = 0; i < RATE; i++) {
      out[o++] = s[i];
      if (o == m) return;
    }
    orc_perm(variant, s);
  }
}
void orc_hash_no_pad(int variant, const gl_t* in, size_t n, gl_t out[4]) {
  orc_hash_n_to_m_no_pad(variant, in, n, out, 4);
}
// config.rs Hasher::hash_pad: append 1, zero-fill to rate-1 mod rate, append 1.
void orc_hash_pad(int variant, const gl_t* in, size_t n, gl_t out[4]) {
  size_t len = n + 1;
  while ((len + 1) % RATE) len++;
  len++;
  gl_t* buf = calloc(len, sizeof(gl_t));
  memcpy(buf, in, n * sizeof(gl_t));
  buf[n] = 1;
  buf[len - 1] = 1;
  orc_hash_no_pad(variant, buf, len, out);
  free(buf);
}
// config.rs Hasher::hash_or_noop: <= 4 limbs are used verbatim (zero-padded).
void orc_hash_or_noop(int variant, const gl_t* in, size_t n, gl_t out[4]) {
  if (n <= 4) {
    memset(out, 0, 4 * sizeof(gl_t));
    memcpy(out, in, n * sizeof(gl_t));
  } else {
    orc_hash_no_pad(variant, in, n, out);
  }
}
// hashing.rs compress: perm([l || r || 0000])[0..4]
void orc_two_to_one(int variant, const gl_t l[4], const gl_t r[4], gl_t out[4]) {
  gl_t s[W] = {0};
  memcpy(s, l, 32);
  memcpy(s + 4, r, 32);
  orc_perm(variant, s);
  memcpy(out, s, 32);
}
void orc_hash_no_pad_batch(int variant, const gl_t* in, size_t in_len, size_t count, size_t out_len, gl_t* out) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < count; i++) orc_hash_n_to_m_no_pad(variant, in + i * in_len, in_len, out + i * out_len, out_len);
}

// merkle_tree.rs: leaves -> hash_or_noop -> pairwise two_to_one up to the cap.
// `levels` receives level 0 (leaf digests, L*4), level 1 (L/2*4) ... cap level (2^cap_h * 4),
// concatenated; total (2L - 2^cap_h) * 4 limbs. (plonky2's own in-memory `digests` order is a
// recursive split layout that only matters for params files, SURVEY App. B "Merkle".)
void orc_merkle_build(int variant, const gl_t* leaves, size_t leaf_len, unsigned log_leaves, unsigned cap_h, gl_t* levels) {
  size_t L = (size_t)1 << log_leaves;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < L; i++) orc_hash_or_noop(variant, leaves + i * leaf_len, leaf_len, levels + 4 * i);
  gl_t* cur = levels;
  for (unsigned lv = log_leaves; lv > cap_h; lv--) {
    size_t n = (size_t)1 << lv;
    gl_t* nxt = cur + 4 * n;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n / 2; i++) orc_two_to_one(variant, cur + 8 * i, cur + 8 * i + 4, nxt + 4 * i);
    cur = nxt;
  }
}
size_t orc_merkle_levels_len(unsigned log_leaves, unsigned cap_h) {
  return 4 * ((((size_t)1 << log_leaves) << 1) - ((size_t)1 << cap_h));
}
const gl_t* orc_merkle_cap_ptr(const gl_t* levels, unsigned log_leaves, unsigned cap_h) {
  return levels + orc_merkle_levels_len(log_leaves, cap_h) - 4 * ((size_t)1 << cap_h);
}
// merkle_tree.rs prove(): siblings bottom-up, log_leaves - cap_h of them.
void orc_merkle_prove(const gl_t* levels, unsigned log_leaves, unsigned cap_h, size_t idx, gl_t* siblings) {
  const gl_t* cur = levels;
  for (unsigned lv = log_leaves; lv > cap_h; lv--) {
    memcpy(siblings, cur + 4 * (idx ^ 1), 32);
    siblings += 4;
    cur += 4 * ((size_t)1 << lv);
    idx >>= 1;
  }
}
// merkle_proofs.rs verify_merkle_proof_to_cap
int orc_merkle_verify(int variant, const gl_t* leaf, size_t leaf_len, size_t idx, const gl_t* siblings,
                      unsigned n_sib, const gl_t* cap) {
  gl_t cur[4];
  orc_hash_or_noop(variant, leaf, leaf_len, cur);
  for (unsigned i = 0; i < n_sib; i++) {
    gl_t nx[4];
    if (idx & 1) orc_two_to_one(variant, siblings + 4 * i, cur, nx);
    else orc_two_to_one(variant, cur, siblings + 4 * i, nx);
    memcpy(cur, nx, 32);
    idx >>= 1;
  }
  return memcmp(cur, cap + 4 * idx, 32) == 0;
}
