// TEST INFRASTRUCTURE -- CPU oracle (see gl.h header). Never linked into the product.
//
// GF(p^5) square roots / sgn0, the Ecgfp5 group (double-odd curve y^2 = x(x^2 + 2x + 263 z)),
// simplified SWU map-to-curve and the multiset digests built from them.
// Follows, in-tree:
//   mp2-common/src/group_hashing/sswu_value.rs:31-77   simple_swu (exact operation order)
//   mp2-common/src/group_hashing/utils.rs:9-82         2/3, A_sw, B_sw, Z_sw, -1/Z, -B/A
//   mp2-common/src/group_hashing/field_to_curve.rs:36-48  map_to_curve_point
//   mp2-common/src/group_hashing/curve_add.rs:17-33    add_curve_point / add_weierstrass_point
//   mp2-common/src/group_hashing/mod.rs:149-234        11-limb Weierstrass form, field_hashed_scalar_mul
//   mp2-common/src/poseidon.rs:120-133                 hash_to_int_value
//   mp2-v1/src/values_extraction/mod.rs:499-571        row_unique_data, compute_row_id, compute_table_row_digest
//   verifiable-db/src/cells_tree/mod.rs:65-72          Cell::values_digest
// and, absent [dep] plonky2_ecgfp5 @9260603 (curve/{base_field,curve,scalar_field}.rs), the
// published ecgfp5 construction (Pornin): group = E[r] + N, N = (0,0) neutral, law P (+) Q = P+Q+N,
// canonical encoding w = y/x, decode picks the non-square root x of x^2 - (w^2-a)x + b.
// Here the group law is evaluated with textbook affine chord/tangent formulas on E -- slow but
// self-evidently the curve's law; any correct formula set yields the same encodings.
// Pinned by the three known-answer tests of sswu_value.rs:88-118 (tests/golden/sswu_kat.json).
#include "gl.h"
#include <stdlib.h>

void orc_hash_n_to_m_no_pad(int variant, const gl_t* in, size_t n, gl_t* out, size_t m);

// ---- base-field and quintic square roots --------------------------------------------------
static int gl_sqrt(gl_t a, gl_t* out) {  // Tonelli-Shanks, p-1 = 2^32 * (2^32-1)
  if (a == 0) { *out = 0; return 1; }
  if (gl_pow(a, (GL_P - 1) / 2) != 1) return 0;
  const uint64_t q = 0xFFFFFFFFULL;
  unsigned M = 32;
  gl_t c = GL_TWO_GEN, t = gl_pow(a, q), R = gl_pow(a, (q + 1) / 2);
  while (t != 1) {
    unsigned i = 0;
    gl_t t2 = t;
    while (t2 != 1) { t2 = gl_sqr(t2); i++; }
    gl_t b = c;
    for (unsigned j = 0; j + i + 1 < M; j++) b = gl_sqr(b);
    M = i;
    c = gl_sqr(b);
    t = gl_mul(t, c);
    R = gl_mul(R, b);
  }
  *out = R;
  return 1;
}
int orc_gl5_sqrt(const gl_t x_[5], gl_t out_[5]) {
  gl5_t x; memcpy(&x, x_, 40);
  gl5_t v = x;
  for (int i = 0; i < 31; i++) v = gl5_sqr(v);
  gl5_t v32 = v;
  for (int i = 0; i < 32; i++) v32 = gl5_sqr(v32);
  gl5_t d = gl5_mul(gl5_mul(x, v32), gl5_inv(v));          // x^((p+1)/2)
  gl5_t e = gl5_frob(gl5_mul(d, gl5_frob(gl5_frob(d))));   // d^(p+p^3) = x^((r-1)/2)
  gl5_t f = gl5_sqr(e);                                    // x^(r-1)
  gl_t g = gl5_mul(x, f).c[0];                             // x^r in GF(p)
  gl_t s;
  if (!gl_sqrt(g, &s)) return 0;
  gl5_t r = gl5_scale(gl5_inv(e), s);
  memcpy(out_, &r, 40);
  return 1;
}
static int gl5_sqrt(gl5_t x, gl5_t* out) { return orc_gl5_sqrt(x.c, out->c); }
static int gl5_is_square(gl5_t x) {  // legendre != -1 (0 counts as square)
  gl_t n = gl5_norm(x);
  return n == 0 || gl_pow(n, (GL_P - 1) / 2) == 1;
}
// RFC 9380 sgn0 for an extension: parity of the first non-zero coefficient
static int gl5_sgn0(gl5_t x) {
  int sign = 0, zero = 1;
  for (int i = 0; i < 5; i++) {
    int sign_i = (int)(x.c[i] & 1), zero_i = x.c[i] == 0;
    sign = sign || (zero && sign_i);
    zero = zero && zero_i;
  }
  return sign;
}
int orc_gl5_sgn0(const gl_t x[5]) { gl5_t t; memcpy(&t, x, 40); return gl5_sgn0(t); }
void orc_gl5_mul(const gl_t a[5], const gl_t b[5], gl_t o[5]) { gl5_t x, y; memcpy(&x, a, 40); memcpy(&y, b, 40); x = gl5_mul(x, y); memcpy(o, &x, 40); }
void orc_gl5_inv(const gl_t a[5], gl_t o[5]) { gl5_t x; memcpy(&x, a, 40); x = gl5_inv(x); memcpy(o, &x, 40); }

// ---- curve --------------------------------------------------------------------------------
typedef struct { gl5_t x, y; int inf; } ec_t;  // inf: the point at infinity of E (not a group element)
static gl5_t EC_A(void) { return gl5_from(2); }
static gl5_t EC_B(void) { gl5_t b = gl5_zero(); b.c[1] = 263; return b; }
static ec_t ec_N(void) { ec_t p; p.x = gl5_zero(); p.y = gl5_zero(); p.inf = 0; return p; }
static ec_t ec_dbl(ec_t p) {
  if (p.inf || gl5_is_zero(p.y)) { ec_t o = ec_N(); o.inf = 1; return o; }
  gl5_t x2 = gl5_sqr(p.x);
  gl5_t num = gl5_add(gl5_add(gl5_scale(x2, 3), gl5_scale(gl5_mul(EC_A(), p.x), 2)), EC_B());
  gl5_t lam = gl5_mul(num, gl5_inv(gl5_scale(p.y, 2)));
  ec_t r; r.inf = 0;
  r.x = gl5_sub(gl5_sub(gl5_sub(gl5_sqr(lam), EC_A()), p.x), p.x);
  r.y = gl5_sub(gl5_mul(lam, gl5_sub(p.x, r.x)), p.y);
  return r;
}
static ec_t ec_add(ec_t p, ec_t q) {
  if (p.inf) return q;
  if (q.inf) return p;
  if (gl5_eq(p.x, q.x)) {
    if (gl5_eq(p.y, q.y)) return ec_dbl(p);
    ec_t o = ec_N(); o.inf = 1; return o;
  }
  gl5_t lam = gl5_mul(gl5_sub(q.y, p.y), gl5_inv(gl5_sub(q.x, p.x)));
  ec_t r; r.inf = 0;
  r.x = gl5_sub(gl5_sub(gl5_sub(gl5_sqr(lam), EC_A()), p.x), q.x);
  r.y = gl5_sub(gl5_mul(lam, gl5_sub(p.x, r.x)), p.y);
  return r;
}
// group law of ecgfp5: P (+) Q = P + Q + N; neutral N
static ec_t grp_add(ec_t p, ec_t q) { return ec_add(ec_add(p, q), ec_N()); }
static gl5_t grp_encode(ec_t p) { return gl5_mul(p.y, gl5_inv(p.x)); }  // N -> 0
static int grp_decode(gl5_t w, ec_t* out) {
  gl5_t e = gl5_sub(gl5_sqr(w), EC_A());
  gl5_t delta = gl5_sub(gl5_sqr(e), gl5_scale(EC_B(), 4));
  gl5_t r;
  if (!gl5_sqrt(delta, &r)) {
    *out = ec_N();
    return gl5_is_zero(w);
  }
  gl_t half = gl_inv(2);
  gl5_t x1 = gl5_scale(gl5_add(e, r), half), x2 = gl5_scale(gl5_sub(e, r), half);
  gl5_t x = gl5_is_square(x1) ? x2 : x1;
  out->x = x; out->y = gl5_mul(w, x); out->inf = 0;
  return 1;
}
static ec_t grp_mul(ec_t p, const uint32_t* k_le, int n_limbs) {  // scalar as little-endian u32 limbs
  ec_t acc = ec_N();
  for (int i = n_limbs * 32 - 1; i >= 0; i--) {
    acc = grp_add(acc, acc);
    if ((k_le[i / 32] >> (i % 32)) & 1) acc = grp_add(acc, p);
  }
  return acc;
}
// mod.rs:163-174 ToFields for WeierstrassPoint: [x0..x4, y0..y4, is_inf]; [dep] to_weierstrass():
// X = x + a/3, Y = -w*x = -y, neutral -> (0, 0, inf) -- sign convention recalled, see DESIGN.md.
static void grp_to_weierstrass(ec_t p, gl_t out[11]) {
  if (gl5_is_zero(p.x)) { memset(out, 0, 88); out[10] = 1; return; }
  gl5_t two_thirds = gl5_from(6148914689804861441ULL);  // utils.rs:9-17
  gl5_t X = gl5_add(p.x, two_thirds), Y = gl5_neg(p.y);
  memcpy(out, X.c, 40); memcpy(out + 5, Y.c, 40); out[10] = 0;
}
static int on_curve(ec_t p) {
  gl5_t rhs = gl5_mul(p.x, gl5_add(gl5_add(gl5_sqr(p.x), gl5_mul(EC_A(), p.x)), EC_B()));
  return gl5_eq(gl5_sqr(p.y), rhs);
}

// ---- simplified SWU, sswu_value.rs:31-77 ---------------------------------------------------
static gl5_t mk5(gl_t a, gl_t b, gl_t c, gl_t d, gl_t e) { gl5_t r = {{a, b, c, d, e}}; return r; }
static ec_t simple_swu(gl5_t u) {
  gl5_t two_thirds = gl5_from(6148914689804861441ULL);
  gl5_t a_sw = mk5(6148914689804861439ULL, 263, 0, 0, 0);
  gl5_t b_sw = mk5(15713893096167979237ULL, 6148914689804861265ULL, 0, 0, 0);
  gl5_t z_sw = mk5(GL_P - 4, GL_P - 1, 0, 0, 0);
  gl5_t neg_z_inv = mk5(4795794222525505369ULL, 3412737461722269738ULL, 8370187669276724726ULL, 7130825117388110979ULL, 12052351772713910496ULL);
  gl5_t neg_b_div_a = mk5(6585749426319121644ULL, 16990361517133133838ULL, 3264760655763595284ULL, 16784740989273302855ULL, 13434657726302040770ULL);
  gl5_t denom_part = gl5_mul(z_sw, gl5_sqr(u));
  gl5_t denom = gl5_add(gl5_sqr(denom_part), denom_part);
  gl5_t tv1 = gl5_inv(denom);
  gl5_t x1 = gl5_mul(gl5_is_zero(tv1) ? neg_z_inv : gl5_add(tv1, gl5_from(1)), neg_b_div_a);
  gl5_t x2 = gl5_mul(denom_part, x1);
  gl5_t gx1 = gl5_add(gl5_add(gl5_mul(x1, gl5_sqr(x1)), gl5_mul(a_sw, x1)), b_sw);
  gl5_t gx2 = gl5_add(gl5_add(gl5_mul(x2, gl5_sqr(x2)), gl5_mul(a_sw, x2)), b_sw);
  gl5_t x_sw, y_pos;
  if (gl5_sqrt(gx1, &y_pos)) x_sw = x1;
  else { x_sw = x2; if (!gl5_sqrt(gx2, &y_pos)) abort(); }
  gl5_t x_cand = gl5_sub(x_sw, two_thirds);
  gl5_t y_cand = gl5_sgn0(u) == gl5_sgn0(y_pos) ? y_pos : gl5_neg(y_pos);
  ec_t p;
  if (!grp_decode(gl5_mul(y_cand, gl5_inv(x_cand)), &p)) abort();
  return p;
}
// out: w[5] canonical encoding, wei[11] Weierstrass limbs (either may be NULL)
static void emit(ec_t p, gl_t* w, gl_t* wei) {
  if (w) { gl5_t e = grp_encode(p); memcpy(w, e.c, 40); }
  if (wei) grp_to_weierstrass(p, wei);
}
void orc_swu(const gl_t u[5], gl_t w[5], gl_t wei[11]) {
  gl5_t t; memcpy(&t, u, 40);
  emit(simple_swu(t), w, wei);
}
static ec_t map_to_curve(int variant, const gl_t* in, size_t n) {
  gl5_t h;
  orc_hash_n_to_m_no_pad(variant, in, n, h.c, 5);
  return simple_swu(h);
}
void orc_map_to_curve_batch(int variant, const gl_t* in, size_t in_len, size_t count, gl_t* w_out, gl_t* wei_out) {
#pragma omp parallel for schedule(dynamic, 16)
  for (size_t i = 0; i < count; i++)
    emit(map_to_curve(variant, in + i * in_len, in_len), w_out ? w_out + 5 * i : 0, wei_out ? wei_out + 11 * i : 0);
}
// returns 0 on an invalid encoding
int orc_decode_check(const gl_t w[5]) { gl5_t t; memcpy(&t, w, 40); ec_t p; return grp_decode(t, &p) && on_curve(p); }
// sum of encoded points (curve_add.rs:17-22)
int orc_curve_sum(const gl_t* w_in, size_t count, gl_t w[5], gl_t wei[11]) {
  ec_t acc = ec_N();
  for (size_t i = 0; i < count; i++) {
    gl5_t t; memcpy(&t, w_in + 5 * i, 40);
    ec_t p;
    if (!grp_decode(t, &p)) return 0;
    acc = grp_add(acc, p);
  }
  emit(acc, w, wei);
  return 1;
}
// scalar (little-endian u32 limbs) * decode(w)
int orc_scalar_mul(const gl_t w_in[5], const uint32_t* k_le, int n_limbs, gl_t w[5], gl_t wei[11]) {
  gl5_t t; memcpy(&t, w_in, 40);
  ec_t p;
  if (!grp_decode(t, &p)) return 0;
  emit(grp_mul(p, k_le, n_limbs), w, wei);
  return 1;
}
// poseidon.rs:120-133 hash_to_int_value: e0 + e1*2^64 as 4 little-endian u32 limbs
static void hash_to_int(const gl_t h[4], uint32_t k[4]) {
  k[0] = (uint32_t)h[0]; k[1] = (uint32_t)(h[0] >> 32);
  k[2] = (uint32_t)h[1]; k[3] = (uint32_t)(h[1] >> 32);
}
// mod.rs:220-225 field_hashed_scalar_mul(inputs, base)
int orc_field_hashed_scalar_mul(int variant, const gl_t* inputs, size_t n, const gl_t base_w[5], gl_t w[5], gl_t wei[11]) {
  gl_t h[4]; uint32_t k[4];
  orc_hash_n_to_m_no_pad(variant, inputs, n, h, 4);
  hash_to_int(h, k);
  return orc_scalar_mul(base_w, k, 4, w, wei);
}
// compute_table_row_digest, values_extraction/mod.rs:527-571.
//   col_ids[n_cols]; values[rows][n_cols][8] = U256 as 8 big-endian u32 words (u256.rs:870-877);
//   unique[rows][n_unique][8] = the row-unique columns' values, same packing (mod.rs:499-510:
//   left_pad32 + pack(Big) == the same 8 words).
void orc_row_digest_batch(int variant, const gl_t* col_ids, size_t n_cols, const uint32_t* values,
                          const uint32_t* unique, size_t n_unique, size_t rows, gl_t w[5], gl_t wei[11]) {
  ec_t total = ec_N();
#pragma omp parallel
  {
    ec_t local = ec_N();
#pragma omp for schedule(dynamic, 8) nowait
    for (size_t r = 0; r < rows; r++) {
      ec_t rd = ec_N();
      for (size_t c = 0; c < n_cols; c++) {
        gl_t in[9];
        in[0] = col_ids[c];
        for (int j = 0; j < 8; j++) in[1 + j] = values[(r * n_cols + c) * 8 + j];
        rd = grp_add(rd, map_to_curve(variant, in, 9));
      }
      gl_t* ub = malloc((8 * n_unique + 1) * sizeof(gl_t));
      for (size_t j = 0; j < 8 * n_unique; j++) ub[j] = unique[r * 8 * n_unique + j];
      gl_t h[5], h2[4];
      orc_hash_n_to_m_no_pad(variant, ub, 8 * n_unique, h, 4);  // row_unique_data
      free(ub);
      h[4] = (gl_t)n_cols;                                       // compute_row_id
      orc_hash_n_to_m_no_pad(variant, h, 5, h2, 4);
      uint32_t k[4];
      hash_to_int(h2, k);
      local = grp_add(local, grp_mul(rd, k, 4));
    }
#pragma omp critical
    total = grp_add(total, local);
  }
  emit(total, w, wei);
}

// ---- off-chain table commitment, mp2-v1/src/api.rs:553-603 -----------------------------------------
// add_primary_index_to_digest, verifiable-db/src/block_tree/mod.rs:37-53:
//   HashToInt(H(primary_index_id || index_value.to_fields())) * digest
int orc_add_primary_index_to_digest(int variant, gl_t primary_id, const uint32_t index_value_be[8], const gl_t digest_w[5], gl_t w[5], gl_t wei[11]) {
  gl_t in[9];
  in[0] = primary_id;
  for (int j = 0; j < 8; j++) in[1 + j] = index_value_be[j];
  return orc_field_hashed_scalar_mul(variant, in, 9, digest_w, w, wei);
}
// flatten_poseidon_hash_value, mp2-common/src/poseidon.rs:92-103: per limb [high 32 bits, low 32 bits]
static void flatten_hash(const gl_t h[4], gl_t out[8]) {
  for (int i = 0; i < 4; i++) { out[2 * i] = h[i] >> 32; out[2 * i + 1] = h[i] & 0xFFFFFFFFULL; }
}
static const uint32_t* g_sort_keys;  // qsort context (single-threaded test helper)
static int cmp_u256_be(const void* a, const void* b) {
  const uint32_t* x = g_sort_keys + 8 * *(const size_t*)a;
  const uint32_t* y = g_sort_keys + 8 * *(const size_t*)b;
  for (int j = 0; j < 8; j++) if (x[j] != y[j]) return x[j] < y[j] ? -1 : 1;
  return *(const size_t*)a < *(const size_t*)b ? -1 : (*(const size_t*)a > *(const size_t*)b);
}
// update_off_chain_data_commitment (api.rs:556-596). Rows: primary[rows][8] = the primary-index value of each row, values
// [rows][n_cols][8] = its other columns (ids col_ids), unique[rows][n_unique][8] = the values of its row-unique columns, every
// U256 as 8 big-endian u32 words. old_commitment: 32 bytes or NULL (HashOutput::default() = zeros). Groups the rows by
// increasing primary value (the BTreeMap of :562-570) and, group by group,
//   commitment <- flatten(H(commitment[8] || add_primary_index_to_digest(primary_id, primary, compute_table_row_digest(group)).to_fields()))
// starting from the old commitment packed as 8 little-endian u32 (:572-580); out = the 8 limbs as little-endian u32 bytes (:599-603).
void orc_update_off_chain_data_commitment(int variant, gl_t primary_id, const uint32_t* primary, const gl_t* col_ids, size_t n_cols,
                                          const uint32_t* values, const uint32_t* unique, size_t n_unique, size_t rows,
                                          const uint8_t* old_commitment, uint8_t out[32]) {
  gl_t com[8];
  for (int i = 0; i < 8; i++) {
    com[i] = 0;
    if (old_commitment)
      for (int b = 0; b < 4; b++) com[i] |= (gl_t)old_commitment[4 * i + b] << (8 * b);
  }
  size_t* order = malloc((rows ? rows : 1) * sizeof(size_t));
  for (size_t r = 0; r < rows; r++) order[r] = r;
  g_sort_keys = primary;
  qsort(order, rows, sizeof(size_t), cmp_u256_be);
  uint32_t* gv = malloc((rows * n_cols * 8 + 1) * sizeof(uint32_t));
  uint32_t* gu = malloc((rows * n_unique * 8 + 1) * sizeof(uint32_t));
  for (size_t lo = 0; lo < rows;) {
    size_t hi = lo + 1;
    while (hi < rows && memcmp(primary + 8 * order[hi], primary + 8 * order[lo], 32) == 0) hi++;
    for (size_t i = lo; i < hi; i++) {
      memcpy(gv + (i - lo) * n_cols * 8, values + order[i] * n_cols * 8, n_cols * 32);
      memcpy(gu + (i - lo) * n_unique * 8, unique + order[i] * n_unique * 8, n_unique * 32);
    }
    gl_t dw[5], pw[5], fields[11], payload[19], h[4];
    orc_row_digest_batch(variant, col_ids, n_cols, gv, gu, n_unique, hi - lo, dw, NULL);
    if (!orc_add_primary_index_to_digest(variant, primary_id, primary + 8 * order[lo], dw, pw, fields)) abort();
    memcpy(payload, com, sizeof(com));
    memcpy(payload + 8, fields, sizeof(fields));
    orc_hash_n_to_m_no_pad(variant, payload, 19, h, 4);
    flatten_hash(h, com);
    lo = hi;
  }
  free(order); free(gv); free(gu);
  for (int i = 0; i < 8; i++)
    for (int b = 0; b < 4; b++) out[4 * i + b] = (uint8_t)(com[i] >> (8 * b));
}
