// TEST INFRASTRUCTURE -- CPU oracle (see gl.h header).
// Gate constraint evaluation: the third part of plonky2's compute_quotient_polys (after the Z(1) = 1 and
// partial-product terms), [dep] plonky2/src/plonk/vanishing_poly.rs + plonky2/src/gates/*.rs. PARITY
// UNPINNED: restated from the published source; the checks available here are self-consistency ones (a
// satisfied witness vanishes on H, the verifier identity holds at zeta, an unsatisfied witness fails).
#include "gates.h"
#include "constants.h"
#include <stdlib.h>

void orc_barycentric_weights(unsigned bits, gl_t* domain, gl_t* weights) {
  unsigned n = 1u << bits;
  gl_t w = gl_root_of_unity(bits), x = 1;
  for (unsigned i = 0; i < n; i++) { domain[i] = x; x = gl_mul(x, w); }
  for (unsigned i = 0; i < n; i++) {
    gl_t p = 1;
    for (unsigned j = 0; j < n; j++)
      if (j != i) p = gl_mul(p, gl_sub(domain[i], domain[j]));
    weights[i] = gl_inv(p);
  }
}

#define F gl_t
#define F_ADD gl_add
#define F_SUB gl_sub
#define F_MUL gl_mul
#define F_CONST(x) ((gl_t)(x))
#define FN(n) b_##n
#include "gates_body.inc"
#undef F
#undef F_ADD
#undef F_SUB
#undef F_MUL
#undef F_CONST
#undef FN

#define F gl2_t
#define F_ADD gl2_add
#define F_SUB gl2_sub
#define F_MUL gl2_mul
#define F_CONST(x) gl2_from((gl_t)(x))
#define FN(n) e_##n
#include "gates_body.inc"

unsigned orc_gates_eval_base(const orc_gate* gates, unsigned n_gates, unsigned num_selectors, unsigned num_lookup_selectors,
                             const gl_t* consts, const gl_t* wires, const gl_t* pih, gl_t* acc) {
  return b_eval_gate_constraints(gates, n_gates, num_selectors, num_lookup_selectors, consts, wires, pih, acc);
}
unsigned orc_gates_eval_ext(const orc_gate* gates, unsigned n_gates, unsigned num_selectors, unsigned num_lookup_selectors,
                            const gl2_t* consts, const gl2_t* wires, const gl2_t* pih, gl2_t* acc) {
  return e_eval_gate_constraints(gates, n_gates, num_selectors, num_lookup_selectors, consts, wires, pih, acc);
}
void orc_lookup_shape(orc_lookup_ctx* L, unsigned num_routed, unsigned degree) {
  L->num_lu_slots = num_routed / 2;
  L->num_lut_slots = num_routed / 3;
  L->lu_degree = degree - 1;
  L->num_sldc = (L->num_lu_slots + L->lu_degree - 1) / L->lu_degree;
  L->lut_degree = (L->num_lut_slots + L->num_sldc - 1) / L->num_sldc;
}
gl_t orc_lut_poly(const orc_lookup* lut, unsigned num_lut_slots, const gl_t deltas[4]) {
  size_t rows = (lut->table_len + num_lut_slots - 1) / num_lut_slots, padded = rows * num_lut_slots;
  gl_t acc = 0;  // Horner over the zero-padded, reversed coefficient list = entries in table order
  for (size_t i = 0; i < padded; i++) {
    gl_t c = i < lut->table_len ? gl_add(lut->table[2 * i], gl_mul(deltas[1], lut->table[2 * i + 1])) : 0;
    acc = gl_add(gl_mul(acc, deltas[3]), c);
  }
  return acc;
}
unsigned orc_lookup_terms_base(const orc_lookup_ctx* L, const gl_t* lookup_sel, const gl_t* wires, const gl_t* zs, const gl_t* zs_next,
                               const gl_t deltas[4], gl_t* out) {
  gl_t ev[ORC_MAX_LUTS];
  for (unsigned r = 0; r < L->n_luts; r++) ev[r] = orc_lut_poly(&L->luts[r], L->num_lut_slots, deltas);
  return b_eval_lookup_constraints(L, lookup_sel, wires, zs, zs_next, deltas, ev, out);
}
unsigned orc_lookup_terms_ext(const orc_lookup_ctx* L, const gl2_t* lookup_sel, const gl2_t* wires, const gl2_t* zs, const gl2_t* zs_next,
                              const gl_t deltas[4], gl2_t* out) {
  gl_t ev[ORC_MAX_LUTS];
  for (unsigned r = 0; r < L->n_luts; r++) ev[r] = orc_lut_poly(&L->luts[r], L->num_lut_slots, deltas);
  return e_eval_lookup_constraints(L, lookup_sel, wires, zs, zs_next, deltas, ev, out);
}
unsigned orc_gate_num_constraints(const orc_gate* g) {
  switch (g->kind) {
    case ORC_GATE_CONSTANT: return g->p0;
    case ORC_GATE_PUBLIC_INPUT: return 4;
    case ORC_GATE_ARITHMETIC: return g->p0;
    case ORC_GATE_BASE_SUM: return 1 + g->p0;
    case ORC_GATE_ARITHMETIC_EXT: case ORC_GATE_MUL_EXT: return 2 * g->p0;
    case ORC_GATE_POSEIDON2: case ORC_GATE_POSEIDON: return 1 + 4 + 36 + 22 + 48 + 12;
    case ORC_GATE_POSEIDON_MDS: return 24;
    case ORC_GATE_COSET_INTERPOLATION: return 4 + 4 * (((1u << g->p0) - 2) / (g->p1 - 1));
    case ORC_GATE_U32_ARITHMETIC: return 36 * g->p0;
    case ORC_GATE_U32_RANGE_CHECK: return 17 * g->p0;
    case ORC_GATE_U32_SUBTRACTION: return 19 * g->p0;
    case ORC_GATE_U32_ADD_MANY: return 21 * g->p1;
    case ORC_GATE_COMPARISON: return 6 + 5 * g->p1 + (g->p0 + g->p1 - 1) / g->p1;
    case ORC_GATE_EXPONENTIATION: return g->p0 + 1;
    case ORC_GATE_REDUCING: case ORC_GATE_REDUCING_EXT: return 2 * g->p0;
    case ORC_GATE_RANDOM_ACCESS: return (g->p0 + 2) * g->p1 + g->p2;
    case ORC_GATE_U32_INTERLEAVE: return 34 * g->p0;
    case ORC_GATE_UNINTERLEAVE_TO_B32: case ORC_GATE_UNINTERLEAVE_TO_U32: return 67 * g->p0;
    default: return 0;  // Noop, Lookup, LookupTable
  }
}
// Gate::degree()
unsigned orc_gate_degree(const orc_gate* g) {
  switch (g->kind) {
    case ORC_GATE_CONSTANT: case ORC_GATE_PUBLIC_INPUT: return 1;
    case ORC_GATE_ARITHMETIC: case ORC_GATE_ARITHMETIC_EXT: case ORC_GATE_MUL_EXT: return 3;
    case ORC_GATE_BASE_SUM: return g->p1;
    case ORC_GATE_POSEIDON2: case ORC_GATE_POSEIDON: return 7;
    case ORC_GATE_POSEIDON_MDS: return 1;
    case ORC_GATE_COSET_INTERPOLATION: return g->p1;
    case ORC_GATE_U32_ARITHMETIC: case ORC_GATE_U32_RANGE_CHECK: case ORC_GATE_U32_SUBTRACTION: case ORC_GATE_U32_ADD_MANY: return 4;
    case ORC_GATE_COMPARISON: return 1u << ((g->p0 + g->p1 - 1) / g->p1);
    case ORC_GATE_EXPONENTIATION: return 4;
    case ORC_GATE_REDUCING: case ORC_GATE_REDUCING_EXT: return 2;
    case ORC_GATE_RANDOM_ACCESS: return g->p0 + 1;
    case ORC_GATE_U32_INTERLEAVE: case ORC_GATE_UNINTERLEAVE_TO_B32: case ORC_GATE_UNINTERLEAVE_TO_U32: return 2;
    default: return 0;
  }
}
// C_j at `npts` arbitrary points: consts [num_constants][npts], wires [wires_w][npts] (point-minor), out [maxc][npts].
// On the subgroup H with a satisfied witness every C_j is zero (the prover-side witness check).
unsigned orc_gates_eval_points(const orc_gate* gates, unsigned n_gates, unsigned num_selectors, unsigned num_constants,
                               const gl_t* consts, unsigned wires_w, const gl_t* wires, size_t npts, const gl_t* pih, gl_t* out) {
  unsigned maxc = 0;
  for (unsigned g = 0; g < n_gates; g++) {
    unsigned c = orc_gate_num_constraints(&gates[g]);
    if (c > maxc) maxc = c;
  }
  gl_t* lc = malloc((num_constants + wires_w + 1) * sizeof(gl_t));
  gl_t* lw = lc + num_constants;
  gl_t acc[ORC_MAX_GATE_CONSTRAINTS];
  for (size_t i = 0; i < npts; i++) {
    for (unsigned j = 0; j < num_constants; j++) lc[j] = consts[(size_t)j * npts + i];
    for (unsigned j = 0; j < wires_w; j++) lw[j] = wires[(size_t)j * npts + i];
    orc_gates_eval_base(gates, n_gates, num_selectors, 0, lc, lw, pih, acc);
    for (unsigned j = 0; j < maxc; j++) out[(size_t)j * npts + i] = acc[j];
  }
  free(lc);
  return maxc;
}
