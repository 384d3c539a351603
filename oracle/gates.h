// TEST INFRASTRUCTURE -- CPU oracle (see gl.h header).
// Gate descriptors for the constraint evaluators of gates.c. Same meaning, field for field, as mp2g_gate in
// include/mp2g.h (separate definition: the product never includes oracle headers).
#ifndef MP2_ORACLE_GATES_H
#define MP2_ORACLE_GATES_H
#include "gl.h"
enum {
  ORC_GATE_NOOP = 0,
  ORC_GATE_CONSTANT = 1,        // p0 = num_consts
  ORC_GATE_PUBLIC_INPUT = 2,
  ORC_GATE_ARITHMETIC = 3,      // p0 = num_ops
  ORC_GATE_BASE_SUM = 4,        // p0 = num_limbs, p1 = base
  ORC_GATE_ARITHMETIC_EXT = 5,  // p0 = num_ops
  ORC_GATE_MUL_EXT = 6,         // p0 = num_ops
  ORC_GATE_POSEIDON2 = 7,
  ORC_GATE_EXPONENTIATION = 8,  // p0 = num_power_bits
  ORC_GATE_REDUCING = 9,        // p0 = num_coeffs
  ORC_GATE_REDUCING_EXT = 10,   // p0 = num_coeffs
  ORC_GATE_RANDOM_ACCESS = 11,  // p0 = bits, p1 = num_copies, p2 = num_extra_constants
  ORC_GATE_POSEIDON = 12,
  ORC_GATE_POSEIDON_MDS = 13,
  ORC_GATE_COSET_INTERPOLATION = 14,  // p0 = subgroup_bits (<= 5), p1 = degree
  ORC_GATE_U32_ARITHMETIC = 15,       // p0 = num_ops
  ORC_GATE_U32_RANGE_CHECK = 16,      // p0 = num_input_limbs
  ORC_GATE_U32_SUBTRACTION = 17,      // p0 = num_ops
  ORC_GATE_U32_ADD_MANY = 18,         // p0 = num_addends, p1 = num_ops
  ORC_GATE_COMPARISON = 19,           // p0 = num_bits, p1 = num_chunks
  ORC_GATE_LOOKUP = 20,               // p0 = num_slots (no constraints of its own: the lookup argument carries them)
  ORC_GATE_LOOKUP_TABLE = 21,         // p0 = num_slots
  ORC_GATE_U32_INTERLEAVE = 22,       // p0 = num_ops
  ORC_GATE_UNINTERLEAVE_TO_B32 = 23,  // p0 = num_ops
  ORC_GATE_UNINTERLEAVE_TO_U32 = 24,  // p0 = num_ops
};
// One lookup table and the rows plonky2's CircuitBuilder::add_all_lookups gave it (LookupWire): LookupGate rows
// [last_lu_row, last_lut_row), LookupTableGate rows [last_lut_row, first_lut_row] (the table runs DOWN from
// first_lut_row), then one Noop row.
typedef struct {
  uint32_t last_lu_row, last_lut_row, first_lut_row, table_len;
  const uint16_t* table;  // [table_len][2] = (input, output)
} orc_lookup;
// the lookup argument of a circuit: its tables and the slot geometry of standard_recursion_config
typedef struct {
  const orc_lookup* luts;
  unsigned n_luts;
  unsigned num_lu_slots, num_lut_slots;  // LookupGate::num_slots = routed/2, LookupTableGate::num_slots = routed/3
  unsigned num_sldc, lu_degree, lut_degree;
} orc_lookup_ctx;
// fills the derived fields: num_sldc = ceil(num_lu_slots / (degree - 1)), lu_degree = degree - 1,
// lut_degree = ceil(num_lut_slots / num_sldc); degree = quotient degree factor (8)
void orc_lookup_shape(orc_lookup_ctx* L, unsigned num_routed, unsigned degree);
#define ORC_LOOKUP_SELECTORS 4  // TransSre, TransLdc, InitSre, LastLdc; then one "ends" selector per table
#define ORC_MAX_LUTS 16
#define ORC_MAX_GATE_CONSTRAINTS 160
typedef struct {
  uint32_t kind, p0, p1, p2;
  uint32_t selector_index;            // which selector polynomial filters this gate
  uint32_t group_start, group_end;    // gate indices sharing that selector (gates/selectors.rs groups)
} orc_gate;
// two_adic_subgroup(bits) and its barycentric weights w_i = 1 / prod_{j != i} (x_i - x_j)
void orc_barycentric_weights(unsigned bits, gl_t* domain, gl_t* weights);
unsigned orc_gate_num_constraints(const orc_gate* g);
unsigned orc_gate_degree(const orc_gate* g);
// consts = ALL local constants: num_selectors selectors, num_lookup_selectors lookup selectors, then the gate constants
unsigned orc_gates_eval_base(const orc_gate* gates, unsigned n_gates, unsigned num_selectors, unsigned num_lookup_selectors,
                             const gl_t* consts, const gl_t* wires, const gl_t* pih, gl_t* acc);
unsigned orc_gates_eval_ext(const orc_gate* gates, unsigned n_gates, unsigned num_selectors, unsigned num_lookup_selectors,
                            const gl2_t* consts, const gl2_t* wires, const gl2_t* pih, gl2_t* acc);
// plonk/vanishing_poly.rs check_lookup_constraints for one challenge: lookup_sel = the 4 + n_luts lookup selector
// values, zs / zs_next = the num_sldc + 1 lookup polynomials (RE first) at the point and at g * point, deltas =
// [A, B, alpha, delta]. Writes 4 + n_luts + 2 * num_sldc terms, returns that count.
unsigned orc_lookup_terms_base(const orc_lookup_ctx* L, const gl_t* lookup_sel, const gl_t* wires, const gl_t* zs, const gl_t* zs_next,
                               const gl_t deltas[4], gl_t* out);
unsigned orc_lookup_terms_ext(const orc_lookup_ctx* L, const gl2_t* lookup_sel, const gl2_t* wires, const gl2_t* zs, const gl2_t* zs_next,
                              const gl_t deltas[4], gl2_t* out);
// get_lut_poly: sum_i (in_i + B out_i) delta^(padded_len - 1 - i), padded_len = num_lut_slots * ceil(len / num_lut_slots)
gl_t orc_lut_poly(const orc_lookup* lut, unsigned num_lut_slots, const gl_t deltas[4]);
#endif
