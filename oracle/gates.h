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
};
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
unsigned orc_gates_eval_base(const orc_gate* gates, unsigned n_gates, unsigned num_selectors, const gl_t* consts,
                             const gl_t* wires, const gl_t* pih, gl_t* acc);
unsigned orc_gates_eval_ext(const orc_gate* gates, unsigned n_gates, unsigned num_selectors, const gl2_t* consts,
                            const gl2_t* wires, const gl2_t* pih, gl2_t* acc);
#endif
