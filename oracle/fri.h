// TEST INFRASTRUCTURE -- CPU oracle (see gl.h header).
#ifndef MP2_ORACLE_FRI_H
#define MP2_ORACLE_FRI_H
#include "gl.h"
#include <stddef.h>
// Parameters of one PCS/FRI instance. Field-for-field the same meaning as mp2g_fri_params in
// include/mp2g.h (kept as a separate definition: the product never includes oracle headers).
typedef struct {
  uint32_t variant;      // 0 Poseidon2, 1 Poseidon
  uint32_t log_n;        // degree_bits
  uint32_t rate_bits;    // 3
  uint32_t cap_height;   // 4
  uint32_t pow_bits;     // 16
  uint32_t num_queries;  // 28
  uint32_t n_layers;     // len(reduction_arity_bits)
  uint32_t arity_bits[8];
  uint32_t n_oracles;    // 4: constants_sigmas, wires, zs_partial_products, quotient
  uint32_t oracle_w[8];  // polynomials per oracle
  uint32_t zs_oracle;    // oracle whose first zs_count polys are also opened at g*zeta
  uint32_t zs_count;
  uint32_t num_lookup_polys;  // per challenge (0 = no lookup argument): the last zs_count * num_lookup_polys polynomials
                              // of oracle zs_oracle, opened at zeta and g*zeta, batched after the quotient polynomials
} orc_fri_params;

typedef struct {
  gl_t state[12];
  gl_t in[8];
  gl_t out[8];
  uint32_t n_in, n_out, variant;
} orc_challenger;

void orc_ch_init(orc_challenger* c, int variant);
void orc_ch_observe(orc_challenger* c, const gl_t* e, size_t n);
gl_t orc_ch_get(orc_challenger* c);
gl2_t orc_ch_get_ext(orc_challenger* c);
#endif
