// Launchers of the gate-constraint kernels (gates.hip).
#pragma once
#include "gl.cuh"
#include "mp2g.h"
namespace mp2g {
struct GateTable {
  u32 n_gates, num_selectors;
  u32 num_lookup_selectors;  // 0, or 4 + n_luts constants between the selectors and the gate constants
  mp2g_gate g[MP2G_MAX_GATES];
};
// validates kinds / parameters against the wire and constant counts; returns nullptr or a message
const char* gate_table_check(const GateTable& t, u32 num_constants, u32 wires_w);
u32 gate_num_constraints(const mp2g_gate& g);
u32 gate_degree(const mp2g_gate& g);
// q[b][a][i] (natural order i) = sum_g filter_g sum_j alpha_a^j c_{g,j} at the LDE point of memory column
// p = bitrev(i): C / W are the bit-reversed LDE value matrices [.][N] of the constants (shared) and the
// wires (per proof), N = 8n. quotient_perm_values(..., gates = true) folds q into the vanishing sum.
hipError_t gate_constraints_lde(hipStream_t s, u32 B, const GateTable& t, const u64* C, const u64* W, u64 w_bstride, u32 lg,
                                const u64* alphas, u64 al_bstride, u32 nc, const u64* pi_hash, u64* q);
// out[j][p] = C_j at point p (device pointers; consts [.][npts], wires [.][npts])
hipError_t gate_constraints_points(hipStream_t s, const GateTable& t, const u64* consts, const u64* wires, u64 npts, u32 max_j,
                                   const u64* pi_hash, u64* out);
// flags[b] |= 2 where a gate constraint of proof b is non-zero on the subgroup: consts [.][npts] (shared),
// wires [B][.][npts] with batch stride w_bstride, pi_hash [B][4]
hipError_t gate_check(hipStream_t s, u32 B, const GateTable& t, const u64* consts, const u64* wires, u64 w_bstride, u64 npts,
                      const u64* pi_hash, u32* flags);
}  // namespace mp2g
