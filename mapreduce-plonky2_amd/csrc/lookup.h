// Launchers of the lookup-argument kernels (lookup.hip).
#pragma once
#include "gl.cuh"
#include "mp2g.h"
namespace mp2g {
typedef uint16_t u16;
// the lookup tables of a circuit as the kernels take them (by value): rows of each table (plonky2's LookupWire), the
// table on the device, and the slot geometry (LookupGate / LookupTableGate num_slots, partial polynomial degrees)
struct LookupDev {
  u32 n_luts;
  u32 num_lu_slots, num_lut_slots, num_sldc, lu_degree, lut_degree;
  u32 last_lu_row[MP2G_MAX_LUTS], last_lut_row[MP2G_MAX_LUTS], first_lut_row[MP2G_MAX_LUTS], table_len[MP2G_MAX_LUTS];
  const u16* table[MP2G_MAX_LUTS];  // device, [table_len][2]
};
// lut_eval[b][c][MP2G_MAX_LUTS]: get_lut_poly of every table under proof b's round-c challenges (deltas [B][.] with
// 4 words per round: A, B, alpha, delta)
hipError_t lookup_table_polys(hipStream_t s, u32 B, const LookupDev& L, const u64* deltas, u64 d_bstride, u32 nc, u64* lut_eval);
// compute_all_lookup_polys: wires [B][.][n] subgroup values; polys + b * p_bstride = proof b's nc * (num_sldc + 1)
// lookup polynomials [.][n] (round-major: RE, then the partial Sum/LDC polynomials)
// flags (may be NULL; needs lut_eval): flags[b] |= 4 when proof b's lookups do not close (a looked-up pair missing from
// its table / wrong multiplicities / table rows that are not the registered table)
hipError_t lookup_polys(hipStream_t s, u32 B, const LookupDev& L, const u64* wires, u64 w_bstride, u32 log_n, const u64* deltas,
                        u64 d_bstride, u32 nc, u64* polys, u64 p_bstride, const u64* lut_eval, u32* flags);
// q[b][a][i] <- (lookup terms alpha-reduced) + alpha_a^(#terms) q[b][a][i] on the 8n-point coset; C / W / Z are the
// bit-reversed LDE matrices of the constants (shared), wires and zs oracle; sel_off = index of the first lookup
// selector among the constants, lu_off = index of the first lookup polynomial in the zs oracle
hipError_t quotient_lookup_values(hipStream_t s, u32 B, const LookupDev& L, const u64* C, u32 sel_off, const u64* W, u64 w_bstride,
                                  const u64* Z, u64 z_bstride, u32 lu_off, u32 log_n, const u64* deltas, u64 d_bstride,
                                  const u64* lut_eval, const u64* alphas, u64 al_bstride, u32 nc, u64* q);
}  // namespace mp2g
