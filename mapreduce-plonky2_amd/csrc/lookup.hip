// The lookup argument of prove() on gfx950, batched over proofs.
//
// Replaces [dep] plonky2 plonk/prover.rs compute_lookup_polys / compute_all_lookup_polys (the RE, Sum and LDC
// polynomials of the lookup tables, after the Tip5 paper's formulation) and plonk/vanishing_poly.rs
// check_lookup_constraints + get_lut_poly (their constraints inside compute_quotient_polys), for the circuits of
// the reference that use lookup tables (mp2-v1/src/values_extraction/gadgets/column_gadget.rs:53-68,136-153).
// The gate rows of a table are "upside down" (CircuitBuilder::add_all_lookups): every recurrence steps from row + 1
// to row, so no constraint needs the next row's wires.
//
//   lut_eval_kernel        get_lut_poly of every table for every (proof, challenge round)
//   lookup_rows_kernel     per lookup row: the row's own contribution to RE and to the 6 partial Sum/LDC polynomials
//                          (one batched inversion per row), written in place
//   lookup_scan_kernel     per (table, round, proof): the running sums down the rows (a few adds per row)
//   quotient_lookup_kernel the lookup terms at every LDE point, folded into the alpha-reduction between the
//                          partial-product terms (quotient_perm_kernel) and the gate terms (gates.hip)
#include "lookup.h"

namespace mp2g {

#define LU_MAX_SLOTS 40   // LookupGate::num_slots under standard_recursion_config (num_routed / 2)
#define LU_MAX_SLDC 8

// padded table polynomial at delta: sum_i (in_i + B out_i) delta^(padded - 1 - i)
__global__ void __launch_bounds__(256) lut_eval_kernel(LookupDev L, const u64* __restrict__ deltas, u64 d_bstride, u32 nc, u64* __restrict__ out) {
  const u32 r = blockIdx.x, c = blockIdx.y, b = blockIdx.z, t = threadIdx.x;
  const u64* d = deltas + b * d_bstride + 4 * c;
  const u64 dB = d[1], dD = d[3];
  const u32 len = L.table_len[r], rows = (len + L.num_lut_slots - 1) / L.num_lut_slots, padded = rows * L.num_lut_slots;
  const u16* tab = L.table[r];
  u64 acc = 0;
  for (u32 i = t; i < len; i += 256)
    acc = gl_add(acc, gl_mul(gl_add(tab[2 * i], gl_mul(dB, tab[2 * i + 1])), gl_pow(dD, padded - 1 - i)));
  __shared__ u64 red[256];
  red[t] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)t < s) red[t] = gl_add(red[t], red[t + s]);
    __syncthreads();
  }
  if (t == 0) out[((u64)b * nc + c) * MP2G_MAX_LUTS + r] = red[0];
}

__global__ void zero_rows_kernel(u64* p, u64 bstride, u64 words) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i < words) p[blockIdx.y * bstride + i] = 0;
}

// One lane per lookup row of table r: grid (ceil(rows/64), n_luts, nc * B).
// polys: [B][...] with proof stride p_bstride; round c's lookup polynomials start at polys + c * nlp * n: RE, then
// the num_sldc partial polynomials. Writes the row's own contributions (the scan kernel accumulates them).
__global__ void __launch_bounds__(64) lookup_rows_kernel(LookupDev L, const u64* __restrict__ wires, u64 w_bstride, u32 log_n,
                                                          const u64* __restrict__ deltas, u64 d_bstride, u32 nc, u64* __restrict__ polys,
                                                          u64 p_bstride) {
  const u32 r = blockIdx.y, c = blockIdx.z % nc, b = blockIdx.z / nc;
  const u32 lo = L.last_lu_row[r], mid = L.last_lut_row[r], hi = L.first_lut_row[r];
  const u32 row = lo + blockIdx.x * 64 + threadIdx.x;
  if (row > hi) return;
  const u64 n = (u64)1 << log_n;
  const u64* d = deltas + b * d_bstride + 4 * c;
  const u64 dA = d[0], dB = d[1], dAl = d[2], dD = d[3];
  const u64* w = wires + b * w_bstride + row;
  const u32 ns = L.num_sldc, nlp = ns + 1;
  u64* out = polys + b * p_bstride + (u64)c * nlp * n + row;
  const bool table_row = row >= mid;
  const u32 slots = table_row ? L.num_lut_slots : L.num_lu_slots, per = table_row ? 3 : 2, deg = table_row ? L.lut_degree : L.lu_degree;
  // alpha - (inp + A out) of every slot, inverted together
  u64 den[LU_MAX_SLOTS], pre[LU_MAX_SLOTS];
  u64 acc = 1, re = 0;
  for (u32 s = 0; s < slots; s++) {
    const u64 inp = w[(u64)(per * s) << log_n], outp = w[(u64)(per * s + 1) << log_n];
    if (table_row) re = gl_add(gl_mul(re, dD), gl_add(inp, gl_mul(dB, outp)));
    u64 v = gl_sub(dAl, gl_add(inp, gl_mul(dA, outp)));
    den[s] = v;
    pre[s] = acc;
    acc = gl_mul(acc, v ? v : 1);
  }
  u64 inv = gl_inv(acc);
  for (int s = (int)slots - 1; s >= 0; s--) {
    const u64 v = den[s];
    den[s] = v ? gl_mul(inv, pre[s]) : 0;  // 1 / (alpha - combo); 0 -> 0 (negligible probability over alpha)
    inv = gl_mul(inv, v ? v : 1);
  }
  if (table_row) out[0] = re;  // the row's chunk of the RE Horner form: sum_s combo_B(s) delta^(slots - 1 - s)
  for (u32 p = 0; p < ns; p++) {
    u64 sum = 0;
    for (u32 s = p * deg; s < (p + 1) * deg && s < slots; s++)
      sum = gl_add(sum, table_row ? gl_mul(w[(u64)(3 * s + 2) << log_n], den[s]) : den[s]);
    out[(u64)(p + 1) << log_n] = table_row ? sum : gl_sub(0, sum);  // Sum adds mult / (alpha - combo), LDC subtracts 1 / (alpha - combo)
  }
}
// grid (n_luts, nc, B), one 64-lane wave per (table, challenge round, proof): the table's rows are walked downwards, carrying RE
// (Horner in delta^slots over the LookupTable rows: re <- re dpow + chunk) and the running Sum - LDC value through the partial
// polynomials of each row and on to the next row. Both are scans: the RE recurrence composes as affine maps (a, b): re -> a re + b,
// the running value is a plain prefix sum over the (row, partial polynomial) cells in walk order. Each lane takes a run of
// consecutive rows, folds it, the wave scans the lanes' folds, and each lane walks its run again with its carry-in (a 65536-entry
// table spans ~2500 rows: 40 per lane instead of 2500 dependent steps on one lane).
__global__ void __launch_bounds__(64) lookup_scan_kernel(LookupDev L, u32 log_n, const u64* __restrict__ deltas, u64 d_bstride, u32 nc,
                                                        u64* __restrict__ polys, u64 p_bstride, const u64* __restrict__ lut_eval,
                                                        u32* __restrict__ flags) {
  const u32 lane = threadIdx.x, r = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const u64 n = (u64)1 << log_n;
  const u32 ns = L.num_sldc, nlp = ns + 1;
  u64* out = polys + b * p_bstride + (u64)c * nlp * n;
  const u64 dpow = gl_pow(deltas[b * d_bstride + 4 * c + 3], L.num_lut_slots);
  // walk position t = 0 .. total-1 visits row first_lut_row - t
  const u32 top = L.first_lut_row[r], bottom = L.last_lu_row[r], lut_end = L.last_lut_row[r];
  const u32 total = top - bottom + 1, per = (total + 63) / 64;
  const u32 t0 = lane * per < total ? lane * per : total, t1 = t0 + per < total ? t0 + per : total;
  // fold of this lane's run: RE map (A, Bc) over its LookupTable rows, sum of its Sum / LDC cells
  u64 A = 1, Bc = 0, tot = 0;
  for (u32 t = t0; t < t1; t++) {
    const u32 row = top - t;
    if (row >= lut_end) { Bc = gl_add(gl_mul(Bc, dpow), out[row]); A = gl_mul(A, dpow); }
    for (u32 p = 0; p < ns; p++) tot = gl_add(tot, out[((u64)(p + 1) << log_n) + row]);
  }
  // exclusive scans over the lanes (lane 0 walks first): carry-in RE = composition of the earlier lanes' maps applied to 0
  u64 cA = A, cB = Bc, cT = tot;  // inclusive
  for (u32 d = 1; d < 64; d <<= 1) {
    const u64 pA = __shfl_up(cA, d), pB = __shfl_up(cB, d), pT = __shfl_up(cT, d);
    if (lane >= d) {
      // earlier map (pA, pB) first, then this one (cA, cB): re -> cA (pA re + pB) + cB
      cB = gl_add(gl_mul(cA, pB), cB);
      cA = gl_mul(cA, pA);
      cT = gl_add(cT, pT);
    }
  }
  u64 re = __shfl_up(cB, 1), run = __shfl_up(cT, 1);  // applied to re = 0: the map's constant term
  if (lane == 0) { re = 0; run = 0; }
  for (u32 t = t0; t < t1; t++) {
    const u32 row = top - t;
    if (row >= lut_end) {
      re = gl_add(gl_mul(re, dpow), out[row]);
      out[row] = re;
    }
    for (u32 p = 0; p < ns; p++) {
      u64* cell = out + ((u64)(p + 1) << log_n) + row;
      run = gl_add(run, *cell);
      *cell = run;
    }
    // witness check (what the LastLdc and the table-end constraints enforce): every looked-up pair is in the table
    // with the stated multiplicities iff Sum - LDC returns to zero; the table rows hold the registered table iff RE
    // ends at the table's polynomial
    if (flags && row == lut_end && re != lut_eval[((u64)b * nc + c) * MP2G_MAX_LUTS + r]) atomicOr(&flags[b], 4u);
  }
  // the running value after the last cell of the walk: held by the lane that walked the end
  const u64 final_run = __shfl(cT, 63);
  if (flags && lane == 0 && final_run != 0) atomicOr(&flags[b], 4u);
}

// The lookup terms of the vanishing polynomial at every LDE point, alpha-reduced and folded in front of the gate
// sum: q[b][a][i] <- sum_k alpha_a^k T_k + alpha_a^(#T) q[b][a][i]. One lane per LDE column p in memory order (as
// quotient_perm_kernel). C = bit-reversed LDE of the constants (shared), W = wires, Z = the zs oracle.
__global__ void __launch_bounds__(256) quotient_lookup_kernel(LookupDev L, const u64* __restrict__ C, u32 sel_off, const u64* __restrict__ W,
                                                              u64 w_bstride, const u64* __restrict__ Z, u64 z_bstride, u32 lu_off, u32 log_n,
                                                              const u64* __restrict__ deltas, u64 d_bstride, const u64* __restrict__ lut_eval,
                                                              const u64* __restrict__ alphas, u64 al_bstride, u32 nc, u64* __restrict__ q) {
  const u32 lg = log_n + 3;
  const u64 N = (u64)1 << lg;
  const u32 p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (p >= N) return;
  const u32 i = bitrev32(p, lg);
  const u64 pn = bitrev32((i + 8) & (u32)(N - 1), lg);
  const u64* w = W + b * w_bstride + p;
  const u64* sel = C + ((u64)sel_off << lg) + p;
  const u32 ns = L.num_sldc, nlp = ns + 1;
  u64 acc[2] = {0, 0}, apow[2] = {1, 1}, al[2] = {0, 0};
  for (u32 a = 0; a < nc; a++) al[a] = alphas[b * al_bstride + a];
  auto push = [&](u64 term) {
    for (u32 a = 0; a < nc; a++) {
      acc[a] = gl_add(acc[a], gl_mul(term, apow[a]));
      apow[a] = gl_mul(apow[a], al[a]);
    }
  };
  const u64 s_sre = sel[0], s_ldc = sel[N], s_init = sel[2 * N], s_last = sel[3 * N];
  for (u32 c = 0; c < nc; c++) {
    const u64* d = deltas + b * d_bstride + 4 * c;
    const u64 dA = d[0], dB = d[1], dAl = d[2], dD = d[3];
    const u64* z = Z + b * z_bstride + ((u64)(lu_off + c * nlp) << lg);
    const u64 z_re = z[p], z_re_next = z[pn];
    push(gl_mul(s_last, z[((u64)ns << lg) + p]));  // last LDC
    push(gl_mul(s_init, z[((u64)1 << lg) + p]));   // initial Sum
    push(gl_mul(s_init, z_re));                    // initial RE
    for (u32 r = 0; r < L.n_luts; r++)             // RE at the end of table r = the table's polynomial
      push(gl_mul(sel[(u64)(4 + r) << lg], gl_sub(z_re, lut_eval[((u64)b * nc + c) * MP2G_MAX_LUTS + r])));
    u64 cur = z_re_next;
    for (u32 s = 0; s < L.num_lut_slots; s++)
      cur = gl_add(gl_mul(cur, dD), gl_add(w[(u64)(3 * s) << lg], gl_mul(dB, w[(u64)(3 * s + 1) << lg])));
    push(gl_mul(s_sre, gl_sub(z_re, cur)));
    for (u32 poly = 0; poly < ns; poly++) {
      const u64 prev = poly == 0 ? z[((u64)ns << lg) + pn] : z[((u64)poly << lg) + p];
      const u64 diff = gl_sub(z[((u64)(poly + 1) << lg) + p], prev);
      // prod (alpha - combo_j) and sum_i mult_i prod_{j != i} (alpha - combo_j) over the slots of this partial
      // polynomial, by prefix products: P_k = prod_{j < k} f_j, S_k = S_{k-1} f_k + m_k P_k
      u64 prod = 1, sum = 0;
      for (u32 s = poly * L.lut_degree; s < (poly + 1) * L.lut_degree && s < L.num_lut_slots; s++) {
        const u64 f = gl_sub(dAl, gl_add(w[(u64)(3 * s) << lg], gl_mul(dA, w[(u64)(3 * s + 1) << lg])));
        sum = gl_add(gl_mul(sum, f), gl_mul(w[(u64)(3 * s + 2) << lg], prod));
        prod = gl_mul(prod, f);
      }
      push(gl_mul(s_sre, gl_sub(gl_mul(prod, diff), sum)));
      prod = 1; sum = 0;
      for (u32 s = poly * L.lu_degree; s < (poly + 1) * L.lu_degree && s < L.num_lu_slots; s++) {
        const u64 f = gl_sub(dAl, gl_add(w[(u64)(2 * s) << lg], gl_mul(dA, w[(u64)(2 * s + 1) << lg])));
        sum = gl_add(gl_mul(sum, f), prod);
        prod = gl_mul(prod, f);
      }
      push(gl_mul(s_ldc, gl_add(gl_mul(prod, diff), sum)));
    }
  }
  for (u32 a = 0; a < nc; a++) {
    u64* dst = q + (((u64)b * nc + a) << lg) + i;
    *dst = gl_add(acc[a], gl_mul(apow[a], *dst));
  }
}

hipError_t lookup_polys(hipStream_t s, u32 B, const LookupDev& L, const u64* wires, u64 w_bstride, u32 log_n, const u64* deltas,
                        u64 d_bstride, u32 nc, u64* polys, u64 p_bstride, const u64* lut_eval, u32* flags) {
  if (!L.n_luts) return hipSuccess;
  const u64 n = (u64)1 << log_n, words = (u64)nc * (L.num_sldc + 1) * n;
  hipLaunchKernelGGL(zero_rows_kernel, dim3((u32)((words + 255) / 256), B), dim3(256), 0, s, polys, p_bstride, words);
  u32 max_rows = 0;
  for (u32 r = 0; r < L.n_luts; r++) {
    const u32 rows = L.first_lut_row[r] - L.last_lu_row[r] + 1;
    if (rows > max_rows) max_rows = rows;
  }
  hipLaunchKernelGGL(lookup_rows_kernel, dim3((max_rows + 63) / 64, L.n_luts, nc * B), dim3(64), 0, s, L, wires, w_bstride, log_n, deltas,
                     d_bstride, nc, polys, p_bstride);
  hipLaunchKernelGGL(lookup_scan_kernel, dim3(L.n_luts, nc, B), dim3(64), 0, s, L, log_n, deltas, d_bstride, nc, polys, p_bstride, lut_eval,
                     flags);
  return hipGetLastError();
}
hipError_t lookup_table_polys(hipStream_t s, u32 B, const LookupDev& L, const u64* deltas, u64 d_bstride, u32 nc, u64* lut_eval) {
  if (!L.n_luts) return hipSuccess;
  hipLaunchKernelGGL(lut_eval_kernel, dim3(L.n_luts, nc, B), dim3(256), 0, s, L, deltas, d_bstride, nc, lut_eval);
  return hipGetLastError();
}
hipError_t quotient_lookup_values(hipStream_t s, u32 B, const LookupDev& L, const u64* C, u32 sel_off, const u64* W, u64 w_bstride,
                                  const u64* Z, u64 z_bstride, u32 lu_off, u32 log_n, const u64* deltas, u64 d_bstride,
                                  const u64* lut_eval, const u64* alphas, u64 al_bstride, u32 nc, u64* q) {
  if (!L.n_luts) return hipSuccess;
  if (nc < 1 || nc > 2) return hipErrorInvalidValue;
  const u64 N = (u64)8 << log_n;
  hipLaunchKernelGGL(quotient_lookup_kernel, dim3((u32)((N + 255) / 256), B), dim3(256), 0, s, L, C, sel_off, W, w_bstride, Z, z_bstride,
                     lu_off, log_n, deltas, d_bstride, lut_eval, alphas, al_bstride, nc, q);
  return hipGetLastError();
}
}  // namespace mp2g
