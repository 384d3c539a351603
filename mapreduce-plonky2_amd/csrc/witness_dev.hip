// Witness generation on the device for circuits built by recursion.py: the executor of csrc/witness.hip as a kernel.
//
// Replaces [dep] plonky2 iop/generator.rs generate_partial_witness (first line of prove(), recursion-framework/src/
// circuit_builder.rs:308, universal_verifier_gadget/wrap_circuit.rs:143) for a BATCH of proofs of one circuit without the host:
// the recorded witness program is data independent and the same for every proof, so it is scheduled once (witness.hip:
// dependency levels; ~300 levels for a 10-20 k instruction verifier circuit, the long ones being the Merkle paths and leaf
// hashes of the 28 FRI query rounds) and replayed by one 512-lane block per proof: at every level lane i takes the level's
// i-th instruction (instructions of a level are ordered by opcode so that waves stay uniform), one block barrier between
// levels. Values live in a per-proof slot table in global memory (written once each: SSA), wires go into a row-major staging
// matrix [B][n][135] (a gate row's wires are one contiguous run) that witness.hip transposes into the prover's [B][135][n]. Arithmetic is gl.cuh / poseidon.cuh: the Poseidon2 gate's S-box inputs are the weak
// representatives of the sponge kernels, canonicalised where they become wires.
#include "gl.cuh"
#include "poseidon.cuh"
#include "poseidon_wave.cuh"
#include "witness.h"
#include "witness_ops.h"

namespace mp2g {
namespace {
// wires: this proof's staging matrix in ROW-major order [n][135] (a gate row's wires are consecutive words: a Poseidon2 row's 135 stores
// touch 17 sectors instead of 135; witness.hip transposes the batch into the prover's polynomial-major layout afterwards)
#define W(col, row) wires[(u64)(row) * NUM_WIRES + (u64)(col)]

GLD void exec_p2(const u64* t, u64* vals, u64* wires, u64 n) {
  // Poseidon2Gate: inputs 0..11, outputs 12..23, swap 24, deltas 25..28, S-box inputs 29.., 65.., 87..
  const u64 row = t[0];
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { s[i] = vals[t[1 + i]]; W(i, row) = s[i]; }
  const u64 swap = vals[t[13]];
  W(24, row) = swap;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const u64 delta = gl_mul(swap, gl_sub(s[i + 4], s[i]));
    W(25 + i, row) = delta;
    s[i] = gl_add(s[i], delta);
    s[i + 4] = gl_sub(s[i + 4], delta);
  }
  p2_external(s);
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
      s[i] = gl_canon(gl_addw(s[i], c_p2_ext[12 * r + i]));
      if (r) W(29 + 12 * (r - 1) + i, row) = s[i];
      s[i] = p2_sbox0(s[i]);
    }
    p2_external(s);
  }
#pragma unroll 1
  for (int r = 0; r < 22; r++) {
    const u64 x = gl_canon(gl_addw(s[0], c_p2_int[r]));
    W(65 + r, row) = x;
    s[0] = p2_sbox0(x);
    p2_internal(s);
  }
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
      s[i] = gl_canon(gl_addw(s[i], c_p2_ext[12 * (4 + r) + i]));
      W(87 + 12 * r + i, row) = s[i];
      s[i] = p2_sbox0(s[i]);
    }
    p2_external(s);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) { const u64 o = gl_canon(s[i]); W(12 + i, row) = o; vals[t[14 + i]] = o; }
}

// The same gate with its state spread over lanes 0..11 of an aligned 16-lane group (poseidon_wave.cuh): a level that holds only a
// few Poseidon2 rows -- the Fiat-Shamir chain, the upper part of a Merkle path -- is a chain of dependent permutations, and one
// permutation takes a lone lane ~45 us but a 12-lane group ~12 us. Every lane of the group calls; l = lane & 15.
GLD void exec_p2_coop(const u64* t, u64* vals, u64* wires, u64 n, int l) {
  const u64 row = t[0];
  const bool on = l < 12;
  const int li = on ? l : 0;
  u64 x = on ? vals[t[1 + li]] : 0;
  if (on) W(l, row) = x;
  const u64 swap = vals[t[13]];
  if (l == 0) W(24, row) = swap;
  {  // the swap of inputs[0..4) and [4..8): delta_i = swap (in[i + 4] - in[i]), lanes i and i + 4 both form it
    const u64 other = wp_shfl(x, (l ^ 4) & 15);
    const u64 lo_v = l < 4 ? x : other, hi_v = l < 4 ? other : x;
    const u64 delta = gl_mul(swap, gl_sub(hi_v, lo_v));
    if (l < 4) { W(25 + l, row) = delta; x = gl_add(x, delta); }
    else if (l < 8) x = gl_sub(x, delta);
  }
  u64 rc[8];
#pragma unroll
  for (int r = 0; r < 8; r++) rc[r] = c_p2_ext[12 * r + li];
  const u64 d = c_p2_diag[li];
  x = wp2_external(x, l);
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    const u64 k = r == 0 ? rc[0] : (r == 1 ? rc[1] : (r == 2 ? rc[2] : rc[3]));
    const u64 in = gl_canon(gl_addw(x, k));
    if (r && on) W(29 + 12 * (r - 1) + l, row) = in;
    x = wp2_external(p2_sbox0(in), l);
  }
#pragma unroll 1
  for (int r = 0; r < 22; r++) {
    const u64 in = gl_canon(gl_addw(x, c_p2_int[r]));
    if (l == 0) W(65 + r, row) = in;
    x = wp2_internal(l == 0 ? p2_sbox0(in) : x, l, d);
  }
#pragma unroll 1
  for (int r = 4; r < 8; r++) {
    const u64 k = r == 4 ? rc[4] : (r == 5 ? rc[5] : (r == 6 ? rc[6] : rc[7]));
    const u64 in = gl_canon(gl_addw(x, k));
    if (on) W(87 + 12 * (r - 4) + l, row) = in;
    x = wp2_external(p2_sbox0(in), l);
  }
  if (on) { const u64 o = gl_canon(x); W(12 + l, row) = o; vals[t[14 + l]] = o; }
}

GLD void exec_poseidon(const u64* t, u64* vals, u64* wires, u64 n) {
  // PoseidonGate ([dep] gates/poseidon.rs): the wire layout of the Poseidon2 gate, the original permutation (4 + 22 + 4 rounds)
  const u64 row = t[0];
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { s[i] = vals[t[1 + i]]; W(i, row) = s[i]; }
  const u64 swap = vals[t[13]];
  W(24, row) = swap;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const u64 delta = gl_mul(swap, gl_sub(s[i + 4], s[i]));
    W(25 + i, row) = delta;
    s[i] = gl_add(s[i], delta);
    s[i + 4] = gl_sub(s[i + 4], delta);
  }
#pragma unroll 1
  for (int r = 0; r < 30; r++) {
    if (r >= 4 && r < 26) {
#pragma unroll
      for (int i = 1; i < 12; i++) s[i] = gl_addw(s[i], c_p_rc[12 * r + i]);
      const u64 x = gl_canon(gl_addw(s[0], c_p_rc[12 * r]));
      W(65 + r - 4, row) = x;
      s[0] = p2_sbox0(x);
    } else {
      const int base = r < 4 ? 29 + 12 * (r - 1) : 87 + 12 * (r - 26);
#pragma unroll
      for (int i = 0; i < 12; i++) {
        const u64 x = gl_canon(gl_addw(s[i], c_p_rc[12 * r + i]));
        if (r) W(base + i, row) = x;
        s[i] = p2_sbox0(x);
      }
    }
    poseidon_mds(s);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) { const u64 o = gl_canon(s[i]); W(12 + i, row) = o; vals[t[14 + i]] = o; }
}

GLD void exec_one(const u64* t, u64* vals, u64* wires, u64 n, const u64* domtab) {
  const u64 op = *t++;
  switch (op) {
    case OP_WIRE: W(t[1], t[0]) = vals[t[2]]; break;
    case OP_ARITH: {
      const u64 row = t[0], i = t[1], c0 = t[2], c1 = t[3];
      const u64 m0 = vals[t[4]], m1 = vals[t[5]], ad = vals[t[6]];
      const u64 o = gl_add(gl_mul(gl_mul(m0, m1), c0), gl_mul(ad, c1));
      W(4 * i, row) = m0; W(4 * i + 1, row) = m1; W(4 * i + 2, row) = ad; W(4 * i + 3, row) = o;
      vals[t[7]] = o;
      break;
    }
    case OP_ARITH_EXT: {
      const u64 row = t[0], i = t[1], c0 = t[2], c1 = t[3];
      const gl2 m0 = gl2_make(vals[t[4]], vals[t[5]]), m1 = gl2_make(vals[t[6]], vals[t[7]]), ad = gl2_make(vals[t[8]], vals[t[9]]);
      const gl2 o = gl2_add(gl2_scale(gl2_mul(m0, m1), c0), gl2_scale(ad, c1));
      const u64 b = 8 * i;
      W(b, row) = m0.a; W(b + 1, row) = m0.b; W(b + 2, row) = m1.a; W(b + 3, row) = m1.b;
      W(b + 4, row) = ad.a; W(b + 5, row) = ad.b; W(b + 6, row) = o.a; W(b + 7, row) = o.b;
      vals[t[10]] = o.a; vals[t[11]] = o.b;
      break;
    }
    case OP_P2: exec_p2(t, vals, wires, n); break;
    case OP_POSEIDON: exec_poseidon(t, vals, wires, n); break;
    case OP_BASE_SUM: {
      const u64 row = t[0], x = vals[t[1]];
      W(0, row) = x;
#pragma unroll 1
      for (u32 i = 0; i < BASE_SUM_LIMBS; i++) { const u64 b = (x >> i) & 1; W(1 + i, row) = b; vals[t[2 + i]] = b; }
      break;
    }
    case OP_RA: {
      const u64 row = t[0], c = t[1], idx = vals[t[2]];
      const u32 vs = 1u << RA_BITS, base = (2 + vs) * (u32)c, routed = (2 + vs) * RA_COPIES + 2;
      W(base, row) = idx;
#pragma unroll 1
      for (u32 i = 0; i < vs; i++) W(base + 2 + i, row) = vals[t[3 + i]];
      for (u32 i = 0; i < RA_BITS; i++) W(routed + c * RA_BITS + i, row) = (idx >> i) & 1;
      const u64 o = vals[t[3 + (idx & (vs - 1))]];
      W(base + 1, row) = o;
      vals[t[19]] = o;
      break;
    }
    case OP_REDUCING: case OP_REDUCING_EXT: {
      const bool ext = op == OP_REDUCING_EXT;
      const u32 nc = ext ? RED_EXT_COEFFS : RED_COEFFS, start_accs = 6 + (ext ? 2 * nc : nc);
      const u64 row = t[0];
      const gl2 alpha = gl2_make(vals[t[1]], vals[t[2]]);
      gl2 acc = gl2_make(vals[t[3]], vals[t[4]]);
      W(2, row) = alpha.a; W(3, row) = alpha.b; W(4, row) = acc.a; W(5, row) = acc.b;
#pragma unroll 1
      for (u32 i = 0; i < nc; i++) {
        gl2 cf;
        if (ext) { cf = gl2_make(vals[t[5 + 2 * i]], vals[t[6 + 2 * i]]); W(6 + 2 * i, row) = cf.a; W(7 + 2 * i, row) = cf.b; }
        else { cf = gl2_make(vals[t[5 + i]], 0); W(6 + i, row) = cf.a; }
        acc = gl2_add(gl2_mul(acc, alpha), cf);
        if (i < nc - 1) { W(start_accs + 2 * i, row) = acc.a; W(start_accs + 2 * i + 1, row) = acc.b; }
      }
      W(0, row) = acc.a; W(1, row) = acc.b;
      const u32 o = 5 + (ext ? 2 * nc : nc);
      vals[t[o]] = acc.a; vals[t[o + 1]] = acc.b;
      break;
    }
    case OP_COSET: {
      const u64 row = t[0];
      const u32 bits = (u32)t[1], npts = 1u << bits;
      const u64* dom = domtab + 32 * bits;
      const u64* bw = domtab + 32 * (6 + bits);
      const u32 nint0 = (npts - 2) / 7, deg = (npts - 2) / (nint0 + 1) + 2, nint = (npts - 2) / (deg - 1);
      const u32 w_pt = 1 + 2 * npts, w_val = w_pt + 2, w_int = w_val + 2, w_sh = w_int + 4 * nint;
      const u64 shift = vals[t[2]];
      W(0, row) = shift;
      const u64* v = t + 3;
#pragma unroll 1
      for (u32 i = 0; i < 2 * npts; i++) W(1 + i, row) = vals[v[i]];
      const gl2 pt = gl2_make(vals[v[2 * npts]], vals[v[2 * npts + 1]]);
      W(w_pt, row) = pt.a; W(w_pt + 1, row) = pt.b;
      const gl2 sh = gl2_scale(pt, gl_inv(shift));
      W(w_sh, row) = sh.a; W(w_sh + 1, row) = sh.b;
      gl2 ev = gl2_make(0, 0), pr = gl2_make(1, 0);
      u32 start = 0, endi = deg;
#pragma unroll 1
      for (u32 c = 0; c <= nint; c++) {
#pragma unroll 1
        for (u32 i = start; i < endi; i++) {
          const gl2 val = gl2_scale(gl2_make(vals[v[2 * i]], vals[v[2 * i + 1]]), bw[i]);
          const gl2 term = gl2_make(gl_sub(sh.a, dom[i]), sh.b);
          const gl2 nev = gl2_add(gl2_mul(ev, term), gl2_mul(val, pr));
          pr = gl2_mul(pr, term);
          ev = nev;
        }
        if (c == nint) break;
        W(w_int + 2 * c, row) = ev.a; W(w_int + 2 * c + 1, row) = ev.b;
        W(w_int + 2 * (nint + c), row) = pr.a; W(w_int + 2 * (nint + c) + 1, row) = pr.b;
        start = 1 + (deg - 1) * (c + 1);
        endi = start + deg - 1 < npts ? start + deg - 1 : npts;
      }
      W(w_val, row) = ev.a; W(w_val + 1, row) = ev.b;
      vals[v[2 * npts + 2]] = ev.a; vals[v[2 * npts + 3]] = ev.b;
      break;
    }
    case OP_HINT_DIV_EXT: {
      const gl2 num = gl2_make(vals[t[0]], vals[t[1]]), den = gl2_make(vals[t[2]], vals[t[3]]);
      const gl2 q = gl2_mul(num, gl2_inv(den));
      vals[t[4]] = q.a; vals[t[5]] = q.b;
      break;
    }
    case OP_HINT_LO63: vals[t[1]] = vals[t[0]] & (((u64)1 << 63) - 1); break;
    case OP_HINT_HI: vals[t[1]] = vals[t[0]] >> 63; break;
    case OP_HINT_SPLIT: vals[t[2]] = vals[t[0]] & (((u64)1 << t[1]) - 1); vals[t[3]] = vals[t[0]] >> t[1]; break;
    default:  // the leaf-circuit gates (witness_ops.h: shared with the host executor); anything else was refused at create
      exec_gate_op(op, t, vals, [wires](u64 col, u64 row, u64 v) { W(col, row) = v; });
      break;
  }
}
#undef W

#ifndef WIT_LANES_N
#define WIT_LANES_N 512
#endif
#ifndef WIT_BOUNDS
#define WIT_BOUNDS WIT_LANES_N
#endif
constexpr int WIT_LANES = WIT_LANES_N;
#ifdef WIT_PROF
__device__ u64 g_wit_prof[8];
#endif
__global__ void __launch_bounds__(WIT_BOUNDS) witness_exec_kernel(const u64* __restrict__ tape, const u32* __restrict__ sched,
                                                                const u32* __restrict__ level_off, const u32* __restrict__ level_p2, u32 n_levels, u32 n_slots, u32 log_n,
                                                                const u32* __restrict__ input_sids, u32 n_inputs, const u64* __restrict__ consts,
                                                                u32 n_consts, const u64* __restrict__ domtab, const u32* __restrict__ probe,
                                                                u32 n_probe, const u64* __restrict__ inputs, u64* vals_all, u64* wires_all,
                                                                u64* probe_out) {
  const u32 b = blockIdx.x, tid = threadIdx.x;
  const u64 n = (u64)1 << log_n;
  u64* vals = vals_all + (u64)b * n_slots;
  u64* wires = wires_all + (u64)b * NUM_WIRES * n;
  for (u32 i = tid; i < n_consts; i += WIT_LANES) vals[consts[2 * i]] = consts[2 * i + 1];
  for (u32 i = tid; i < n_inputs; i += WIT_LANES) vals[input_sids[i]] = inputs[(u64)b * n_inputs + i];
  __syncthreads();
#ifdef WIT_PROF
  // tools/dbg/witness_prof.sh: shader cycles of block 0 per class of level (0 narrow Poseidon2, 1 wide Poseidon2, 2 reducing / interpolation /
  // inverse, 3 the rest), left in the side buffer g_wit_prof (read back with mp2g_dbg_witness_prof; nothing of the caller's is touched)
  u64 prof[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
#endif
  for (u32 l = 0; l < n_levels; l++) {
    const u32 lo = level_off[l], hi = level_off[l + 1];
    const u32 p2_lo = level_p2[2 * l], p2_n = level_p2[2 * l + 1];  // the level's Poseidon2 rows are sched[p2_lo .. p2_lo + p2_n)
#ifdef WIT_PROF
    const u64 t_start = __builtin_readcyclecounter();
    u32 cls = p2_n ? (p2_n * 16 <= 2 * WIT_LANES ? 0 : 1) : 3;
    if (!p2_n)
      for (u32 i = lo; i < hi; i++) { const u64 op = tape[sched[i]]; if (op == OP_REDUCING || op == OP_REDUCING_EXT || op == OP_COSET || op == OP_HINT_DIV_EXT) { cls = 2; break; } }
#endif
    if (p2_n && p2_n * 16 <= 2 * WIT_LANES) {
      // few Poseidon2 rows: one 16-lane group each (latency; up to two rounds of groups: 2 x ~23 us against ~62 us one lane per
      // row), the level's other instructions one lane each
      for (u32 g = tid >> 4; g < p2_n; g += WIT_LANES / 16) exec_p2_coop(tape + sched[p2_lo + g] + 1, vals, wires, n, (int)(tid & 15));
      const u32 rest = (hi - lo) - p2_n;
      for (u32 i = tid; i < rest; i += WIT_LANES) {
        const u32 j = lo + i;
        exec_one(tape + sched[j < p2_lo ? j : j + p2_n], vals, wires, n, domtab);
      }
    } else {
      for (u32 i = lo + tid; i < hi; i += WIT_LANES) exec_one(tape + sched[i], vals, wires, n, domtab);
    }
    __syncthreads();  // the level's slot writes (global memory, this block's) are visible to the next level's reads
#ifdef WIT_PROF
    prof[cls] += __builtin_readcyclecounter() - t_start; cnt[cls]++;
#endif
  }
  for (u32 i = tid; i < n_probe; i += WIT_LANES) probe_out[(u64)b * n_probe + i] = vals[probe[i]];
#ifdef WIT_PROF
  if (b == 0 && tid == 0)
    for (int k = 0; k < 4; k++) { g_wit_prof[2 * k] = prof[k]; g_wit_prof[2 * k + 1] = cnt[k]; }
#endif
}
}  // namespace

hipError_t witness_exec_launch(hipStream_t s, const WitnessDev& d, u32 n_levels, u32 n_slots, u32 log_n, u32 n_inputs, u32 n_consts,
                               u32 n_probe, const u64* d_inputs, u32 batch, u64* d_vals, u64* d_wires, u64* d_probe_out) {
  hipLaunchKernelGGL(witness_exec_kernel, dim3(batch), dim3(WIT_LANES), 0, s, d.tape.p, (const u32*)d.sched.p, (const u32*)d.level_off.p,
                     (const u32*)d.level_p2.p, n_levels, n_slots, log_n, (const u32*)d.input_sids.p, n_inputs, d.consts.p, n_consts, d.domtab.p,
                     (const u32*)d.probe.p, n_probe, d_inputs, d_vals, d_wires, d_probe_out);
  return hipGetLastError();
}
}  // namespace mp2g
#ifdef WIT_PROF
// debug builds only (tools/dbg/witness_prof.sh): (cycles, levels) per class of level of block 0 of the last witness launch
extern "C" int mp2g_dbg_witness_prof(uint64_t* out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(mp2g::g_wit_prof), 8 * sizeof(uint64_t)) == hipSuccess ? 0 : 1;
}
#endif
