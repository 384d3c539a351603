// Witness generation for circuits built by recursion.py (host code; no GPU involved).
//
// Replaces [dep] plonky2 iop/generator.rs generate_partial_witness for the circuits of the recursion framework
// (first line of prove(), recursion-framework/src/circuit_builder.rs:308, wrap_circuit.rs:143): the wrap circuit,
// the universal-verifier circuits and their logic. plonky2 walks a dependency graph of per-gate generators; here the
// builder's operations are recorded once, in order, as a straight-line program over value slots (recursion.py OP_*),
// because the structure of these circuits does not depend on the witness. Replaying the program for one input
// vector (the inner proof, its verifier data, the circuit's own inputs) fills the 135 x n wire matrix; a batch of
// input vectors is replayed by a pool of host threads, one proof per thread at a time, straight into the layout
// mp2g_prover_prove_dev takes ([batch][135][n]).
#define MP2G_DEVCONST static const
#include "gl.cuh"
#include "perm_constants.h"
#include "ctx.h"
#include "witness.h"
#include "witness_ops.h"
#include <algorithm>
#include <atomic>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

using namespace mp2g;
#define NEED(c, msg) do { if (!(c)) return fail("invalid argument: %s", msg); } while (0)

namespace {
// operand count after the opcode (t = the operands, left = how many words the tape still holds after the opcode); ~0u = malformed.
// Variable-length instructions read their counts from their first operands: those are checked to be there before they are read.
u32 op_len(u64 op, const u64* t, size_t left = ~(size_t)0) {
  switch (op) {
    case OP_ARITH: return 8;
    case OP_ARITH_EXT: return 12;
    case OP_P2: case OP_POSEIDON: return 1 + 12 + 1 + 12;
    case OP_BASE_SUM: return 2 + BASE_SUM_LIMBS;
    case OP_RA: return 3 + 16 + 1;
    case OP_REDUCING: return 5 + RED_COEFFS + 2;
    case OP_REDUCING_EXT: return 5 + 2 * RED_EXT_COEFFS + 2;
    case OP_COSET: return left >= 2 && t[1] <= 5 ? 3 + (2u << t[1]) + 4 : ~0u;
    case OP_WIRE: return 3;
    case OP_HINT_DIV_EXT: return 6;
    case OP_HINT_LO63: case OP_HINT_HI: return 2;
    case OP_HINT_SPLIT: return 4;  // source slot, bit position, low slot, high slot (split_low_high's LowHighGenerator)
    case OP_PAR: return left >= 1 && t[0] <= 4096 ? 1 + (u32)t[0] : ~0u;  // section count, then the sections' lengths in words; the sections follow
    case OP_U32_ARITH: case OP_U32_SUB: return 8;
    case OP_U32_ADD_MANY: return left >= 4 && t[3] >= 1 && t[3] <= 16 ? 4 + (u32)t[3] + 3 : ~0u;
    case OP_U32_RANGE_CHECK: return 4;
    case OP_COMPARISON: return 6;
    case OP_BASE_SPLIT: return left >= 3 && t[2] >= 1 && t[2] <= 63 ? 4 + (u32)t[2] : ~0u;
    case OP_MUL_EXT: return 9;
    case OP_EXP: return left >= 2 && t[1] >= 1 && t[1] <= 66 ? 3 + (u32)t[1] + 1 : ~0u;
    default: return ~0u;
  }
}

// Poseidon2 linear layers on canonical values (poseidon.cuh keeps weak forms for the device; the witness wants the
// intermediate states, canonical, as the gate's wires)
// circ(2 M4, M4, M4) with M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]: M4 by its addition chain (t0 = x0 + x1, t1 = x2 + x3,
// t2 = 2 x1 + t1, t3 = 2 x3 + t0, t4 = 4 t1 + t3, t5 = 4 t0 + t2, rows = t3 + t5, t5, t2 + t4, t4) in 128-bit integers -- every
// entry of the layer stays below 2^70 -- and ONE reduction per output limb (a third of a host permutation was spent here on
// 16 modular multiply-by-constant and 28 modular additions per block)
void p2_external(u64 s[12]) {
  typedef unsigned __int128 u128;
  u128 t[12];
  for (int c = 0; c < 3; c++) {
    const u128 x0 = s[4 * c], x1 = s[4 * c + 1], x2 = s[4 * c + 2], x3 = s[4 * c + 3];
    const u128 t0 = x0 + x1, t1 = x2 + x3, t2 = 2 * x1 + t1, t3 = 2 * x3 + t0, t4 = 4 * t1 + t3, t5 = 4 * t0 + t2;
    t[4 * c] = t3 + t5; t[4 * c + 1] = t5; t[4 * c + 2] = t2 + t4; t[4 * c + 3] = t4;
  }
  for (int i = 0; i < 4; i++) {
    const u128 sum = t[i] + t[4 + i] + t[8 + i];
    for (int c = 0; c < 3; c++) {
      const u128 v = t[4 * c + i] + sum;
      s[4 * c + i] = gl_reduce128((u64)v, (u64)(v >> 64));
    }
  }
}
// x_i <- mu_i x_i + sum_j x_j: the sum and each product in 128 bits, one reduction per limb
void p2_internal(u64 s[12]) {
  typedef unsigned __int128 u128;
  u128 sum = 0;
  for (int i = 0; i < 12; i++) sum += s[i];
  const u64 sr = gl_reduce128((u64)sum, (u64)(sum >> 64));
  for (int i = 0; i < 12; i++) {
    const u128 v = (u128)s[i] * POSEIDON2_DIAG_M1[i];
    // lo + hi 2^64 + sr: add sr into the low word with carry into hi (hi <= 2^64 - 2 for canonical factors)
    u64 lo = (u64)v, hi = (u64)(v >> 64);
    const u64 l2 = lo + sr;
    hi += l2 < lo ? 1 : 0;
    s[i] = gl_reduce128(l2, hi);
  }
}
// Poseidon's MDS layer on canonical values: circ [17,15,41,16,2,28,13,13,39,18,34,20] + diag [8,0,...]; entries < 2^6, so a row sum
// stays below 2^74: one reduction per output limb
void poseidon_mds_host(u64 s[12]) {
  typedef unsigned __int128 u128;
  u64 out[12];
  for (int r = 0; r < 12; r++) {
    u128 acc = (u128)s[r] * POSEIDON_MDS_DIAG[r];
    for (int i = 0; i < 12; i++) acc += (u128)s[(i + r) % 12] * POSEIDON_MDS_CIRC[i];
    out[r] = gl_reduce128((u64)acc, (u64)(acc >> 64));
  }
  for (int i = 0; i < 12; i++) s[i] = out[i];
}
}  // namespace

namespace {
// the instructions of [t, end). A parallel region (OP_PAR) is a run of sections that read what came before the region and each
// other's nothing: the builder brackets the query rounds of a FRI verifier this way (28 per verified proof). With inner > 1 the
// sections of a region are dealt to that many threads; every instruction writes value slots and wire cells of its own.
void exec(const mp2g_witness_program& P, const u64* t, const u64* end, u64* vals, u64* wires, u32 inner, u64 cs, u64 rs) {
  // wire (col, row) sits at col * cs + row * rs: (n, 1) = the prover's polynomial-major matrix, (1, 135) = one contiguous row per
  // gate row (what the host writes fastest: a Poseidon2 row is 135 consecutive words instead of 135 cache lines)
#define W(col, row) wires[(u64)(col) * cs + (u64)(row) * rs]
  while (t < end) {
    const u64 op = *t++;
    switch (op) {
      case OP_PAR: {
        const u32 ns = (u32)t[0];
        const u64* len = t + 1;
        const u64* body = t + 1 + ns;
        u64 total = 0;
        for (u32 i = 0; i < ns; i++) total += len[i];
        if (inner <= 1 || ns <= 1) {
          exec(P, body, body + total, vals, wires, 1, cs, rs);
        } else {
          std::vector<const u64*> start(ns + 1);
          start[0] = body;
          for (u32 i = 0; i < ns; i++) start[i + 1] = start[i] + len[i];
          std::atomic<u32> next{0};
          auto worker = [&]() {
            for (;;) {
              const u32 i = next.fetch_add(1);
              if (i >= ns) return;
              exec(P, start[i], start[i + 1], vals, wires, 1, cs, rs);
            }
          };
          const u32 nt = inner < ns ? inner : ns;
          std::vector<std::thread> pool;
          try {
            for (u32 i = 1; i < nt; i++) pool.emplace_back(worker);
          } catch (...) {
            // fewer threads than hoped for: the sections are dealt from a shared counter
          }
          worker();
          for (auto& th : pool) th.join();
        }
        t = body + total;
        break;
      }
      case OP_WIRE: W(t[1], t[0]) = vals[t[2]]; t += 3; break;
      case OP_ARITH: {
        const u64 row = t[0], i = t[1], c0 = t[2], c1 = t[3];
        const u64 m0 = vals[t[4]], m1 = vals[t[5]], ad = vals[t[6]];
        const u64 o = gl_add(gl_mul(gl_mul(m0, m1), c0), gl_mul(ad, c1));
        W(4 * i, row) = m0; W(4 * i + 1, row) = m1; W(4 * i + 2, row) = ad; W(4 * i + 3, row) = o;
        vals[t[7]] = o;
        t += 8;
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
        t += 12;
        break;
      }
      case OP_P2: {  // Poseidon2Gate: inputs 0..11, outputs 12..23, swap 24, deltas 25..28, S-box inputs 29.., 65.., 87..
        const u64 row = t[0];
        u64 in[12], s[12];
        for (int i = 0; i < 12; i++) { in[i] = vals[t[1 + i]]; W(i, row) = in[i]; }
        const u64 swap = vals[t[13]];
        W(24, row) = swap;
        for (int i = 0; i < 4; i++) {
          const u64 delta = gl_mul(swap, gl_sub(in[i + 4], in[i]));
          W(25 + i, row) = delta;
          s[i] = gl_add(in[i], delta);
          s[i + 4] = gl_sub(in[i + 4], delta);
        }
        for (int i = 8; i < 12; i++) s[i] = in[i];
        p2_external(s);
        for (int r = 0; r < 4; r++) {
          for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], POSEIDON2_RC_EXT[12 * r + i]);
          if (r) for (int i = 0; i < 12; i++) W(29 + 12 * (r - 1) + i, row) = s[i];
          for (int i = 0; i < 12; i++) s[i] = gl_pow7(s[i]);
          p2_external(s);
        }
        for (int r = 0; r < 22; r++) {
          s[0] = gl_add(s[0], POSEIDON2_RC_INT[r]);
          W(65 + r, row) = s[0];
          s[0] = gl_pow7(s[0]);
          p2_internal(s);
        }
        for (int r = 0; r < 4; r++) {
          for (int i = 0; i < 12; i++) { s[i] = gl_add(s[i], POSEIDON2_RC_EXT[12 * (4 + r) + i]); W(87 + 12 * r + i, row) = s[i]; }
          for (int i = 0; i < 12; i++) s[i] = gl_pow7(s[i]);
          p2_external(s);
        }
        for (int i = 0; i < 12; i++) { W(12 + i, row) = s[i]; vals[t[14 + i]] = s[i]; }
        t += 26;
        break;
      }
      case OP_POSEIDON: {  // PoseidonGate: the wire layout of the Poseidon2 gate, the original permutation (30 rounds, 4 + 22 + 4)
        const u64 row = t[0];
        u64 in[12], s[12];
        for (int i = 0; i < 12; i++) { in[i] = vals[t[1 + i]]; W(i, row) = in[i]; }
        const u64 swap = vals[t[13]];
        W(24, row) = swap;
        for (int i = 0; i < 4; i++) {
          const u64 delta = gl_mul(swap, gl_sub(in[i + 4], in[i]));
          W(25 + i, row) = delta;
          s[i] = gl_add(in[i], delta);
          s[i + 4] = gl_sub(in[i + 4], delta);
        }
        for (int i = 8; i < 12; i++) s[i] = in[i];
        for (int r = 0; r < 30; r++) {
          for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], POSEIDON_RC[12 * r + i]);
          if (r >= 4 && r < 26) {
            W(65 + r - 4, row) = s[0];
            s[0] = gl_pow7(s[0]);
          } else {
            if (r) { const int base = r < 4 ? 29 + 12 * (r - 1) : 87 + 12 * (r - 26); for (int i = 0; i < 12; i++) W(base + i, row) = s[i]; }
            for (int i = 0; i < 12; i++) s[i] = gl_pow7(s[i]);
          }
          poseidon_mds_host(s);
        }
        for (int i = 0; i < 12; i++) { W(12 + i, row) = s[i]; vals[t[14 + i]] = s[i]; }
        t += 26;
        break;
      }
      case OP_BASE_SUM: {
        const u64 row = t[0], x = vals[t[1]];
        W(0, row) = x;
        for (u32 i = 0; i < BASE_SUM_LIMBS; i++) { const u64 b = (x >> i) & 1; W(1 + i, row) = b; vals[t[2 + i]] = b; }
        t += 2 + BASE_SUM_LIMBS;
        break;
      }
      case OP_RA: {
        const u64 row = t[0], c = t[1], idx = vals[t[2]];
        const u32 vs = 1u << RA_BITS, base = (2 + vs) * (u32)c, routed = (2 + vs) * RA_COPIES + 2;
        W(base, row) = idx;
        for (u32 i = 0; i < vs; i++) W(base + 2 + i, row) = vals[t[3 + i]];
        for (u32 i = 0; i < RA_BITS; i++) W(routed + c * RA_BITS + i, row) = (idx >> i) & 1;
        const u64 o = vals[t[3 + (idx & (vs - 1))]];
        W(base + 1, row) = o;
        vals[t[19]] = o;
        t += 20;
        break;
      }
      case OP_REDUCING: case OP_REDUCING_EXT: {
        const bool ext = op == OP_REDUCING_EXT;
        const u32 nc = ext ? RED_EXT_COEFFS : RED_COEFFS, start_accs = 6 + (ext ? 2 * nc : nc);
        const u64 row = t[0];
        const gl2 alpha = gl2_make(vals[t[1]], vals[t[2]]);
        gl2 acc = gl2_make(vals[t[3]], vals[t[4]]);
        W(2, row) = alpha.a; W(3, row) = alpha.b; W(4, row) = acc.a; W(5, row) = acc.b;
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
        t += o + 2;
        break;
      }
      case OP_COSET: {
        const u64 row = t[0];
        const u32 bits = (u32)t[1], npts = 1u << bits;
        // CosetInterpolationGate::with_max_degree(bits, 8)
        const u32 nint0 = (npts - 2) / 7, deg = (npts - 2) / (nint0 + 1) + 2, nint = (npts - 2) / (deg - 1);
        const u32 w_pt = 1 + 2 * npts, w_val = w_pt + 2, w_int = w_val + 2, w_sh = w_int + 4 * nint;
        const u64 shift = vals[t[2]];
        W(0, row) = shift;
        const u64* v = t + 3;
        for (u32 i = 0; i < 2 * npts; i++) W(1 + i, row) = vals[v[i]];
        const gl2 pt = gl2_make(vals[v[2 * npts]], vals[v[2 * npts + 1]]);
        W(w_pt, row) = pt.a; W(w_pt + 1, row) = pt.b;
        const gl2 sh = gl2_scale(pt, gl_inv(shift));
        W(w_sh, row) = sh.a; W(w_sh + 1, row) = sh.b;
        gl2 ev = gl2_make(0, 0), pr = gl2_make(1, 0);
        u32 start = 0, endi = deg;
        for (u32 c = 0; c <= nint; c++) {
          for (u32 i = start; i < endi; i++) {
            const gl2 val = gl2_scale(gl2_make(vals[v[2 * i]], vals[v[2 * i + 1]]), P.bw[bits][i]);
            const gl2 term = gl2_make(gl_sub(sh.a, P.dom[bits][i]), sh.b);
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
        t += 3 + 2 * npts + 4;
        break;
      }
      case OP_HINT_DIV_EXT: {
        const gl2 num = gl2_make(vals[t[0]], vals[t[1]]), den = gl2_make(vals[t[2]], vals[t[3]]);
        const gl2 q = gl2_mul(num, gl2_inv(den));
        vals[t[4]] = q.a; vals[t[5]] = q.b;
        t += 6;
        break;
      }
      case OP_HINT_LO63: vals[t[1]] = vals[t[0]] & (((u64)1 << 63) - 1); t += 2; break;
      case OP_HINT_HI: vals[t[1]] = vals[t[0]] >> 63; t += 2; break;
      case OP_HINT_SPLIT: vals[t[2]] = vals[t[0]] & (((u64)1 << t[1]) - 1); vals[t[3]] = vals[t[0]] >> t[1]; t += 4; break;
      default:  // the leaf-circuit gates (witness_ops.h: shared with the device executor)
        if (!exec_gate_op(op, t, vals, [&](u64 col, u64 row, u64 v) { W(col, row) = v; })) return;  // validated at create
        t += op_len(op, t);
        break;
    }
  }
#undef W
}
// one proof: vals = scratch of n_slots words, wires = [135][n] (zero-filled here)
void run_one(const mp2g_witness_program& P, const u64* inputs, u64* vals, u64* wires, u32 inner, bool rows) {
  const u64 n = (u64)1 << P.log_n;
  memset(wires, 0, NUM_WIRES * n * sizeof(u64));
  for (size_t i = 0; i < P.consts.size(); i += 2) vals[P.consts[i]] = P.consts[i + 1];
  for (size_t i = 0; i < P.input_sids.size(); i++) vals[P.input_sids[i]] = inputs[i];
  exec(P, P.tape.data(), P.tape.data() + P.tape.size(), vals, wires, inner, rows ? 1 : n, rows ? NUM_WIRES : 1);
}
}  // namespace

extern "C" {
static int witness_program_create(const uint64_t* tape, size_t tape_len, uint32_t n_slots, uint32_t log_n, const uint32_t* input_sids,
                                  uint32_t n_inputs, const uint64_t* const_slots, uint32_t n_consts, mp2g_witness_program** out) {
  NEED(out && (tape || !tape_len) && (input_sids || !n_inputs) && (const_slots || !n_consts), "pointers");
  NEED(log_n >= 1 && log_n <= 20 && n_slots >= 1, "log_n / n_slots");
  mp2g_witness_program* P = new (std::nothrow) mp2g_witness_program();
  if (!P) return fail("out of memory");
  P->tape.assign(tape, tape + tape_len);
  P->input_sids.assign(input_sids, input_sids + n_inputs);
  P->consts.assign(const_slots, const_slots + 2 * (size_t)n_consts);
  P->n_slots = n_slots; P->log_n = log_n;
  const u64 n = (u64)1 << log_n;
  // validate: opcodes, lengths, slot and row / column bounds -- the tape indexes host memory
  auto bad = [&](const char* msg) { delete P; return fail("invalid witness program: %s", msg); };
  for (uint32_t i = 0; i < n_inputs; i++) if (input_sids[i] >= n_slots) return bad("input slot out of range");
  for (uint32_t i = 0; i < n_consts; i++) if (const_slots[2 * i] >= n_slots || const_slots[2 * i + 1] >= GL_P) return bad("constant slot / value");
  const u64* t = P->tape.data();
  const u64* end = t + tape_len;
  const u64* par_end = nullptr;        // end of the parallel region being scanned (regions do not nest)
  std::vector<const u64*> boundaries;  // its section boundaries: each must fall on an instruction start
  size_t next_boundary = 0;
  while (t < end) {
    if (par_end) {
      while (next_boundary < boundaries.size() && t == boundaries[next_boundary]) next_boundary++;
      if (next_boundary < boundaries.size() && t > boundaries[next_boundary]) return bad("a parallel section ends inside an instruction");
      if (t >= par_end) { par_end = nullptr; boundaries.clear(); next_boundary = 0; }
    }
    const u64 op = *t++;
    if (op < OP_ARITH || op >= OP_END) return bad("unknown opcode");
    if (op == OP_COSET && (t + 2 > end || t[1] < 2 || t[1] > 5)) return bad("CosetInterpolation bits");
    if (op == OP_PAR && (t + 1 > end || par_end)) return bad("parallel region header / nesting");
    const u32 len = op_len(op, t, (size_t)(end - t));
    if (len == ~0u || t + len > end) return bad("truncated instruction");
    if (op == OP_PAR) {
      const u64* body = t + len;
      const u64* b = body;
      for (u32 i = 1; i < len; i++) {
        if (t[i] > (u64)(end - b)) return bad("parallel section length");
        b += t[i];
        boundaries.push_back(b);
      }
      par_end = b;
      next_boundary = 0;
      t += len;
      continue;
    }
    // operand classes: rows < n; everything else that is not a constant or a small index is a slot
    u32 first_slot = 0;
    switch (op) {
      case OP_ARITH: case OP_ARITH_EXT: if (t[0] >= n || t[1] >= (op == OP_ARITH ? 20u : 10u) || t[2] >= GL_P || t[3] >= GL_P) return bad("arithmetic operands"); first_slot = 4; break;
      case OP_P2: case OP_POSEIDON: case OP_BASE_SUM: case OP_REDUCING: case OP_REDUCING_EXT: if (t[0] >= n) return bad("row"); first_slot = 1; break;
      case OP_RA: if (t[0] >= n || t[1] >= RA_COPIES) return bad("random access operands"); first_slot = 2; break;
      case OP_COSET: if (t[0] >= n) return bad("row"); first_slot = 2; break;
      case OP_WIRE: if (t[0] >= n || t[1] >= NUM_WIRES) return bad("wire"); first_slot = 2; break;
      case OP_HINT_SPLIT: if (t[0] >= n_slots || t[1] < 1 || t[1] > 63) return bad("split hint"); first_slot = 2; break;
      case OP_U32_ARITH: if (t[0] >= n || t[2] < 1 || t[2] > 3 || t[1] >= t[2]) return bad("U32Arithmetic operands"); first_slot = 3; break;
      case OP_U32_SUB: if (t[0] >= n || t[2] < 1 || t[2] > 6 || t[1] >= t[2]) return bad("U32Subtraction operands"); first_slot = 3; break;
      case OP_U32_ADD_MANY: if (t[0] >= n || t[2] < 1 || t[1] >= t[2] || (t[3] + 3 + 18) * t[2] > NUM_WIRES) return bad("U32AddMany operands"); first_slot = 4; break;
      case OP_U32_RANGE_CHECK: if (t[0] >= n || t[2] < 1 || t[2] > 7 || t[1] >= t[2]) return bad("U32RangeCheck operands"); first_slot = 3; break;
      case OP_COMPARISON: {
        if (t[0] >= n || t[1] < 1 || t[1] > 63 || t[2] < 1 || t[2] > 16) return bad("Comparison operands");
        const u64 cb = (t[1] + t[2] - 1) / t[2];
        if (4 + 5 * t[2] + cb + 1 > NUM_WIRES) return bad("Comparison operands");
        first_slot = 3;
        break;
      }
      case OP_BASE_SPLIT: if (t[0] >= n || t[1] < 1 || t[1] > 2 || t[1] * t[2] > 63) return bad("BaseSplit operands"); first_slot = 3; break;
      case OP_MUL_EXT: if (t[0] >= n || t[1] >= 13 || t[2] >= GL_P) return bad("MulExtension operands"); first_slot = 3; break;
      case OP_EXP: if (t[0] >= n) return bad("row"); first_slot = 2; break;
      default: first_slot = 0; break;
    }
    for (u32 i = first_slot; i < len; i++) if (t[i] >= n_slots) return bad("slot out of range");
    t += len;
  }
  for (; par_end && next_boundary < boundaries.size(); next_boundary++)
    if (boundaries[next_boundary] != end) return bad("a parallel section ends inside an instruction");
  for (u32 bits = 2; bits <= 5; bits++) {
    const u32 npts = 1u << bits;
    const u64 w = gl_root_of_unity(bits);
    u64 x = 1;
    for (u32 i = 0; i < npts; i++) { P->dom[bits][i] = x; x = gl_mul(x, w); }
    for (u32 i = 0; i < npts; i++) {
      u64 pr = 1;
      for (u32 j = 0; j < npts; j++) if (j != i) pr = gl_mul(pr, gl_sub(P->dom[bits][i], P->dom[bits][j]));
      P->bw[bits][i] = gl_inv(pr);
    }
  }
  // level schedule for the device executor (witness_dev.hip): level(i) = 1 + max level of the slots i reads
  {
    struct Ins { u32 off, lvl, op; };
    std::vector<Ins> ins;
    std::vector<u32> lvl(n_slots, 0);
    std::vector<uint8_t> written(n_slots, 0), was_read(n_slots, 0);
    for (uint32_t i = 0; i < n_inputs; i++) written[input_sids[i]] = 1;
    for (uint32_t i = 0; i < n_consts; i++) written[const_slots[2 * i]] = 1;
    const u64* base = P->tape.data();
    u32 max_lvl = 0;
    for (const u64* q = base; q < end;) {
      const u64 op = *q;
      const u64* a = q + 1;
      const u32 len = op_len(op, a);
      if (op == OP_PAR) { q = a + len; continue; }  // the sections follow as ordinary instructions
      u32 r0 = 0, nr = 0, w0 = 0, nw = 0;
      switch (op) {
        case OP_ARITH: r0 = 4; nr = 3; w0 = 7; nw = 1; break;
        case OP_ARITH_EXT: r0 = 4; nr = 6; w0 = 10; nw = 2; break;
        case OP_P2: case OP_POSEIDON: r0 = 1; nr = 13; w0 = 14; nw = 12; break;
        case OP_BASE_SUM: r0 = 1; nr = 1; w0 = 2; nw = BASE_SUM_LIMBS; break;
        case OP_RA: r0 = 2; nr = 17; w0 = 19; nw = 1; break;
        case OP_REDUCING: r0 = 1; nr = 4 + RED_COEFFS; w0 = 5 + RED_COEFFS; nw = 2; break;
        case OP_REDUCING_EXT: r0 = 1; nr = 4 + 2 * RED_EXT_COEFFS; w0 = 5 + 2 * RED_EXT_COEFFS; nw = 2; break;
        case OP_COSET: r0 = 2; nr = 3 + (2u << a[1]); w0 = 5 + (2u << a[1]); nw = 2; break;
        case OP_WIRE: r0 = 2; nr = 1; break;
        case OP_HINT_DIV_EXT: r0 = 0; nr = 4; w0 = 4; nw = 2; break;
        case OP_HINT_LO63: case OP_HINT_HI: r0 = 0; nr = 1; w0 = 1; nw = 1; break;
        case OP_HINT_SPLIT: r0 = 0; nr = 1; w0 = 2; nw = 2; break;
        case OP_U32_ARITH: case OP_U32_SUB: r0 = 3; nr = 3; w0 = 6; nw = 2; break;
        case OP_U32_ADD_MANY: r0 = 4; nr = (u32)a[3] + 1; w0 = 5 + (u32)a[3]; nw = 2; break;
        case OP_U32_RANGE_CHECK: r0 = 3; nr = 1; break;
        case OP_COMPARISON: r0 = 3; nr = 2; w0 = 5; nw = 1; break;
        case OP_BASE_SPLIT: r0 = 3; nr = 1; w0 = 4; nw = (u32)a[2]; break;
        case OP_MUL_EXT: r0 = 3; nr = 4; w0 = 7; nw = 2; break;
        case OP_EXP: r0 = 2; nr = 1 + (u32)a[1]; w0 = 3 + (u32)a[1]; nw = 1; break;
        default: break;
      }
      u32 l = 0;
      for (u32 i = 0; i < nr; i++) { const u32 sl = (u32)a[r0 + i]; if (lvl[sl] > l) l = lvl[sl]; was_read[sl] = 1; }
      l += 1;
      for (u32 i = 0; i < nw; i++) {
        const u32 sl = (u32)a[w0 + i];
        // a slot written twice, or written after an earlier instruction read it (the host replay saw 0 there): the level
        // schedule would reorder the accesses, so the device replay refuses the program
        if (written[sl] || was_read[sl]) P->ssa = false;
        written[sl] = 1;
        lvl[sl] = l;
      }
      if ((size_t)(q - base) > 0xFFFFFFFFu) return bad("tape too long");
      ins.push_back({(u32)(q - base), l, (u32)op});
      if (l > max_lvl) max_lvl = l;
      q = a + len;
    }
    std::stable_sort(ins.begin(), ins.end(), [](const Ins& x, const Ins& y) { return x.lvl != y.lvl ? x.lvl < y.lvl : x.op < y.op; });
    P->sched.resize(ins.size());
    P->level_off.assign(max_lvl + 1, 0);
    for (size_t i = 0; i < ins.size(); i++) { P->sched[i] = ins[i].off; P->level_off[ins[i].lvl]++; }
    // counts per level (index 1..max_lvl) -> offsets: level l's instructions are sched[level_off[l-1] .. level_off[l])
    u32 acc = 0;
    for (u32 l = 0; l <= max_lvl; l++) { acc += P->level_off[l]; P->level_off[l] = acc; }
    P->level_p2.assign(2 * (size_t)max_lvl, 0);
    for (size_t i = 0; i < ins.size(); i++)
      if (ins[i].op == OP_P2) {
        u32* e = &P->level_p2[2 * (ins[i].lvl - 1)];
        if (!e[1]) e[0] = (u32)i;
        e[1]++;
      }
  }
  *out = P;
  return 0;
}
// no C++ exception crosses the C ABI: allocation and thread-creation failures become error returns
int mp2g_witness_program_create(const uint64_t* tape, size_t tape_len, uint32_t n_slots, uint32_t log_n, const uint32_t* input_sids,
                                uint32_t n_inputs, const uint64_t* const_slots, uint32_t n_consts, mp2g_witness_program** out) {
  try {
    return witness_program_create(tape, tape_len, n_slots, log_n, input_sids, n_inputs, const_slots, n_consts, out);
  } catch (const std::bad_alloc&) {
    return fail("out of memory while building the witness program");
  } catch (...) {
    return fail("witness program creation failed");
  }
}
uint32_t mp2g_witness_program_num_inputs(const mp2g_witness_program* P) { return P ? (uint32_t)P->input_sids.size() : 0; }
uint32_t mp2g_witness_program_num_levels(const mp2g_witness_program* P) { return P && !P->level_off.empty() ? (uint32_t)P->level_off.size() - 1 : 0; }
int mp2g_witness_program_set_probe(mp2g_witness_program* P, const uint32_t* probe_sids, uint32_t n_probe) {
  NEED(P && (probe_sids || !n_probe), "program / probe");
  for (uint32_t i = 0; i < n_probe; i++) NEED(probe_sids[i] < P->n_slots, "probe slot");
  std::lock_guard<std::mutex> g(P->dev_mu);
  NEED(P->dev.empty(), "set the probe before the first device run");
  P->probe.assign(probe_sids, probe_sids + n_probe);
  return 0;
}
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail("%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
// the program's read-only data on the context's device, uploaded once
static int witness_dev_data(mp2g_witness_program* P, mp2g_ctx* c, WitnessDev** out) {
  std::lock_guard<std::mutex> g(P->dev_mu);
  for (auto* d : P->dev)
    if (d->device == c->device) { *out = d; return 0; }
  WitnessDev* d = new (std::nothrow) WitnessDev();
  if (!d) return fail("out of memory");
  d->device = c->device;
  auto up = [&](DevBuf& b, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = b.alloc(bytes);
    if (e == hipSuccess && bytes) e = hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
    return e;
  };
  hipError_t e = up(d->tape, P->tape.data(), P->tape.size() * 8);
  if (e == hipSuccess) e = up(d->sched, P->sched.data(), P->sched.size() * 4);
  if (e == hipSuccess) e = up(d->level_off, P->level_off.data(), P->level_off.size() * 4);
  if (e == hipSuccess) e = up(d->level_p2, P->level_p2.data(), P->level_p2.size() * 4);
  if (e == hipSuccess) e = up(d->input_sids, P->input_sids.data(), P->input_sids.size() * 4);
  if (e == hipSuccess) e = up(d->consts, P->consts.data(), P->consts.size() * 8);
  if (e == hipSuccess) e = up(d->probe, P->probe.data(), P->probe.size() * 4);
  if (e == hipSuccess) {
    std::vector<u64> tab(2 * 6 * 32, 0);
    memcpy(tab.data(), P->dom, sizeof(P->dom));
    memcpy(tab.data() + 6 * 32, P->bw, sizeof(P->bw));
    e = up(d->domtab, tab.data(), tab.size() * 8);
  }
  if (e != hipSuccess) { delete d; return fail("witness program upload: %s", hipGetErrorString(e)); }
  P->dev.push_back(d);
  *out = d;
  return 0;
}
int mp2g_witness_program_run_dev(mp2g_witness_program* P, mp2g_ctx* c, const uint64_t* d_inputs, uint32_t batch, uint64_t* d_wires,
                                 uint64_t* d_probe_out) {
  NEED(P && c && d_inputs && d_wires && batch >= 1, "program / ctx / inputs / wires");
  NEED(P->probe.empty() || d_probe_out, "probe output");
  if (!P->ssa) return fail("the witness program writes a slot twice, or writes a slot an earlier instruction read: it cannot be level-scheduled for the device");
  WitnessDev* d = nullptr;
  int rc = witness_dev_data(P, c, &d);
  if (rc) return rc;
  const size_t vals_words = (size_t)batch * P->n_slots, wire_words = (size_t)batch * NUM_WIRES << P->log_n;
  if (c->wit_vals.bytes < vals_words * 8 || c->wit_rows.bytes < wire_words * 8) {
    CKH(hipStreamSynchronize(c->stream));  // a kernel queued earlier may still use the old buffers
    if (c->wit_vals.bytes < vals_words * 8) CKH(c->wit_vals.alloc(vals_words * 8));
    if (c->wit_rows.bytes < wire_words * 8) CKH(c->wit_rows.alloc(wire_words * 8));
  }
  CKH(hipMemsetAsync(c->wit_vals.p, 0, vals_words * 8, c->stream));
  CKH(hipMemsetAsync(c->wit_rows.p, 0, wire_words * 8, c->stream));
  // the executor fills a row-major staging matrix (one contiguous run of words per gate row); the prover's polynomial-major
  // [batch][135][n] is made from it by the tiled transpose (every word of d_wires is written)
  CKH(witness_exec_launch(c->stream, *d, (u32)P->level_off.size() - 1, P->n_slots, P->log_n, (u32)P->input_sids.size(), (u32)(P->consts.size() / 2),
                          (u32)P->probe.size(), (const u64*)d_inputs, batch, c->wit_vals.p, c->wit_rows.p, (u64*)d_probe_out));
  {
    int rc2 = mp2g_wires_from_rows_dev(c, c->wit_rows.p, d_wires, P->log_n, NUM_WIRES, batch);
    if (rc2) return rc2;
  }
  return 0;
}
static int witness_run(const mp2g_witness_program* P, const uint64_t* inputs, uint32_t batch, uint32_t threads, uint64_t* wires,
                       const uint32_t* probe_sids, uint32_t n_probe, uint64_t* probe_out, bool rows) {
  NEED(P && inputs && wires && batch >= 1, "program / inputs / wires");
  NEED(!n_probe || (probe_sids && probe_out), "probe");
  for (uint32_t i = 0; i < n_probe; i++) NEED(probe_sids[i] < P->n_slots, "probe slot");
  const size_t n_in = P->input_sids.size();
  for (size_t i = 0; i < (size_t)batch * n_in; i++) NEED(inputs[i] < GL_P, "inputs must be canonical field elements");
  const size_t per = (size_t)NUM_WIRES << P->log_n;
  if (!threads) threads = std::thread::hardware_concurrency();
  if (!threads) threads = 1;
  // fewer proofs than threads: the spare ones work inside the proofs, on the sections of their parallel regions
  const uint32_t inner = threads > batch ? (threads / batch > 16 ? 16 : threads / batch) : 1;  // 16: past that the thread starts cost more than they save
  if (threads > batch) threads = batch;
  std::atomic<uint32_t> next{0};
  auto worker = [&]() {
    std::vector<u64> vals(P->n_slots);
    for (;;) {
      const uint32_t b = next.fetch_add(1);
      if (b >= batch) return;
      std::fill(vals.begin(), vals.end(), 0);
      run_one(*P, inputs + (size_t)b * n_in, vals.data(), wires + (size_t)b * per, inner, rows);
      for (uint32_t i = 0; i < n_probe; i++) probe_out[(size_t)b * n_probe + i] = vals[probe_sids[i]];
    }
  };
  std::atomic<int> failed{0};
  auto guarded = [&]() {
    try { worker(); } catch (...) { failed.store(1); }  // bad_alloc of a slot table / an inner pool: report, do not terminate
  };
  std::vector<std::thread> pool;
  try {
    for (uint32_t i = 1; i < threads; i++) pool.emplace_back(guarded);
  } catch (...) {
    // the host refuses more threads: go on with the ones that started (the work queue is shared)
  }
  guarded();
  for (auto& th : pool) th.join();
  if (failed.load()) return fail("witness replay: out of memory or thread creation failed");
  return 0;
}
int mp2g_witness_program_run(const mp2g_witness_program* P, const uint64_t* inputs, uint32_t batch, uint32_t threads, uint64_t* wires,
                             const uint32_t* probe_sids, uint32_t n_probe, uint64_t* probe_out) {
  return witness_run(P, inputs, batch, threads, wires, probe_sids, n_probe, probe_out, false);
}
int mp2g_witness_program_run_rows(const mp2g_witness_program* P, const uint64_t* inputs, uint32_t batch, uint32_t threads, uint64_t* rows,
                                  const uint32_t* probe_sids, uint32_t n_probe, uint64_t* probe_out) {
  return witness_run(P, inputs, batch, threads, rows, probe_sids, n_probe, probe_out, true);
}
void mp2g_witness_program_free(mp2g_witness_program* P) { delete P; }
}  // extern "C"
