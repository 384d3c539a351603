// Gate constraint evaluation for the quotient polynomial.
//
// Replaces [dep] plonky2 plonk/vanishing_poly.rs evaluate_gate_constraints_base_batch, gates/gate.rs
// eval_filtered_base_batch / compute_filter and the eval_unfiltered_base of the gates listed in
// include/mp2g.h, as reached from compute_quotient_polys inside prove()
// (recursion-framework/src/circuit_builder.rs:308). One lane per LDE point; the ~135 wire and the few
// constant values of a point are 8 B/lane coalesced streams of the polynomial-major LDE matrices, read
// on demand by the gate that needs them (L2 absorbs wires shared by several gates). plonky2 sums
// filter_g * c_{g,j} into slot j and alpha-reduces the slots; here each gate alpha-reduces its own
// constraints and the filter multiplies the reduced value -- the same field element with
// (#constraints - 1) fewer multiplications per gate and challenge.
#include "gates.h"
#include <cstdlib>
#include "poseidon.cuh"

namespace mp2g {

u32 gate_num_constraints(const mp2g_gate& g) {
  switch (g.kind) {
    case MP2G_GATE_CONSTANT: return g.p0;
    case MP2G_GATE_PUBLIC_INPUT: return 4;
    case MP2G_GATE_ARITHMETIC: return g.p0;
    case MP2G_GATE_BASE_SUM: return 1 + g.p0;
    case MP2G_GATE_ARITHMETIC_EXT: case MP2G_GATE_MUL_EXT: return 2 * g.p0;
    case MP2G_GATE_POSEIDON2: case MP2G_GATE_POSEIDON: return 1 + 4 + 36 + 22 + 48 + 12;
    case MP2G_GATE_POSEIDON_MDS: return 24;
    case MP2G_GATE_COSET_INTERPOLATION: return 4 + 4 * (((1u << g.p0) - 2) / (g.p1 - 1));
    case MP2G_GATE_U32_ARITHMETIC: return 36 * g.p0;
    case MP2G_GATE_U32_RANGE_CHECK: return 17 * g.p0;
    case MP2G_GATE_U32_SUBTRACTION: return 19 * g.p0;
    case MP2G_GATE_U32_ADD_MANY: return 21 * g.p1;
    case MP2G_GATE_COMPARISON: return g.p1 ? 6 + 5 * g.p1 + (g.p0 + g.p1 - 1) / g.p1 : 0;
    case MP2G_GATE_EXPONENTIATION: return g.p0 + 1;
    case MP2G_GATE_REDUCING: case MP2G_GATE_REDUCING_EXT: return 2 * g.p0;
    case MP2G_GATE_RANDOM_ACCESS: return (g.p0 + 2) * g.p1 + g.p2;
    case MP2G_GATE_U32_INTERLEAVE: return 34 * g.p0;
    case MP2G_GATE_UNINTERLEAVE_TO_B32: case MP2G_GATE_UNINTERLEAVE_TO_U32: return 67 * g.p0;
    default: return 0;  // Noop, Lookup, LookupTable
  }
}
u32 gate_degree(const mp2g_gate& g) {
  switch (g.kind) {
    case MP2G_GATE_CONSTANT: case MP2G_GATE_PUBLIC_INPUT: return 1;
    case MP2G_GATE_ARITHMETIC: case MP2G_GATE_ARITHMETIC_EXT: case MP2G_GATE_MUL_EXT: return 3;
    case MP2G_GATE_BASE_SUM: return g.p1;
    case MP2G_GATE_POSEIDON2: case MP2G_GATE_POSEIDON: return 7;
    case MP2G_GATE_POSEIDON_MDS: return 1;
    case MP2G_GATE_COSET_INTERPOLATION: return g.p1;
    case MP2G_GATE_U32_ARITHMETIC: case MP2G_GATE_U32_RANGE_CHECK: case MP2G_GATE_U32_SUBTRACTION: case MP2G_GATE_U32_ADD_MANY: return 4;
    case MP2G_GATE_COMPARISON: return g.p1 ? 1u << ((g.p0 + g.p1 - 1) / g.p1) : 0;
    case MP2G_GATE_EXPONENTIATION: return 4;
    case MP2G_GATE_REDUCING: case MP2G_GATE_REDUCING_EXT: return 2;
    case MP2G_GATE_RANDOM_ACCESS: return g.p0 + 1;
    case MP2G_GATE_U32_INTERLEAVE: case MP2G_GATE_UNINTERLEAVE_TO_B32: case MP2G_GATE_UNINTERLEAVE_TO_U32: return 2;
    default: return 0;
  }
}
// highest wire index + 1 and gate constants a descriptor touches
static void gate_footprint(const mp2g_gate& g, u32& wires, u32& consts) {
  wires = 0; consts = 0;
  switch (g.kind) {
    case MP2G_GATE_CONSTANT: wires = g.p0; consts = g.p0; break;
    case MP2G_GATE_PUBLIC_INPUT: wires = 4; break;
    case MP2G_GATE_ARITHMETIC: wires = 4 * g.p0; consts = 2; break;
    case MP2G_GATE_BASE_SUM: wires = 1 + g.p0; break;
    case MP2G_GATE_ARITHMETIC_EXT: wires = 8 * g.p0; consts = 2; break;
    case MP2G_GATE_MUL_EXT: wires = 6 * g.p0; consts = 1; break;
    case MP2G_GATE_POSEIDON2: case MP2G_GATE_POSEIDON: wires = 135; break;
    case MP2G_GATE_POSEIDON_MDS: wires = 48; break;
    case MP2G_GATE_COSET_INTERPOLATION: wires = 1 + 2 * (1u << g.p0) + 6 + 4 * (((1u << g.p0) - 2) / (g.p1 - 1)); break;
    case MP2G_GATE_U32_ARITHMETIC: wires = 38 * g.p0; break;
    case MP2G_GATE_U32_RANGE_CHECK: wires = 17 * g.p0; break;
    case MP2G_GATE_U32_SUBTRACTION: wires = 21 * g.p0; break;
    case MP2G_GATE_U32_ADD_MANY: wires = (g.p0 + 3 + 18) * g.p1; break;
    case MP2G_GATE_COMPARISON: wires = g.p1 ? 4 + 5 * g.p1 + (g.p0 + g.p1 - 1) / g.p1 + 1 : 0; break;
    case MP2G_GATE_EXPONENTIATION: wires = 2 * g.p0 + 2; break;
    case MP2G_GATE_REDUCING: wires = 6 + g.p0 + 2 * (g.p0 - 1); break;
    case MP2G_GATE_REDUCING_EXT: wires = 6 + 2 * g.p0 + 2 * (g.p0 - 1); break;
    case MP2G_GATE_RANDOM_ACCESS: wires = (2 + (1u << g.p0)) * g.p1 + g.p2 + g.p0 * g.p1; consts = g.p2; break;
    case MP2G_GATE_LOOKUP: wires = 2 * g.p0; break;
    case MP2G_GATE_LOOKUP_TABLE: wires = 3 * g.p0; break;
    case MP2G_GATE_U32_INTERLEAVE: wires = 34 * g.p0; break;
    case MP2G_GATE_UNINTERLEAVE_TO_B32: case MP2G_GATE_UNINTERLEAVE_TO_U32: wires = 67 * g.p0; break;
    default: break;
  }
}
const char* gate_table_check(const GateTable& t, u32 num_constants, u32 wires_w) {
  if (t.n_gates > MP2G_MAX_GATES) return "too many gates";
  if (t.num_selectors == 0 || t.num_selectors > num_constants) return "num_selectors must be in 1..num_constants";
  for (u32 i = 0; i < t.n_gates; i++) {
    const mp2g_gate& g = t.g[i];
    if (g.kind > MP2G_GATE_UNINTERLEAVE_TO_U32) return "unknown gate kind";
    if (g.kind >= MP2G_GATE_LOOKUP && g.p0 < 1) return "gate needs at least one slot / operation";
    if ((g.kind == MP2G_GATE_U32_ARITHMETIC || g.kind == MP2G_GATE_U32_RANGE_CHECK || g.kind == MP2G_GATE_U32_SUBTRACTION) && g.p0 < 1)
      return "u32 gate needs at least one operation";
    if (g.kind == MP2G_GATE_U32_ADD_MANY && (g.p0 < 1 || g.p0 > 16 || g.p1 < 1)) return "U32AddManyGate needs 1..16 addends and an operation";
    if (g.kind == MP2G_GATE_COMPARISON && (g.p1 < 1 || g.p0 < g.p1 || (g.p0 + g.p1 - 1) / g.p1 > 4))
      return "ComparisonGate needs num_chunks >= 1 and chunks of at most 4 bits";
    if (g.kind == MP2G_GATE_COSET_INTERPOLATION && (g.p0 < 2 || g.p0 > 5 || g.p1 < 2 || g.p1 > (1u << g.p0)))
      return "CosetInterpolationGate needs 2..5 subgroup bits and 2 <= degree <= 2^bits";
    if (g.kind == MP2G_GATE_BASE_SUM && (g.p1 < 2 || g.p0 < 1)) return "BaseSumGate needs base >= 2 and a limb";
    if ((g.kind == MP2G_GATE_REDUCING || g.kind == MP2G_GATE_REDUCING_EXT || g.kind == MP2G_GATE_EXPONENTIATION) && g.p0 < 1)
      return "gate needs at least one coefficient / power bit";
    if (g.kind == MP2G_GATE_RANDOM_ACCESS && (g.p0 < 1 || g.p0 > 6 || g.p1 < 1)) return "RandomAccessGate needs 1..6 bits and a copy";
    u32 w, c;
    gate_footprint(g, w, c);
    if (w > wires_w) return "gate needs more wires than the wires oracle has";
    if (t.num_selectors + t.num_lookup_selectors + c > num_constants) return "gate needs more constants than the preprocessed oracle has";
    if (g.selector_index >= t.num_selectors) return "selector_index out of range";
    if (!(g.group_start <= i && i < g.group_end && g.group_end <= t.n_gates)) return "gate is not inside its selector group";
    if (gate_num_constraints(g) > MP2G_MAX_GATE_CONSTRAINTS) return "gate has too many constraints";
  }
  return nullptr;
}

struct Alg { u64 a, b; };  // ExtensionAlgebra element over the evaluation field, X^2 = 7
GLD Alg alg_mul(Alg x, Alg y) {
  return Alg{gl_mul_add(x.a, y.a, gl_mul_small_w(gl_mulw(x.b, y.b), 7)), gl_mul_add(x.a, y.b, gl_mulw(x.b, y.a))};
}
GLD Alg alg_add(Alg x, Alg y) { return Alg{gl_add(x.a, y.a), gl_add(x.b, y.b)}; }
GLD Alg alg_sub(Alg x, Alg y) { return Alg{gl_sub(x.a, y.a), gl_sub(x.b, y.b)}; }
GLD Alg alg_scale(Alg x, u64 c) { return Alg{gl_mul(x.a, c), gl_mul(x.b, c)}; }

// a - b for any u64 a and canonical b, as some u64 representative
GLD u64 gl_subw(u64 a, u64 b) {
  u64 r;
  bool br = __builtin_sub_overflow(a, b, &r);  // borrow: r = a - b + 2^64 >= 2^32, so -EPS cannot borrow again
  return r - (br ? GL_EPS : 0);
}
// sum_j alpha^j C_j: gl_cols (gl.cuh), carry-free columns
typedef gl_cols LazySum;
// eval_unfiltered_base of one gate: wire(j) / cst(j) fetch local wire j / gate constant j (after the
// selector prefix), emit(c) receives the constraints in plonky2's order. WEAK = true lets the Poseidon
// gates hand over un-canonicalised representatives (the lazy accumulator of the LDE kernel takes any u64).
// one copy of RandomAccessGate<BITS>: bit constraints, index reconstruction, the fold of the 2^BITS items by the index bits
// (least significant first), the claimed element; constraint order of gates/random_access.rs eval_unfiltered
template <int BITS, class WireF, class Emit>
GLD void ra_copy(WireF& wire, Emit& emit, u32 w0, u32 b0) {
  constexpr int VS = 1 << BITS;
  u64 bit[BITS], items[VS];
#pragma unroll
  for (int i = 0; i < BITS; i++) bit[i] = wire(b0 + i);
#pragma unroll
  for (int k = 0; k < VS; k++) items[k] = wire(w0 + 2 + k);
  const u64 access = wire(w0), claimed = wire(w0 + 1);
#pragma unroll
  for (int i = 0; i < BITS; i++) emit(gl_mul(bit[i], gl_sub(bit[i], 1)));
  u64 idx = 0;
#pragma unroll
  for (int i = BITS; i-- > 0;) idx = gl_add(gl_add(idx, idx), bit[i]);
  emit(gl_sub(idx, access));
#pragma unroll
  for (int i = 0; i < BITS; i++) {
#pragma unroll
    for (int k = 0; k < (VS >> (i + 1)); k++) items[k] = gl_add(items[2 * k], gl_mul(bit[i], gl_sub(items[2 * k + 1], items[2 * k])));
  }
  emit(gl_sub(items[0], claimed));
}

template <bool WEAK, class WireF, class ConstF, class Emit>
__device__ __forceinline__ void eval_gate(const mp2g_gate g, WireF wire, ConstF cst, const u64* __restrict__ pih, Emit emit) {
  // state limb + round constant - wire
  auto diff = [](u64 s, u64 rc, u64 in) { return WEAK ? gl_subw(gl_addw(s, rc), in) : gl_sub(gl_add(gl_canon(s), rc), in); };
  auto diff0 = [](u64 s, u64 in) { return WEAK ? gl_subw(s, in) : gl_sub(gl_canon(s), in); };
  // product that only feeds further products or emit(): a weak representative is enough there
  auto mulx = [](u64 a, u64 b) { return WEAK ? gl_mulw(a, b) : gl_mul(a, b); };
  // Up to 16 wires in flight at once. The gate bodies are loops with run-time bounds, and a loop that fetches one
  // wire per iteration exposes one global-load latency per constraint (measured: the limb-heavy gates ran at half of
  // their instruction-issue time); fetching a batch first lets the loads overlap.
  auto load16 = [&](u32 first, u32 count, u64(&buf)[16]) {
#pragma unroll
    for (u32 k = 0; k < 16; k++) buf[k] = k < count ? wire(first + k) : 0;
  };
  auto range4 = [&](u64 limb) {  // limb (limb - 1)(limb - 2)(limb - 3) = z (z + 2) with z = limb (limb - 3): two products instead of three
    const u64 z = mulx(limb, gl_sub(limb, 3));
    return mulx(z, WEAK ? gl_addw(z, 2) : gl_add(z, 2));
  };
  // Horner step in base 4 and the final difference: weak representatives where the caller takes them (4 acc as a 66-bit integer reduced
  // without canonicalisation, limb / b canonical wire values)
  auto horner4 = [](u64 acc, u64 limb) {
    if (!WEAK) return gl_add(gl_mul_small(acc, 4), limb);
    return gl_addw(gl_reduce96w(acc << 2, acc >> 62), limb);
  };
  auto subx = [](u64 a, u64 b) { return WEAK ? gl_subw(a, b) : gl_sub(a, b); };
  switch (g.kind) {
    case MP2G_GATE_CONSTANT:
      for (u32 i = 0; i < g.p0; i++) emit(gl_sub(cst(i), wire(i)));
      break;
    case MP2G_GATE_PUBLIC_INPUT:
      for (u32 i = 0; i < 4; i++) emit(gl_sub(wire(i), pih[i]));
      break;
    case MP2G_GATE_ARITHMETIC: {
      const u64 c0 = cst(0), c1 = cst(1);
      u64 lm[16];
      for (u32 i0 = 0; i0 < g.p0; i0 += 4) {  // four operations = sixteen wires in flight
        const u32 cnt = g.p0 - i0 < 4 ? g.p0 - i0 : 4;
        load16(4 * i0, 4 * cnt, lm);
#pragma unroll
        for (u32 k = 0; k < 4; k++)
          if (k < cnt) emit(gl_sub(lm[4 * k + 3], gl_mul_add(gl_mulw(lm[4 * k], lm[4 * k + 1]), c0, gl_mulw(lm[4 * k + 2], c1))));
      }
      break;
    }
    case MP2G_GATE_BASE_SUM: {
      u64 acc = 0, lm[16];
      for (u32 hi = g.p0; hi > 0;) {  // limbs p0-1 .. 0 in batches of 16
        const u32 lo = hi > 16 ? hi - 16 : 0, cnt = hi - lo;
        load16(1 + lo, cnt, lm);
#pragma unroll
        for (int j = 15; j >= 0; j--)
          if ((u32)j < cnt) acc = gl_add(gl_mul_small(acc, g.p1), lm[j]);
        hi = lo;
      }
      emit(gl_sub(acc, wire(0)));
      for (u32 lo = 0; lo < g.p0; lo += 16) {
        const u32 cnt = g.p0 - lo < 16 ? g.p0 - lo : 16;
        load16(1 + lo, cnt, lm);
#pragma unroll
        for (int j = 0; j < 16; j++) {
          if ((u32)j < cnt) {
            u64 pr = lm[j];  // k = 0 factor
            for (u32 k = 1; k < g.p1; k++) pr = mulx(pr, gl_sub(lm[j], k));
            emit(pr);
          }
        }
      }
      break;
    }
    case MP2G_GATE_ARITHMETIC_EXT: {
      const u64 c0 = cst(0), c1 = cst(1);
      u64 lm[16];
      for (u32 i0 = 0; i0 < g.p0; i0 += 2) {  // two operations = sixteen wires in flight
        const u32 cnt = g.p0 - i0 < 2 ? g.p0 - i0 : 2;
        load16(8 * i0, 8 * cnt, lm);
#pragma unroll
        for (u32 k = 0; k < 2; k++) {
          if (k < cnt) {
            const u64* w8 = lm + 8 * k;
            Alg m0{w8[0], w8[1]}, m1{w8[2], w8[3]}, ad{w8[4], w8[5]}, o{w8[6], w8[7]};
            Alg d = alg_sub(o, alg_add(alg_scale(alg_mul(m0, m1), c0), alg_scale(ad, c1)));
            emit(d.a); emit(d.b);
          }
        }
      }
      break;
    }
    case MP2G_GATE_MUL_EXT: {
      const u64 c0 = cst(0);
      u64 lm[16];
      for (u32 i0 = 0; i0 < g.p0; i0 += 2) {  // two operations = twelve wires in flight
        const u32 cnt = g.p0 - i0 < 2 ? g.p0 - i0 : 2;
        load16(6 * i0, 6 * cnt, lm);
#pragma unroll
        for (u32 k = 0; k < 2; k++) {
          if (k < cnt) {
            const u64* w6 = lm + 6 * k;
            Alg m0{w6[0], w6[1]}, m1{w6[2], w6[3]}, o{w6[4], w6[5]};
            Alg d = alg_sub(o, alg_scale(alg_mul(m0, m1), c0));
            emit(d.a); emit(d.b);
          }
        }
      }
      break;
    }
    case MP2G_GATE_POSEIDON2: {
      // wires: input 0..11, output 12..23, swap 24, delta 25..28, S-box inputs 29..64 (full rounds 1..3),
      // 65..86 (partial rounds), 87..134 (second full rounds). The state runs in weak form between the
      // wire substitutions (poseidon.cuh) and is canonicalised where it meets a wire.
      const u64 swap = wire(24);
      emit(gl_mul(swap, gl_sub(swap, 1)));
      u64 s[12];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        u64 lhs = wire(i), rhs = wire(i + 4), delta = wire(25 + i);
        emit(gl_sub(gl_mul(swap, gl_sub(rhs, lhs)), delta));
        s[i] = gl_add(lhs, delta);
        s[i + 4] = gl_sub(rhs, delta);
      }
#pragma unroll
      for (int i = 8; i < 12; i++) s[i] = wire(i);
      p2_external(s);
#pragma unroll 1
      for (int r = 0; r < 4; r++) {
        if (r != 0) {
#pragma unroll
          for (int i = 0; i < 12; i++) {
            u64 in = wire(29 + 12 * (r - 1) + i);
            emit(diff(s[i], c_p2_ext[12 * r + i], in));
            s[i] = p2_sbox(in, 0);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 12; i++) s[i] = p2_sbox(s[i], c_p2_ext[i]);
        }
        p2_external(s);
      }
      u64 nxt = wire(65);
#pragma unroll 1
      for (int r = 0; r < 22; r++) {
        const u64 in = nxt;
        if (r < 21) nxt = wire(65 + r + 1);  // the next round's wire is on its way while this round computes
        emit(diff(s[0], c_p2_int[r], in));
        s[0] = p2_sbox(in, 0);
        p2_internal(s);
      }
#pragma unroll 1
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
          u64 in = wire(87 + 12 * r + i);
          emit(diff(s[i], c_p2_ext[12 * (4 + r) + i], in));
          s[i] = p2_sbox(in, 0);
        }
        p2_external(s);
      }
#pragma unroll
      for (int i = 0; i < 12; i++) emit(diff0(s[i], wire(12 + i)));
      break;
    }
    case MP2G_GATE_POSEIDON: {
      // gates/poseidon.rs: same wire layout as Poseidon2Gate. plonky2 evaluates the partial rounds in the
      // "fast" factorisation; that is an exact linear identity for arbitrary S-box outputs, so the
      // constraint polynomials equal those of the plain round structure used here.
      const u64 swap = wire(24);
      emit(gl_mul(swap, gl_sub(swap, 1)));
      u64 s[12];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        u64 lhs = wire(i), rhs = wire(i + 4), delta = wire(25 + i);
        emit(gl_sub(gl_mul(swap, gl_sub(rhs, lhs)), delta));
        s[i] = gl_add(lhs, delta);
        s[i + 4] = gl_sub(rhs, delta);
      }
#pragma unroll
      for (int i = 8; i < 12; i++) s[i] = wire(i);
#pragma unroll 1
      for (int r = 0; r < 30; r++) {
        if (r >= 4 && r < 26) {
          u64 in = wire(65 + (r - 4));
          emit(diff(s[0], c_p_rc[12 * r], in));
          s[0] = p2_sbox(in, 0);
#pragma unroll
          for (int i = 1; i < 12; i++) s[i] = gl_addw(s[i], c_p_rc[12 * r + i]);
        } else if (r == 0) {
#pragma unroll
          for (int i = 0; i < 12; i++) s[i] = p2_sbox(s[i], c_p_rc[i]);
        } else {
          const int base = r < 4 ? 29 + 12 * (r - 1) : 87 + 12 * (r - 26);
#pragma unroll
          for (int i = 0; i < 12; i++) {
            u64 in = wire(base + i);
            emit(diff(s[i], c_p_rc[12 * r + i], in));
            s[i] = p2_sbox(in, 0);
          }
        }
        poseidon_mds(s);
      }
#pragma unroll
      for (int i = 0; i < 12; i++) emit(diff0(s[i], wire(12 + i)));
      break;
    }
    case MP2G_GATE_POSEIDON_MDS: {
      // 12 extension inputs (wires 2i, 2i+1), outputs at 24 + 2i; constraints limb-major, D components each
      u64 s0[12], s1[12];
#pragma unroll
      for (int i = 0; i < 12; i++) { s0[i] = wire(2 * i); s1[i] = wire(2 * i + 1); }
      poseidon_mds(s0);
      poseidon_mds(s1);
#pragma unroll
      for (int i = 0; i < 12; i++) {
        emit(gl_sub(wire(24 + 2 * i), gl_canon(s0[i])));
        emit(gl_sub(wire(25 + 2 * i), gl_canon(s1[i])));
      }
      break;
    }
    case MP2G_GATE_COSET_INTERPOLATION: {
      // gates/coset_interpolation.rs: barycentric interpolation over the 2^p0-point subgroup in chunks of
      // `degree` (then degree - 1) points; the running (eval, prod) pair is checked against intermediate
      // wires between chunks. For a two-adic subgroup the weights are w_i = x_i / n.
      const u32 npts = 1u << g.p0, deg = g.p1, nint = (npts - 2) / (deg - 1);
      const u32 w_pt = 1 + 2 * npts, w_val = w_pt + 2, w_int = w_val + 2, w_sh = w_int + 4 * nint;
      const Alg pt{wire(w_pt), wire(w_pt + 1)}, sh{wire(w_sh), wire(w_sh + 1)};
      Alg d0 = alg_sub(pt, alg_scale(sh, wire(0)));
      emit(d0.a); emit(d0.b);
      const u64 om = gl_root_of_unity(g.p0), ninv = gl_inv(npts);
      u64 xi = 1;
      Alg ev{0, 0}, pr{1, 0};
      u32 start = 0, end = deg;
      for (u32 c = 0; c <= nint; c++) {
        for (u32 i = start; i < end; i++) {
          Alg val = alg_scale(Alg{wire(1 + 2 * i), wire(2 + 2 * i)}, gl_mul(xi, ninv));
          Alg term{gl_sub(sh.a, xi), sh.b};
          Alg nev = alg_add(alg_mul(ev, term), alg_mul(val, pr));
          pr = alg_mul(pr, term);
          ev = nev;
          xi = gl_mul(xi, om);
        }
        if (c == nint) break;
        Alg iev{wire(w_int + 2 * c), wire(w_int + 2 * c + 1)}, ipr{wire(w_int + 2 * (nint + c)), wire(w_int + 2 * (nint + c) + 1)};
        Alg d1 = alg_sub(iev, ev), d2 = alg_sub(ipr, pr);
        emit(d1.a); emit(d1.b); emit(d2.a); emit(d2.b);
        ev = iev; pr = ipr;
        start = 1 + (deg - 1) * (c + 1);
        end = start + deg - 1 < npts ? start + deg - 1 : npts;
      }
      Alg d3 = alg_sub(Alg{wire(w_val), wire(w_val + 1)}, ev);
      emit(d3.a); emit(d3.b);
      break;
    }
    case MP2G_GATE_U32_ARITHMETIC: {
      // plonky2-u32 arithmetic_u32.rs: per op m0, m1, addend, output_low, output_high, inverse (routed), then 32
      // two-bit limbs of the 64-bit output
      const u32 ops = g.p0;
      for (u32 i = 0; i < ops; i++) {
        const u32 b = 6 * i;
        const u64 computed = gl_mul_add(wire(b), wire(b + 1), wire(b + 2));
        const u64 lo = wire(b + 3), hi = wire(b + 4), inv = wire(b + 5);
        const u64 hi_not_max = gl_sub(gl_mul(inv, gl_sub(0xFFFFFFFFull, hi)), 1);
        emit(gl_mul(hi_not_max, lo));
        emit(gl_sub(gl_mul_add(hi, (u64)1 << 32, lo), computed));
        u64 clo = 0, chi = 0, lm[16];
        load16(6 * ops + 32 * i + 16, 16, lm);  // limbs 31..16: the high half
#pragma unroll
        for (int j = 15; j >= 0; j--) { emit(range4(lm[j])); chi = horner4(chi, lm[j]); }
        load16(6 * ops + 32 * i, 16, lm);       // limbs 15..0
#pragma unroll
        for (int j = 15; j >= 0; j--) { emit(range4(lm[j])); clo = horner4(clo, lm[j]); }
        emit(subx(clo, lo));
        emit(subx(chi, hi));
      }
      break;
    }
    case MP2G_GATE_U32_RANGE_CHECK: {
      const u32 k = g.p0;
      for (u32 i = 0; i < k; i++) {
        u64 lm[16];
        load16(k + 16 * i, 16, lm);
        u64 acc = 0;
#pragma unroll
        for (int j = 15; j >= 0; j--) acc = horner4(acc, lm[j]);
        emit(subx(acc, wire(i)));
#pragma unroll
        for (int j = 0; j < 16; j++) emit(range4(lm[j]));
      }
      break;
    }
    case MP2G_GATE_U32_SUBTRACTION: {
      const u32 ops = g.p0;
      for (u32 i = 0; i < ops; i++) {
        const u32 b = 5 * i;
        const u64 initial = gl_sub(gl_sub(wire(b), wire(b + 1)), wire(b + 2));
        const u64 res = wire(b + 3), bo = wire(b + 4);
        emit(gl_sub(res, gl_add(initial, gl_mul((u64)1 << 32, bo))));
        u64 comb = 0, lm[16];
        load16(5 * ops + 16 * i, 16, lm);
#pragma unroll
        for (int j = 15; j >= 0; j--) { emit(range4(lm[j])); comb = horner4(comb, lm[j]); }
        emit(subx(comb, res));
        emit(gl_mul(bo, gl_sub(1, bo)));
      }
      break;
    }
    case MP2G_GATE_U32_ADD_MANY: {
      const u32 na = g.p0, ops = g.p1, per = na + 3;
      for (u32 i = 0; i < ops; i++) {
        const u32 b = per * i;
        u64 computed = wire(b + na);
        for (u32 j = 0; j < na; j++) computed = gl_add(computed, wire(b + j));
        const u64 res = wire(b + na + 1), co = wire(b + na + 2);
        emit(gl_sub(gl_mul_add(co, (u64)1 << 32, res), computed));
        u64 cres = 0, ccar = 0, lm[16];
        {
          const u64 l17 = wire(per * ops + 18 * i + 17), l16 = wire(per * ops + 18 * i + 16);
          load16(per * ops + 18 * i, 16, lm);
          emit(range4(l17)); ccar = l17;
          emit(range4(l16)); ccar = horner4(ccar, l16);
        }
#pragma unroll
        for (int j = 15; j >= 0; j--) { emit(range4(lm[j])); cres = horner4(cres, lm[j]); }
        emit(subx(cres, res));
        emit(subx(ccar, co));
      }
      break;
    }
    case MP2G_GATE_COMPARISON: {
      // plonky2-u32 comparison.rs: first <= second over p0 bits in p1 chunks
      const u32 nch = g.p1, cb = (g.p0 + nch - 1) / nch, cs = 1u << cb;
      const u32 o_fc = 4, o_sc = 4 + nch, o_ed = 4 + 2 * nch, o_ce = 4 + 3 * nch, o_iv = 4 + 4 * nch, o_bits = 4 + 5 * nch;
      u64 a1 = 0, a2 = 0;
      for (u32 i = nch; i-- > 0;) { a1 = gl_add(gl_mul_small(a1, cs), wire(o_fc + i)); a2 = gl_add(gl_mul_small(a2, cs), wire(o_sc + i)); }
      emit(gl_sub(a1, wire(0)));
      emit(gl_sub(a2, wire(1)));
      u64 msd = 0;
      for (u32 i = 0; i < nch; i++) {
        const u64 f = wire(o_fc + i), sc = wire(o_sc + i), ce = wire(o_ce + i), iv = wire(o_iv + i);
        u64 p1 = f, p2 = sc;
        for (u32 x = 1; x < cs; x++) { p1 = mulx(p1, gl_sub(f, x)); p2 = mulx(p2, gl_sub(sc, x)); }
        emit(p1);
        emit(p2);
        const u64 diff = gl_sub(sc, f);
        emit(gl_sub(gl_mul(diff, wire(o_ed + i)), gl_sub(1, ce)));
        emit(gl_mul(ce, diff));
        emit(gl_sub(iv, gl_mul(ce, msd)));
        msd = gl_add(iv, gl_mul(gl_sub(1, ce), diff));
      }
      const u64 msd_w = wire(3);
      emit(gl_sub(msd_w, msd));
      u64 bc = 0;
      for (u32 i = cb + 1; i-- > 0;) bc = gl_add(gl_add(bc, bc), wire(o_bits + i));
      for (u32 i = 0; i <= cb; i++) { const u64 bit = wire(o_bits + i); emit(gl_mul(bit, gl_sub(1, bit))); }
      emit(gl_sub(gl_add(cs, msd_w), bc));
      emit(gl_sub(wire(2), wire(o_bits + cb)));
      break;
    }
    case MP2G_GATE_EXPONENTIATION: {
      const u32 nb = g.p0;
      const u64 base = wire(0);
      u64 prev_int = 1;
      for (u32 i = 0; i < nb; i++) {
        u64 prev = i == 0 ? 1 : gl_mul(prev_int, prev_int);
        u64 bit = wire(1 + (nb - 1 - i));
        u64 mulby = gl_mul_add(bit, base, gl_sub(1, bit));
        u64 cur = wire(nb + 2 + i);
        emit(gl_sub(gl_mul(prev, mulby), cur));
        prev_int = cur;
      }
      emit(gl_sub(wire(nb + 1), prev_int));
      break;
    }
    case MP2G_GATE_REDUCING:
    case MP2G_GATE_REDUCING_EXT: {
      const u32 n = g.p0;
      const bool ext = g.kind == MP2G_GATE_REDUCING_EXT;
      const u32 start_accs = 6 + (ext ? 2 * n : n);
      Alg alpha{wire(2), wire(3)}, acc{wire(4), wire(5)};
      for (u32 i = 0; i < n; i++) {
        Alg coeff = ext ? Alg{wire(6 + 2 * i), wire(7 + 2 * i)} : Alg{wire(6 + i), 0};
        Alg nxt = i == n - 1 ? Alg{wire(0), wire(1)} : Alg{wire(start_accs + 2 * i), wire(start_accs + 2 * i + 1)};
        Alg d = alg_sub(alg_add(alg_mul(acc, alpha), coeff), nxt);
        emit(d.a); emit(d.b);
        acc = nxt;
      }
      break;
    }
    case MP2G_GATE_RANDOM_ACCESS: {
      const u32 bits = g.p0, copies = g.p1, extra = g.p2, vs = 1u << bits;
      const u32 routed = (2 + vs) * copies + extra;
      for (u32 c = 0; c < copies; c++) {
        const u32 w0 = (2 + vs) * c, b0 = routed + c * bits;
        // the fold of one copy with the vector size a compile-time constant: the item list stays in registers and its 2^bits + bits
        // wire loads are in flight together (with run-time bounds the list lived in scratch memory and the kernel sat parked on it
        // for 82 % of its cycles, tools/dbg/step_pmc.sh)
        switch (bits) {
          case 1: ra_copy<1>(wire, emit, w0, b0); break;
          case 2: ra_copy<2>(wire, emit, w0, b0); break;
          case 3: ra_copy<3>(wire, emit, w0, b0); break;
          case 4: ra_copy<4>(wire, emit, w0, b0); break;
          default: {  // 32 / 64 items: run-time bounds (a register-resident list this long would halve every circuit's occupancy)
            u64 items[32];
            for (u32 i = 0; i < bits; i++) {
              u64 b = wire(b0 + i);
              emit(gl_mul(b, gl_sub(b, 1)));
            }
            u64 idx = 0;
            for (u32 i = bits; i-- > 0;) idx = gl_add(gl_add(idx, idx), wire(b0 + i));
            emit(gl_sub(idx, wire(w0)));
            {
              const u64 b = wire(b0);
              for (u32 k = 0; k < vs / 2; k++) {
                u64 x = wire(w0 + 2 + 2 * k), y = wire(w0 + 3 + 2 * k);
                items[k] = gl_add(x, gl_mul(b, gl_sub(y, x)));
              }
            }
            for (u32 i = 1, len = vs / 2; i < bits; i++, len >>= 1) {
              const u64 b = wire(b0 + i);
              for (u32 k = 0; k < len / 2; k++) items[k] = gl_add(items[2 * k], gl_mul(b, gl_sub(items[2 * k + 1], items[2 * k])));
            }
            emit(gl_sub(items[0], wire(w0 + 1)));
          }
        }
      }
      for (u32 i = 0; i < extra; i++) emit(gl_sub(cst(i), wire((2 + vs) * copies + i)));
      break;
    }
    case MP2G_GATE_U32_INTERLEAVE: {  // x, x_interleaved per op; 32 bit wires after the routed ones, most significant first
      const u32 ops = g.p0;
      u64 bt[16];
      for (u32 i = 0; i < ops; i++) {
        u64 x = 0, xi = 0;
        for (u32 h = 0; h < 2; h++) {
          load16(2 * ops + 32 * i + 16 * h, 16, bt);
#pragma unroll
          for (int j = 0; j < 16; j++) {
            emit(mulx(bt[j], gl_sub(1, bt[j])));
            x = gl_add(gl_add(x, x), bt[j]);
            xi = gl_add(gl_mul_small(xi, 4), bt[j]);
          }
        }
        emit(gl_sub(x, wire(2 * i)));
        emit(gl_sub(xi, wire(2 * i + 1)));
      }
      break;
    }
    case MP2G_GATE_UNINTERLEAVE_TO_B32:
    case MP2G_GATE_UNINTERLEAVE_TO_U32: {  // x_interleaved, x_evens, x_odds per op; 64 bit wires, most significant first
      const u32 ops = g.p0;
      const bool spread = g.kind == MP2G_GATE_UNINTERLEAVE_TO_B32;
      u64 bt[16];
      for (u32 i = 0; i < ops; i++) {
        u64 x = 0, ev = 0, od = 0;
        for (u32 h = 0; h < 4; h++) {
          load16(3 * ops + 64 * i + 16 * h, 16, bt);
#pragma unroll
          for (int j = 0; j < 16; j++) {
            emit(mulx(bt[j], gl_sub(1, bt[j])));
            x = gl_add(gl_add(x, x), bt[j]);
            u64& d = (j & 1) ? ev : od;  // wire 16h + j has weight 2^(63 - 16h - j): odd j = even weight
            d = gl_add(spread ? gl_mul_small(d, 4) : gl_add(d, d), bt[j]);
          }
        }
        emit(gl_sub(x, wire(3 * i)));
        emit(gl_sub(ev, wire(3 * i + 1)));
        emit(gl_sub(od, wire(3 * i + 2)));
      }
      break;
    }
    default: break;  // Noop; LookupGate / LookupTableGate: no constraints of their own
  }
}

// gates/gate.rs compute_filter
template <class ConstAll>
GLD u64 gate_filter(const GateTable& t, u32 gi, ConstAll call) {
  const mp2g_gate& g = t.g[gi];
  const u64 s = call(g.selector_index);
  u64 f = 1;
  for (u32 r = g.group_start; r < g.group_end; r++)
    if (r != gi) f = gl_mul(f, gl_sub(r, s));
  if (t.num_selectors > 1) f = gl_mul(f, gl_sub(0xFFFFFFFFull, s));
  return f;
}

// One launch per gate of the table, specialised on the gate kind (the generic switch needs ~250 VGPRs
// and runs at 2 waves/SIMD; a specialised body keeps only its own live state). The alpha powers of the
// block's proof sit in LDS, and a gate's constraints accumulate un-reduced, sum_j c_j alpha^j as a 136-bit
// integer (lo, hi, top), reduced once: per constraint and challenge one 64x64 multiply and a carry chain
// instead of two modular multiplications. q[b][a][i] is written by the first launch and added to by the rest.
template <u32 KIND>
__global__ void __launch_bounds__(256) gate_constraints_lde_kernel(mp2g_gate g, u32 gi, u32 num_selectors, u32 cst_off, u32 n_cons,
                                                                   const u64* __restrict__ C, const u64* __restrict__ W,
                                                                   u64 w_bstride, u32 lg, const u64* __restrict__ alphas,
                                                                   u64 al_bstride, u32 nc, const u64* __restrict__ pi_hash,
                                                                   u64* __restrict__ q, int first) {
  __shared__ u64 apw[2][MP2G_MAX_GATE_CONSTRAINTS];
  const u64 N = (u64)1 << lg;
  const u32 p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  for (u32 j = threadIdx.x; j < 2 * n_cons; j += 256) {
    const u32 a = j >= n_cons ? 1 : 0, e = j - a * n_cons;
    apw[a][e] = a < nc ? gl_pow(alphas[b * al_bstride + a], e) : 0;
  }
  __syncthreads();
  if (p >= N) return;
  g.kind = KIND;
  const u64* w = W + b * w_bstride + p;
  const u64* c = C + p;
  const u64* pih = pi_hash + 4 * b;
  auto wire = [&](u32 j) { return w[(u64)j << lg]; };
  auto cst = [&](u32 j) { return c[(u64)(cst_off + j) << lg]; };  // after the selectors and the lookup selectors
  // gates/gate.rs compute_filter
  u64 f = 1;
  {
    const u64 s = c[(u64)g.selector_index << lg];
    for (u32 r = g.group_start; r < g.group_end; r++)
      if (r != gi) f = gl_mul(f, gl_sub(r, s));
    if (num_selectors > 1) f = gl_mul(f, gl_sub(0xFFFFFFFFull, s));
  }
  LazySum acc[2];
  u32 j = 0;
  eval_gate<true>(g, wire, cst, pih, [&](u64 v) {
#pragma unroll
    for (u32 a = 0; a < 2; a++) acc[a].add(v, apw[a][j]);
    j++;
  });
  const u32 i = bitrev32(p, lg);
  for (u32 a = 0; a < nc; a++) {
    u64 r = gl_mul(f, acc[a].value());
    u64* dst = q + (((u64)b * nc + a) << lg) + i;
    *dst = first ? r : gl_add(*dst, r);
  }
}

// The light gates of a table in ONE launch: Constant, PublicInput, Arithmetic, BaseSum, ArithmeticExtension and MulExtension
// rows all live on the first 80 wire columns, evaluate a few dozen constraints each and are bound by streaming those wires and the
// accumulator (5-6.6 cycles per VALU instruction against ~3.1 for the compute-bound gates, tools/dbg/step_pmc.sh). Fused, the
// wires and constants of a point are fetched once (the loads of the gates overlap), the alpha powers are shared (same alpha, the
// longest gate's table) and q is read and written once instead of once per gate. Each case pins the kind at compile time so that
// eval_gate's switch folds to that gate's body, as in the per-kind kernels.
#define MP2G_MAX_LIGHT_GATES 12
struct LightGates {
  u32 n;
  mp2g_gate g[MP2G_MAX_LIGHT_GATES];
  u32 gi[MP2G_MAX_LIGHT_GATES];
};
static bool gate_is_light(const mp2g_gate& g) {
  switch (g.kind) {
    case MP2G_GATE_CONSTANT: case MP2G_GATE_PUBLIC_INPUT: case MP2G_GATE_ARITHMETIC: case MP2G_GATE_BASE_SUM:
    case MP2G_GATE_ARITHMETIC_EXT: case MP2G_GATE_MUL_EXT: return true;
    default: return false;
  }
}
__global__ void __launch_bounds__(256) gate_constraints_lde_light_kernel(LightGates lg_, u32 num_selectors, u32 cst_off, u32 max_cons,
                                                                         const u64* __restrict__ C, const u64* __restrict__ W,
                                                                         u64 w_bstride, u32 lg, const u64* __restrict__ alphas,
                                                                         u64 al_bstride, u32 nc, const u64* __restrict__ pi_hash,
                                                                         u64* __restrict__ q, int first) {
  __shared__ u64 apw[2][MP2G_MAX_GATE_CONSTRAINTS];
  const u64 N = (u64)1 << lg;
  const u32 p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  for (u32 j = threadIdx.x; j < 2 * max_cons; j += 256) {
    const u32 a = j >= max_cons ? 1 : 0, e = j - a * max_cons;
    apw[a][e] = a < nc ? gl_pow(alphas[b * al_bstride + a], e) : 0;
  }
  __syncthreads();
  if (p >= N) return;
  const u64* w = W + b * w_bstride + p;
  const u64* c = C + p;
  const u64* pih = pi_hash + 4 * b;
  auto wire = [&](u32 j) { return w[(u64)j << lg]; };
  auto cst = [&](u32 j) { return c[(u64)(cst_off + j) << lg]; };
  u64 total[2] = {0, 0};
  for (u32 k = 0; k < lg_.n; k++) {
    mp2g_gate g = lg_.g[k];
    const u32 gi = lg_.gi[k];
    u64 f = 1;
    {
      const u64 s = c[(u64)g.selector_index << lg];
      for (u32 r = g.group_start; r < g.group_end; r++)
        if (r != gi) f = gl_mul(f, gl_sub(r, s));
      if (num_selectors > 1) f = gl_mul(f, gl_sub(0xFFFFFFFFull, s));
    }
    LazySum acc[2];
    u32 j = 0;
    auto emit = [&](u64 v) {
#pragma unroll
      for (u32 a = 0; a < 2; a++) acc[a].add(v, apw[a][j]);
      j++;
    };
    switch (g.kind) {
#define LIGHT_CASE(K) case K: g.kind = K; eval_gate<true>(g, wire, cst, pih, emit); break;
      LIGHT_CASE(MP2G_GATE_CONSTANT)
      LIGHT_CASE(MP2G_GATE_PUBLIC_INPUT)
      LIGHT_CASE(MP2G_GATE_ARITHMETIC)
      LIGHT_CASE(MP2G_GATE_BASE_SUM)
      LIGHT_CASE(MP2G_GATE_ARITHMETIC_EXT)
      LIGHT_CASE(MP2G_GATE_MUL_EXT)
#undef LIGHT_CASE
      default: break;
    }
    for (u32 a = 0; a < nc; a++) total[a] = gl_add(total[a], gl_mul(f, acc[a].value()));
  }
  const u32 i = bitrev32(p, lg);
  for (u32 a = 0; a < nc; a++) {
    u64* dst = q + (((u64)b * nc + a) << lg) + i;
    *dst = first ? total[a] : gl_add(*dst, total[a]);
  }
}

__global__ void __launch_bounds__(256) gate_constraints_points_kernel(GateTable t, const u64* __restrict__ consts,
                                                                      const u64* __restrict__ wires, u64 npts, u32 max_j,
                                                                      const u64* __restrict__ pih, u64* __restrict__ out) {
  const u64 p = (u64)blockIdx.x * 256 + threadIdx.x;
  if (p >= npts) return;
  auto wire = [&](u32 j) { return wires[(u64)j * npts + p]; };
  auto call = [&](u32 j) { return consts[(u64)j * npts + p]; };
  const u32 ns = t.num_selectors + t.num_lookup_selectors;
  auto cst = [&](u32 j) { return consts[(u64)(ns + j) * npts + p]; };
  for (u32 j = 0; j < max_j; j++) out[(u64)j * npts + p] = 0;
  for (u32 gi = 0; gi < t.n_gates; gi++) {
    if (t.g[gi].kind == MP2G_GATE_NOOP) continue;
    const u64 f = gate_filter(t, gi, call);
    u32 j = 0;
    eval_gate<false>(t.g[gi], wire, cst, pih, [&](u64 v) {
      u64* o = out + (u64)j * npts + p;
      *o = gl_add(*o, gl_mul(f, v));
      j++;
    });
  }
}

// Witness check on the subgroup H (what makes plonky2's prove() panic on an unsatisfied witness): on a row
// of H only the row's own gate has a non-zero filter, so a violated constraint cannot cancel against another
// gate's. flags[b] |= 2 when any gate constraint of proof b is non-zero somewhere.
__global__ void __launch_bounds__(256) gate_check_kernel(GateTable t, const u64* __restrict__ consts, const u64* __restrict__ wires,
                                                         u64 w_bstride, u64 npts, const u64* __restrict__ pi_hash, u32* __restrict__ flags) {
  const u64 p = (u64)blockIdx.x * 256 + threadIdx.x;
  const u32 b = blockIdx.y;
  if (p >= npts) return;
  const u64* w = wires + b * w_bstride;
  auto wire = [&](u32 j) { return w[(u64)j * npts + p]; };
  auto call = [&](u32 j) { return consts[(u64)j * npts + p]; };
  const u32 ns = t.num_selectors + t.num_lookup_selectors;
  auto cst = [&](u32 j) { return consts[(u64)(ns + j) * npts + p]; };
  bool bad = false;
  for (u32 gi = 0; gi < t.n_gates; gi++) {
    if (t.g[gi].kind == MP2G_GATE_NOOP) continue;
    if (gate_filter(t, gi, call) == 0) continue;  // not this row's gate
    eval_gate<false>(t.g[gi], wire, cst, pi_hash + 4 * b, [&](u64 v) { bad |= v != 0; });
  }
  if (bad) atomicOr(&flags[b], 2u);
}
hipError_t gate_check(hipStream_t s, u32 B, const GateTable& t, const u64* consts, const u64* wires, u64 w_bstride, u64 npts,
                      const u64* pi_hash, u32* flags) {
  if (!npts || !B) return hipSuccess;
  hipLaunchKernelGGL(gate_check_kernel, dim3((u32)((npts + 255) / 256), B), dim3(256), 0, s, t, consts, wires, w_bstride, npts, pi_hash,
                     flags);
  return hipGetLastError();
}

hipError_t gate_constraints_lde(hipStream_t s, u32 B, const GateTable& t, const u64* C, const u64* W, u64 w_bstride, u32 lg,
                                const u64* alphas, u64 al_bstride, u32 nc, const u64* pi_hash, u64* q) {
  if (nc < 1 || nc > 2) return hipErrorInvalidValue;
  const u64 N = (u64)1 << lg;
  const dim3 grid((u32)((N + 255) / 256), B), block(256);
  int first = 1;
  // the light gates first, in one launch (MP2G_GATES_UNFUSED=1: one launch per gate, for A/B runs)
  static int unfused = -1;
  if (unfused < 0) { const char* e = getenv("MP2G_GATES_UNFUSED"); unfused = e ? atoi(e) : 0; }
  LightGates lgs{};
  u32 light_cons = 0;
  if (!unfused)
    for (u32 gi = 0; gi < t.n_gates; gi++) {
      const u32 n_cons = gate_num_constraints(t.g[gi]);
      if (n_cons && gate_is_light(t.g[gi]) && lgs.n < MP2G_MAX_LIGHT_GATES) {
        lgs.g[lgs.n] = t.g[gi]; lgs.gi[lgs.n] = gi; lgs.n++;
        if (n_cons > light_cons) light_cons = n_cons;
      }
    }
  if (lgs.n < 2) lgs.n = 0;  // a lone light gate goes the ordinary way
  if (lgs.n) {
    hipLaunchKernelGGL(gate_constraints_lde_light_kernel, grid, block, 0, s, lgs, t.num_selectors, t.num_selectors + t.num_lookup_selectors,
                       light_cons, C, W, w_bstride, lg, alphas, al_bstride, nc, pi_hash, q, first);
    first = 0;
  }
  for (u32 gi = 0; gi < t.n_gates; gi++) {
    const mp2g_gate& g = t.g[gi];
    const u32 n_cons = gate_num_constraints(g);
    if (!n_cons) continue;
    bool fused = false;
    for (u32 k = 0; k < lgs.n; k++) fused |= lgs.gi[k] == gi;
    if (fused) continue;
#define GATE_CASE(K)                                                                                                        \
  case K:                                                                                                                   \
    hipLaunchKernelGGL(gate_constraints_lde_kernel<K>, grid, block, 0, s, g, gi, t.num_selectors,                          \
                       t.num_selectors + t.num_lookup_selectors, n_cons, C, W, w_bstride, lg, alphas, al_bstride, nc, pi_hash, q, \
                       first);                                                                                              \
    break;
    switch (g.kind) {
      GATE_CASE(MP2G_GATE_CONSTANT)
      GATE_CASE(MP2G_GATE_PUBLIC_INPUT)
      GATE_CASE(MP2G_GATE_ARITHMETIC)
      GATE_CASE(MP2G_GATE_BASE_SUM)
      GATE_CASE(MP2G_GATE_ARITHMETIC_EXT)
      GATE_CASE(MP2G_GATE_MUL_EXT)
      GATE_CASE(MP2G_GATE_POSEIDON2)
      GATE_CASE(MP2G_GATE_EXPONENTIATION)
      GATE_CASE(MP2G_GATE_REDUCING)
      GATE_CASE(MP2G_GATE_REDUCING_EXT)
      GATE_CASE(MP2G_GATE_RANDOM_ACCESS)
      GATE_CASE(MP2G_GATE_POSEIDON)
      GATE_CASE(MP2G_GATE_POSEIDON_MDS)
      GATE_CASE(MP2G_GATE_COSET_INTERPOLATION)
      GATE_CASE(MP2G_GATE_U32_ARITHMETIC)
      GATE_CASE(MP2G_GATE_U32_RANGE_CHECK)
      GATE_CASE(MP2G_GATE_U32_SUBTRACTION)
      GATE_CASE(MP2G_GATE_U32_ADD_MANY)
      GATE_CASE(MP2G_GATE_COMPARISON)
      GATE_CASE(MP2G_GATE_U32_INTERLEAVE)
      GATE_CASE(MP2G_GATE_UNINTERLEAVE_TO_B32)
      GATE_CASE(MP2G_GATE_UNINTERLEAVE_TO_U32)
      default: return hipErrorInvalidValue;
    }
#undef GATE_CASE
    first = 0;
  }
  if (first) return hipMemsetAsync(q, 0, (size_t)B * nc * N * sizeof(u64), s);  // a table of Noops only
  return hipGetLastError();
}
hipError_t gate_constraints_points(hipStream_t s, const GateTable& t, const u64* consts, const u64* wires, u64 npts, u32 max_j,
                                   const u64* pi_hash, u64* out) {
  if (!npts) return hipSuccess;
  hipLaunchKernelGGL(gate_constraints_points_kernel, dim3((u32)((npts + 255) / 256)), dim3(256), 0, s, t, consts, wires, npts, max_j,
                     pi_hash, out);
  return hipGetLastError();
}
}  // namespace mp2g
