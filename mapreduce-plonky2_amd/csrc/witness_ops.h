// The generators of the leaf-circuit gates as witness-tape instructions (include/mp2g.h MP2G_OP_U32_ARITH .. MP2G_OP_EXP): ONE
// definition for the host replay (witness.hip) and the device replay (witness_dev.hip). Restates the generators registered at
// mp2-common/src/serialization/circuit_data_serialization.rs:186-231 -- [dep] plonky2-u32 gates/{arithmetic_u32, subtraction_u32,
// add_many_u32, range_check_u32, comparison}.rs, plonky2 gates/{base_sum, multiplication_extension, exponentiation}.rs -- with the
// wire layouts of the gate evaluators (gates.hip / oracle/gates_body.inc). t = the operands after the opcode, vals = the proof's
// slot table, put(col, row, value) writes a wire.
#pragma once
#include "gl.cuh"
#include "witness.h"

namespace mp2g {
template <class Put>
GLHD bool exec_gate_op(u64 op, const u64* t, u64* vals, Put put) {
  switch (op) {
    case OP_U32_ARITH: {
      const u64 row = t[0], i = t[1], ops = t[2];
      const u64 m0 = vals[t[3]], m1 = vals[t[4]], ad = vals[t[5]];
      const u64 out = gl_add(gl_mul(m0, m1), ad);  // u32 operands: below p, the integer m0 m1 + addend itself
      const u64 lo = out & 0xFFFFFFFFull, hi = out >> 32;
      const u64 b = 6 * i;
      put(b, row, m0); put(b + 1, row, m1); put(b + 2, row, ad); put(b + 3, row, lo); put(b + 4, row, hi);
      put(b + 5, row, gl_inv(0xFFFFFFFFull - hi));  // (u32::MAX - high)^-1, 0 when high = u32::MAX (then low must be 0)
      for (u32 j = 0; j < 32; j++) put(6 * ops + 32 * i + j, row, (out >> (2 * j)) & 3);
      vals[t[6]] = lo; vals[t[7]] = hi;
      return true;
    }
    case OP_U32_SUB: {
      const u64 row = t[0], i = t[1], ops = t[2];
      const u64 x = vals[t[3]], y = vals[t[4]], bi = vals[t[5]];
      const u64 r0 = gl_sub(gl_sub(x, y), bi);          // negative differences sit just below p
      const u64 bo = r0 > ((u64)1 << 32) ? 1 : 0;
      const u64 r = gl_add(r0, bo << 32);
      const u64 b = 5 * i;
      put(b, row, x); put(b + 1, row, y); put(b + 2, row, bi); put(b + 3, row, r); put(b + 4, row, bo);
      for (u32 j = 0; j < 16; j++) put(5 * ops + 16 * i + j, row, (r >> (2 * j)) & 3);
      vals[t[6]] = r; vals[t[7]] = bo;
      return true;
    }
    case OP_U32_ADD_MANY: {
      const u64 row = t[0], i = t[1], ops = t[2], na = t[3];
      const u64 per = na + 3, b = per * i;
      u64 tot = 0;
      for (u32 k = 0; k < na; k++) { const u64 a = vals[t[4 + k]]; put(b + k, row, a); tot = gl_add(tot, a); }
      const u64 ci = vals[t[4 + na]];
      tot = gl_add(tot, ci);
      const u64 res = tot & 0xFFFFFFFFull, co = tot >> 32;
      put(b + na, row, ci); put(b + na + 1, row, res); put(b + na + 2, row, co);
      for (u32 j = 0; j < 16; j++) put(per * ops + 18 * i + j, row, (res >> (2 * j)) & 3);
      for (u32 j = 0; j < 2; j++) put(per * ops + 18 * i + 16 + j, row, (co >> (2 * j)) & 3);
      vals[t[5 + na]] = res; vals[t[6 + na]] = co;
      return true;
    }
    case OP_U32_RANGE_CHECK: {
      const u64 row = t[0], i = t[1], k = t[2], x = vals[t[3]];
      put(i, row, x);
      for (u32 j = 0; j < 16; j++) put(k + 16 * i + j, row, (x >> (2 * j)) & 3);
      return true;
    }
    case OP_COMPARISON: {
      const u64 row = t[0];
      const u32 nb = (u32)t[1], nch = (u32)t[2], cb = (nb + nch - 1) / nch;
      const u64 a = vals[t[3]], b = vals[t[4]], cmask = ((u64)1 << cb) - 1;
      put(0, row, a); put(1, row, b);
      // the chunks' differences first, then ALL their inverses from one field inversion (Montgomery's trick: prefix products over the
      // non-zero differences, one gl_inv, back-substitution): a lane replays the whole row, and 16 inversions of 64 squarings each
      // one after the other would make this row the longest instruction of its level
      u64 diff[16], pre[16], msd = 0, acc = 1;
      for (u32 i = 0; i < nch; i++) {
        const u64 fc = cb * i < 64 ? (a >> (cb * i)) & cmask : 0, sc = cb * i < 64 ? (b >> (cb * i)) & cmask : 0;
        put(4 + i, row, fc); put(4 + nch + i, row, sc);
        diff[i] = gl_sub(sc, fc);
        pre[i] = acc;                                   // product of the non-zero differences before chunk i
        if (diff[i]) acc = gl_mul(acc, diff[i]);
      }
      u64 inv_all = gl_inv(acc);                        // acc != 0: a product of non-zero field elements (1 when every chunk is equal)
      for (u32 i = nch; i-- > 0;) {
        if (!diff[i]) { pre[i] = 1; continue; }         // equality dummy of equal chunks: 1
        const u64 inv_i = gl_mul(inv_all, pre[i]);      // 1 / diff[i]
        inv_all = gl_mul(inv_all, diff[i]);
        pre[i] = inv_i;
      }
      for (u32 i = 0; i < nch; i++) {
        const u64 eq = diff[i] == 0 ? 1 : 0;
        put(4 + 2 * nch + i, row, pre[i]);              // equality dummy: 1 / (second - first), 1 for equal chunks
        put(4 + 3 * nch + i, row, eq);
        const u64 iv = eq ? msd : 0;
        put(4 + 4 * nch + i, row, iv);
        msd = eq ? iv : diff[i];                        // intermediate + (1 - equal) diff
      }
      put(3, row, msd);
      const u64 val = gl_add((u64)1 << cb, msd);  // 2^chunk_bits + most significant difference, in [1, 2^(chunk_bits + 1))
      for (u32 i = 0; i <= cb; i++) put(4 + 5 * nch + i, row, (val >> i) & 1);
      const u64 res = (val >> cb) & 1;
      put(2, row, res);
      vals[t[5]] = res;
      return true;
    }
    case OP_BASE_SPLIT: {
      const u64 row = t[0], bb = t[1], n = t[2], x = vals[t[3]], mask = ((u64)1 << bb) - 1;
      put(0, row, x);
      for (u32 j = 0; j < n; j++) { const u64 l = (x >> (bb * j)) & mask; put(1 + j, row, l); vals[t[4 + j]] = l; }
      return true;
    }
    case OP_MUL_EXT: {
      const u64 row = t[0], i = t[1], c0 = t[2];
      const gl2 m0 = gl2_make(vals[t[3]], vals[t[4]]), m1 = gl2_make(vals[t[5]], vals[t[6]]);
      const gl2 o = gl2_scale(gl2_mul(m0, m1), c0);
      const u64 b = 6 * i;
      put(b, row, m0.a); put(b + 1, row, m0.b); put(b + 2, row, m1.a); put(b + 3, row, m1.b); put(b + 4, row, o.a); put(b + 5, row, o.b);
      vals[t[7]] = o.a; vals[t[8]] = o.b;
      return true;
    }
    case OP_EXP: {
      const u64 row = t[0];
      const u32 nb = (u32)t[1];
      const u64 base = vals[t[2]];
      put(0, row, base);
      for (u32 j = 0; j < nb; j++) put(1 + j, row, vals[t[3 + j]]);
      u64 cur = 1;
      for (u32 i = 0; i < nb; i++) {  // most significant bit first: square, then multiply by base where the bit is set
        const u64 prev = i == 0 ? 1 : gl_mul(cur, cur);
        cur = vals[t[3 + nb - 1 - i]] ? gl_mul(prev, base) : prev;
        put(nb + 2 + i, row, cur);
      }
      put(nb + 1, row, cur);
      vals[t[3 + nb]] = cur;
      return true;
    }
    default: return false;
  }
}
}  // namespace mp2g
