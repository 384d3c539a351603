// Shared by witness.hip (host executor, program validation, level schedule) and witness_dev.hip (device executor).
#pragma once
#include <cstdint>
#include <mutex>
#include <vector>
#include "ctx.h"

namespace mp2g {
// the opcodes are the PUBLIC ones (include/mp2g.h enum mp2g_witness_op: operand layouts and rules are documented there)
enum { OP_ARITH = MP2G_OP_ARITH, OP_ARITH_EXT = MP2G_OP_ARITH_EXT, OP_P2 = MP2G_OP_P2, OP_BASE_SUM = MP2G_OP_BASE_SUM, OP_RA = MP2G_OP_RA,
       OP_REDUCING = MP2G_OP_REDUCING, OP_REDUCING_EXT = MP2G_OP_REDUCING_EXT, OP_COSET = MP2G_OP_COSET, OP_WIRE = MP2G_OP_WIRE,
       OP_HINT_DIV_EXT = MP2G_OP_HINT_DIV_EXT, OP_HINT_LO63 = MP2G_OP_HINT_LO63, OP_HINT_HI = MP2G_OP_HINT_HI, OP_HINT_SPLIT = MP2G_OP_HINT_SPLIT,
       OP_PAR = MP2G_OP_PAR, OP_POSEIDON = MP2G_OP_POSEIDON, OP_U32_ARITH = MP2G_OP_U32_ARITH, OP_U32_SUB = MP2G_OP_U32_SUB,
       OP_U32_ADD_MANY = MP2G_OP_U32_ADD_MANY, OP_U32_RANGE_CHECK = MP2G_OP_U32_RANGE_CHECK, OP_COMPARISON = MP2G_OP_COMPARISON,
       OP_BASE_SPLIT = MP2G_OP_BASE_SPLIT, OP_MUL_EXT = MP2G_OP_MUL_EXT, OP_EXP = MP2G_OP_EXP, OP_END = MP2G_OP_END };
const u32 BASE_SUM_LIMBS = 63, RA_BITS = 4, RA_COPIES = 4, RED_COEFFS = 43, RED_EXT_COEFFS = 32, NUM_WIRES = 135;

// the program's read-only data on one device (uploaded at the first device run there)
struct WitnessDev {
  int device = -1;
  DevBuf tape, sched, level_off, level_p2, input_sids, consts, domtab, probe;
};
// device executor (witness_dev.hip): one block per proof walks the level schedule
hipError_t witness_exec_launch(hipStream_t s, const WitnessDev& d, u32 n_levels, u32 n_slots, u32 log_n, u32 n_inputs, u32 n_consts,
                               u32 n_probe, const u64* d_inputs, u32 batch, u64* d_vals, u64* d_wires, u64* d_probe_out);
}  // namespace mp2g

struct mp2g_witness_program {
  std::vector<u64> tape;
  std::vector<u32> input_sids;
  std::vector<u64> consts;  // (sid, value) pairs
  u32 n_slots = 0, log_n = 0;
  u64 dom[6][32], bw[6][32];  // two-adic subgroup of 2^bits points and its barycentric weights, bits <= 5
  // level schedule for the device executor: instruction offsets ordered by (dependency level, opcode); a level's instructions
  // read only slots written at lower levels (every slot is written once: the builder's programs are in SSA form)
  std::vector<u32> sched, level_off;
  std::vector<u32> level_p2;  // per level: first schedule index and count of its Poseidon2 rows (one opcode = one contiguous run)
  bool ssa = true;
  std::vector<u32> probe;  // slots returned next to the wires by the device run (mp2g_witness_program_set_probe)
  std::mutex dev_mu;
  std::vector<mp2g::WitnessDev*> dev;  // per device
  ~mp2g_witness_program() { for (auto* d : dev) delete d; }
};
