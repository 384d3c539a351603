// Internal definitions behind the opaque handles of include/mp2g.h.
#pragma once
#include "../../include/mp2g.h"
#include "merkle.h"
#include "ntt.h"

namespace mp2g {
int fail(const char* fmt, ...);  // records mp2g_last_error(), returns 1
int params_check(const mp2g_fri_params* p);  // prover.hip: every bound the layout arithmetic relies on

// owning device buffer of u64 words
struct DevBuf {
  u64* p = nullptr;
  size_t bytes = 0;
  bool borrowed = false;  // p points into memory another DevBuf owns (a context's prover scratch): never freed here
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p && !borrowed) (void)hipFree(p); }
  hipError_t alloc(size_t b) {
    if (p && !borrowed) (void)hipFree(p);
    p = nullptr; borrowed = false;
    bytes = 0;
    hipError_t e = hipMalloc((void**)&p, b ? b : 8);
    if (e != hipSuccess) { p = nullptr; return e; }  // bytes stays 0: the next caller that needs the buffer retries the allocation
    bytes = b;
    return hipSuccess;
  }
  void release() {
    if (p && !borrowed) (void)hipFree(p);
    p = nullptr; bytes = 0; borrowed = false;
  }
  void borrow(u64* q, size_t b) {
    if (p && !borrowed) (void)hipFree(p);
    p = q; bytes = b; borrowed = true;
  }
};
}  // namespace mp2g

struct mp2g_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  mp2g::NttEngine ntt;
  mp2g::DevBuf wit_vals;  // slot tables of the device witness executor (mp2g_witness_program_run_dev), grown on demand
  mp2g::DevBuf wit_rows;  // its row-major staging wire matrices
  // The provers' per-batch working buffers (coefficients, LDE values, Merkle levels of oracles 1.., quotient values, FRI layers):
  // everything a prove() recomputes from its inputs and nobody reads once its kernels have run. All provers of a context run on
  // the context's ONE stream, so they can use the same memory one after the other: the scratch is as large as the largest
  // prover needs, not the sum over the circuits (a table build holds 17 provers per worker and runs one at a time: 238 GB -> 64 GB
  // at 4 x 48 proofs in flight). MP2G_SHARE_SCRATCH=0 gives every prover buffers of its own again (the A/B switch).
  mp2g::DevBuf prover_scratch;
  uint32_t scratch_users = 0;  // provers that have bound buffers into it; freed when the last of them is (mp2g_prover_free)
  bool share_scratch = true;
};
struct mp2g_tree {
  mp2g_ctx* ctx = nullptr;
  int variant = 0;
  uint32_t leaf_len = 0, log_leaves = 0, cap_h = 0;
  mp2g::DevBuf leaves;  // [L][leaf_len]
  mp2g::DevBuf levels;  // level 0 .. cap level, concatenated
};
// PolynomialBatch: coefficients [w][n], LDE values [w][n<<rate] (polynomial-major, index
// bit-reversed = leaf order), Merkle levels
struct mp2g_batch {
  mp2g_ctx* ctx = nullptr;
  int variant = 0;
  uint32_t log_n = 0, w = 0, rate_bits = 0, cap_h = 0;
  mp2g::DevBuf coeffs, values, levels;
};
