// Host-side NTT engine: per-size twiddle plans, per-coset scale tables and the launch logic of
// ntt.hip. One engine per mp2g context (one per GPU); not thread-safe.
#pragma once
#include "gl.cuh"
#include <map>
#include <memory>

namespace mp2g {

struct NttPlan {
  u32 log_n = 0, log_n1 = 0, log_n2 = 0;
  u64 *tw_a = nullptr, *tw_b = nullptr, *tw4_lo = nullptr, *tw4_hi = nullptr, *tw4_full = nullptr;
  u64 *tc_a = nullptr, *tc_b = nullptr;  // merged twiddles of the shift-twiddle scheme (pass A / pass B), null where the pass size does not run it
  u64 n_inv = 0;
  ~NttPlan();
};
struct CosetTables {
  u32 log_n = 0, logK = 0;
  u64 shift = 0;
  u64 *lo = nullptr, *hi = nullptr, *full = nullptr;
  ~CosetTables();
};

struct NttEngine {
  hipStream_t stream = nullptr;
  std::map<u32, std::unique_ptr<NttPlan>> plans;
  std::map<u64, std::unique_ptr<CosetTables>> cosets;
  u64* scratch = nullptr;
  size_t scratch_words = 0;
  // bumped whenever a device buffer a launched kernel may reference (scratch, coset tables) is freed or
  // replaced: a hipGraph captured under an older generation must not be replayed
  u64 generation = 0;
#ifdef MP2G_EXPERIMENT_NTT_PRIORITY
  // A/B of round 6 (variant libraries only; tools/dbg/ntt_priority_ab.sh): the transforms of this engine run on a second, HIGH
  // priority stream, forked from and joined to `stream` by events around every run(), so that a worker's 0.5 ms transform is not
  // dispatched behind the other workers' sponge workgroups
  hipStream_t hi_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipError_t run_impl(const u64* in, u64* out, u32 log_n, u32 polys, u32 logK, u64 in_poly_stride, u64 out_poly_stride, bool inverse,
                      const CosetTables* pre, bool bitrev_out);
#endif
  ~NttEngine();

  hipError_t plan(u32 log_n, bool inverse, NttPlan** out);
  // tables for the 2^logK cosets shift * w_{n 2^logK}^j of the size-n subgroup
  hipError_t coset(u32 log_n, u32 logK, u64 shift, CosetTables** out);
  hipError_t ensure_scratch(size_t words);
  // `polys` transforms of size 2^log_n, each evaluated on 2^logK cosets (pre != null) or once.
  // Output of (poly, coset j) lands at out + poly*out_poly_stride + bitrev(j)*n.
  hipError_t run(const u64* in, u64* out, u32 log_n, u32 polys, u32 logK, u64 in_poly_stride,
                 u64 out_poly_stride, bool inverse, const CosetTables* pre, bool bitrev_out);
  // data[b][i] *= first * base^i
  hipError_t scale_powers(u64* data, u32 log_n, u32 batch, u64 base, u64 first);
};

}  // namespace mp2g
