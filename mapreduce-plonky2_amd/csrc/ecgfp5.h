// Launchers of the Ecgfp5 kernels (ecgfp5.hip). Points travel between kernels in fractional
// coordinates (X:Z:U:T), 20 u64 words each ("frac").
#pragma once
#include "gl.cuh"

namespace mp2g {
hipError_t ec_map_to_curve(hipStream_t s, int variant, const u64* in, u32 in_len, u32 count, u64* w_out, u64* wei_out, u64* frac_out);
hipError_t ec_decode(hipStream_t s, const u64* w_in, u32 count, u64* frac_out, u32* bad);
hipError_t ec_sum(hipStream_t s, const u64* frac, u32 count, u64* scratch /* 20*1025 words; result in [0,20) */);
hipError_t ec_sum_ranges(hipStream_t s, const u64* frac, const u32* ranges /* [n][2] */, u32 n_ranges, u64* frac_out /* [n][20] */);
hipError_t ec_emit(hipStream_t s, const u64* frac, u32 count, u64* w_out, u64* wei_out);
hipError_t ec_scalar_mul(hipStream_t s, const u64* frac_in, const u32* scalars, u32 count, u64* frac_out);
hipError_t ec_row_digest(hipStream_t s, int variant, const u64* col_ids, u32 n_cols, const u32* values, const u32* unique,
                         u32 n_unique, u32 rows, u64* frac_out);
}  // namespace mp2g
