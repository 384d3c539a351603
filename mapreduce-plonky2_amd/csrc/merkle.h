// Host-side launchers for the sponge / Merkle kernels of merkle.hip.
#pragma once
#include "gl.cuh"

namespace mp2g {
// number of u64 words of the concatenated levels: level 0 (2^log_leaves digests) ... cap level
inline size_t merkle_levels_words(u32 log_leaves, u32 cap_h) {
  return 4 * ((((size_t)1 << log_leaves) << 1) - ((size_t)1 << cap_h));
}
// digests[i] = hash_or_noop(values[0..w)[i]); values poly-major with `stride` words between polys
// `batch` independent trees: tree b reads values + b*in_bstride and writes digests + b*out_bstride
hipError_t leaf_hash_poly_major(hipStream_t st, int variant, const u64* values, u32 w, u64 stride, u64 n_leaves, u64* digests,
                                u32 batch = 1, u64 in_bstride = 0, u64 out_bstride = 0);
// digests[i] = hash_or_noop(leaves[i][0..len))
hipError_t leaf_hash_row_major(hipStream_t st, int variant, const u64* leaves, u32 len, u64 n_leaves, u64* digests);
// FRI layer leaves: leaf i = 2^arity_bits extension values [c0,c1] interleaved, read from SoA
hipError_t leaf_hash_ext_soa(hipStream_t st, int variant, const u64* c0, const u64* c1, u32 arity_bits, u64 n_leaves, u64* digests,
                             u32 batch = 1, u64 in_bstride = 0, u64 out_bstride = 0);
// levels[0] already holds the leaf digests; fills levels 1..(log_leaves-cap_h)
hipError_t merkle_reduce(hipStream_t st, int variant, u64* levels, u32 log_leaves, u32 cap_h, u32 batch = 1, u64 bstride = 0);
// out[i][0..out_len) = hash_n_to_m_no_pad(in[i][0..in_len))
hipError_t hash_no_pad_batch(hipStream_t st, int variant, const u64* in, u32 in_len, u64 count, u32 out_len, u64* out);
// siblings[q][l][4] bottom-up for idx[q]
hipError_t merkle_open(hipStream_t st, const u64* levels, u32 log_leaves, u32 cap_h, const u32* idx, u32 n_idx, u64* siblings);
// out[q][p] = values[p*stride + idx[q]]
hipError_t gather_rows(hipStream_t st, const u64* values, u32 w, u64 stride, const u32* idx, u32 n_idx, u64* out);
// out[i][p] = values[p*stride + i]  (leaf-major copy for hosts that want plonky2's layout)
// in[p][i] (p < w, rows `stride` apart) -> out[i][p], for `batch` matrices in_bstride / out_bstride words apart
hipError_t transpose_to_leaves(hipStream_t st, const u64* values, u32 w, u64 stride, u64 n_leaves, u64* out, u32 batch = 1, u64 in_bstride = 0,
                               u64 out_bstride = 0);
// permutations queued by leaf_hash_poly_major since the library was loaded (host-side count)
unsigned long long leaf_permutations_queued();
}  // namespace mp2g
