// Poseidon2 / Poseidon sponge hashing of Merkle leaves, tree reduction to the cap, batched
// hash_no_pad and opening gathers for gfx950.
//
// Replaces [dep] plonky2 hash/merkle_tree.rs MerkleTree::new (leaf hash_or_noop + two_to_one
// levels + cap), hash/hashing.rs hash_n_to_m_no_pad, and the off-circuit H::hash_no_pad call
// sites listed in SURVEY 8(a5) (mp2-common/src/poseidon.rs:49-51, mp2-v1/src/values_extraction/
// mod.rs:58,159,197,295,362,431,509,519, mp2-v1/src/indexing/cell.rs:145, row.rs:316).
//
// This is the integer-ALU-bound part of a commitment (about 0.74 k modular multiplies per
// permutation, 17 permutations per 135-limb leaf). One lane owns one leaf; the LDE values stay
// polynomial-major in HBM so that lane i reading limb p of leaf i is a perfectly coalesced
// 8 B/lane stream (no transposed copy of the 8n x w matrix is ever materialised).
#include "merkle.h"
#include <atomic>
#include <cstdlib>
#include "poseidon_wave.cuh"

namespace mp2g {

#ifdef MP2G_EXPERIMENT_LEAF_PREFETCH
#define LEAF_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))  // the prefetched limbs must not cost the fourth wave (161 VGPRs unconstrained)
#else
#define LEAF_KERNEL_ATTR
#endif
template <int V>
__global__ void __launch_bounds__(256) LEAF_KERNEL_ATTR leaf_hash_poly_major_kernel(const u64* __restrict__ values, u32 w, u64 stride, u64 n, u64* __restrict__ digests,
                                                                   u64 in_bstride, u64 out_bstride) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  values += blockIdx.y * in_bstride;
  digests += blockIdx.y * out_bstride;
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  if (w <= 4) {
    for (u32 p = 0; p < w; p++) s[p] = values[p * stride + i];
  } else {
    const u64* v = values + i;
#ifdef LEAF_ONE_COPY
    // ONE copy of the permutation in the kernel's text (46 KB instead of 92: the instruction cache two CUs share holds 64 KB, and
    // inside a proving step other streams' kernels compete for it): the short last chunk takes the same loop body, its loads
    // guarded by wave-uniform compares
#pragma clang loop unroll(disable)
    for (u32 p = 0; p < w; p += 8) {
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (p + k < w) s[k] = v[(u64)(p + k) * stride];
      perm<V>(s);
    }
#elif defined(MP2G_EXPERIMENT_LEAF_PREFETCH)
    // A/B of round 6 (variant libraries only): the next chunk's 8 limbs are requested BEFORE the permutation of the current one, so
    // that their latency passes under ~19 k instructions instead of in front of them (one generation of blocks starts in lockstep:
    // every wave of the chip waits for its loads at the same moments)
    u64 nx[8];
#pragma unroll
    for (int k = 0; k < 8; k++) nx[k] = k < (int)w ? v[(u64)k * stride] : 0;
    u32 p = 0;
    for (; p + 8 <= w; p += 8) {
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = nx[k];
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (p + 8 + k < w) nx[k] = v[(u64)(p + 8 + k) * stride];
      perm<V>(s);
    }
    if (p < w) {
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (p + k < w) s[k] = nx[k];
      perm<V>(s);
    }
#else
    u32 p = 0;
    for (; p + 8 <= w; p += 8) {
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = v[(u64)(p + k) * stride];
      perm<V>(s);
    }
    if (p < w) {
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (p + k < w) s[k] = v[(u64)(p + k) * stride];
      perm<V>(s);
    }
#endif
  }
  ulonglong2* d = reinterpret_cast<ulonglong2*>(digests + 4 * i);
  d[0] = make_ulonglong2(s[0], s[1]);
  d[1] = make_ulonglong2(s[2], s[3]);
}
#ifdef MP2G_EXPERIMENT_LEAF_ILP2  // compiled into variant libraries only (tools/dbg/build_variant.sh ilp2 "-DMP2G_EXPERIMENT_LEAF_ILP2" merkle.hip)
// Two sponges per lane (leaves i and i + n/2: both loads stay coalesced streams): the experiment the round-3 review asked for
// (MP2G_LEAF_ILP2=1; tools/dbg/sponge_ilp2.sh holds the numbers). Poseidon2 only, n even, w > 4.
// the body shared by the two builds below
#define LEAF_ILP2_BODY                                                                                      \
  const u64 half = n >> 1;                                                                                  \
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;                                                       \
  if (i >= half) return;                                                                                    \
  values += blockIdx.y * in_bstride;                                                                        \
  digests += blockIdx.y * out_bstride;                                                                      \
  u64 s[12], t[12];                                                                                         \
  _Pragma("unroll") for (int k = 0; k < 12; k++) s[k] = t[k] = 0;                                           \
  const u64* v = values + i;                                                                                \
  u32 p = 0;                                                                                                \
  for (; p + 8 <= w; p += 8) {                                                                              \
    _Pragma("unroll") for (int k = 0; k < 8; k++) { s[k] = v[(u64)(p + k) * stride]; t[k] = v[(u64)(p + k) * stride + half]; } \
    poseidon2_perm2(s, t);                                                                                  \
  }                                                                                                         \
  if (p < w) {                                                                                              \
    _Pragma("unroll") for (int k = 0; k < 8; k++)                                                           \
      if (p + k < w) { s[k] = v[(u64)(p + k) * stride]; t[k] = v[(u64)(p + k) * stride + half]; }           \
    poseidon2_perm2(s, t);                                                                                  \
  }                                                                                                         \
  ulonglong2* d = reinterpret_cast<ulonglong2*>(digests + 4 * i);                                           \
  d[0] = make_ulonglong2(s[0], s[1]);                                                                       \
  d[1] = make_ulonglong2(s[2], s[3]);                                                                       \
  d = reinterpret_cast<ulonglong2*>(digests + 4 * (i + half));                                              \
  d[0] = make_ulonglong2(t[0], t[1]);                                                                       \
  d[1] = make_ulonglong2(t[2], t[3]);
// as the register allocator likes it (194 VGPRs: two waves per SIMD) ...
__global__ void __launch_bounds__(256) leaf_hash_poly_major_ilp2_kernel(const u64* __restrict__ values, u32 w, u64 stride, u64 n, u64* __restrict__ digests,
                                                                        u64 in_bstride, u64 out_bstride) { LEAF_ILP2_BODY }
// ... and held to the three waves per SIMD of the single-sponge kernel (168 VGPRs, the rest spilled)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
leaf_hash_poly_major_ilp2_w3_kernel(const u64* __restrict__ values, u32 w, u64 stride, u64 n, u64* __restrict__ digests, u64 in_bstride, u64 out_bstride) { LEAF_ILP2_BODY }
#endif
template <int V>
__global__ void __launch_bounds__(256) leaf_hash_row_major_kernel(const u64* __restrict__ leaves, u32 len, u64 n, u64* __restrict__ digests) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64* v = leaves + i * len;
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  if (len <= 4) {
    for (u32 p = 0; p < len; p++) s[p] = v[p];
  } else {
    u32 p = 0;
    for (; p + 8 <= len; p += 8) {
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = v[p + k];
      perm<V>(s);
    }
    if (p < len) {
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (p + k < len) s[k] = v[p + k];
      perm<V>(s);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) digests[4 * i + k] = s[k];
}
template <int V>
__global__ void __launch_bounds__(256) leaf_hash_ext_soa_kernel(const u64* __restrict__ c0, const u64* __restrict__ c1, u32 arity_bits, u64 n, u64* __restrict__ digests,
                                                                u64 in_bstride, u64 out_bstride) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  c0 += blockIdx.y * in_bstride;
  c1 += blockIdx.y * in_bstride;
  digests += blockIdx.y * out_bstride;
  u32 arity = 1u << arity_bits;
  const u64* a = c0 + (i << arity_bits);
  const u64* b = c1 + (i << arity_bits);
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  if (arity == 1) {  // 2 limbs: noop
    s[0] = a[0]; s[1] = b[0];
  } else if (arity == 2) {
    s[0] = a[0]; s[1] = b[0]; s[2] = a[1]; s[3] = b[1];
  } else {
    for (u32 p = 0; p < arity; p += 4) {
#pragma unroll
      for (int k = 0; k < 4; k++) { s[2 * k] = a[p + k]; s[2 * k + 1] = b[p + k]; }
      perm<V>(s);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) digests[4 * i + k] = s[k];
}
template <int V>
__global__ void __launch_bounds__(256) merkle_level_kernel(const u64* __restrict__ in, u64* __restrict__ out, u64 n_out, u64 bstride) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  in += blockIdx.y * bstride;
  out += blockIdx.y * bstride;
  const ulonglong2* src = reinterpret_cast<const ulonglong2*>(in + 8 * i);
  ulonglong2 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
  u64 s[12] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y, 0, 0, 0, 0};
  perm<V>(s);
  ulonglong2* d = reinterpret_cast<ulonglong2*>(out + 4 * i);
  d[0] = make_ulonglong2(s[0], s[1]);
  d[1] = make_ulonglong2(s[2], s[3]);
}
// Lane-cooperative two_to_one for the small upper levels of a tree (poseidon_wave.cuh): one node per
// 16-lane group; a level of a few thousand nodes is bound by permutation latency, not throughput.
__global__ void __launch_bounds__(256) merkle_level_wave_kernel(const u64* __restrict__ in, u64* __restrict__ out, u64 n_out, u64 bstride) {
  const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 iraw = gid >> 4;
  const int l = (int)(gid & 15);
  const bool active = iraw < n_out;
  const u64 i = active ? iraw : n_out - 1;
  in += blockIdx.y * bstride;
  out += blockIdx.y * bstride;
  u64 x = l < 8 ? in[8 * i + l] : 0;
  x = wp2_perm(x, l);
  if (active && l < 4) out[4 * i + l] = x;
}
#ifdef MP2G_EXPERIMENT_MERKLE_FUSED
// A/B of round 6 (variant libraries only; tools/dbg/merkle_fused_ab.sh, profiles/r06/merkle_fused_ab.txt): SEVERAL levels of a tree
// in one launch -- a block owns 2^s consecutive nodes of the input level and the whole subtree above them, s <= 6; every 16-lane
// group hashes one node at a time (wp2_perm), a level's digests go to global memory (Merkle paths are read from the levels array)
// and, through LDS, to the block's next level: one barrier per level instead of one launch per level. MEASURED SLOWER than one launch
// per level: the 14 levels of a 2^18-leaf tree 0.324 ms against 0.302 ms, the 11 levels of a 2^15-leaf tree 0.199 against 0.175 ms,
// a lone 2^12-row proof 4.22 against 4.19-4.23 ms, the table block 919 against 920 proofs/s. A level as a launch of its own costs
// ~16-21 us (the launches of a stream queue behind one another: their overhead overlaps the running kernel), a level inside the
// fused kernel a full lane-cooperative permutation plus the barrier. The product keeps one launch per level.
__global__ void __launch_bounds__(256) merkle_subtree_wave_kernel(const u64* __restrict__ in, u64 n_in, u32 s, u64 bstride) {
  __shared__ u64 buf[2][32 * 4];
  const int l = (int)(threadIdx.x & 15), g = (int)(threadIdx.x >> 4);
  const u64* src_g = in + blockIdx.y * bstride + ((u64)blockIdx.x << s) * 4;   // this block's 2^s input nodes
  u64* lvl = const_cast<u64*>(in) + blockIdx.y * bstride + 4 * n_in;          // level 1 of the tree (n_in / 2 nodes)
  u64 n_lvl = n_in >> 1;
  for (u32 j = 1; j <= s; j++) {
    const u32 m = 1u << (s - j);                                               // nodes of this block at level j
    u64* dst_g = lvl + (u64)blockIdx.x * m * 4;
    for (u32 node = g; node < m; node += 16) {
      u64 x = 0;
      if (l < 8) x = j == 1 ? src_g[8 * node + l] : buf[j & 1][8 * node + l];
      x = wp2_perm(x, l);
      if (l < 4) { dst_g[4 * node + l] = x; buf[(j + 1) & 1][4 * node + l] = x; }
    }
    __syncthreads();
    lvl += 4 * n_lvl;
    n_lvl >>= 1;
  }
}
#endif
template <int V>
__global__ void __launch_bounds__(256) hash_no_pad_batch_kernel(const u64* __restrict__ in, u32 in_len, u64 count, u32 out_len, u64* __restrict__ out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const u64* v = in + i * in_len;
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  for (u32 p = 0; p < in_len; p += 8) {
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (p + k < in_len) s[k] = v[p + k];
    perm<V>(s);
  }
  u64* o = out + i * out_len;
  u32 done = 0;
  for (;;) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (done < out_len) o[done] = s[k];
      done++;
    }
    if (done >= out_len) break;
    perm<V>(s);
  }
}
__global__ void merkle_open_kernel(const u64* __restrict__ levels, u32 log_leaves, u32 cap_h, const u32* __restrict__ idx, u32 n_idx, u64* __restrict__ sib) {
  u32 depth = log_leaves - cap_h;
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_idx * depth * 4) return;
  u32 k = t & 3, l = (t >> 2) % depth, q = (t >> 2) / depth;
  u64 off = 0;
  for (u32 j = 0; j < l; j++) off += (u64)4 << (log_leaves - j);
  u32 node = (idx[q] >> l) ^ 1;
  sib[t] = levels[off + 4 * (u64)node + k];
}
__global__ void gather_rows_kernel(const u64* __restrict__ values, u32 w, u64 stride, const u32* __restrict__ idx, u32 n_idx, u64* __restrict__ out) {
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_idx * w) return;
  u32 p = t % w, q = t / w;
  out[t] = values[(u64)p * stride + idx[q]];
}
// 64x64 LDS tile transpose: in[p][i] -> out[i][p]
__global__ void __launch_bounds__(256) transpose_kernel(const u64* __restrict__ values, u32 w, u64 stride, u64 n, u64* __restrict__ out,
                                                        u64 in_bstride, u64 out_bstride) {
  __shared__ u64 tile[64][65];
  values += blockIdx.z * in_bstride;  // one matrix per z
  out += blockIdx.z * out_bstride;
  u64 i0 = (u64)blockIdx.x * 64;
  u32 p0 = blockIdx.y * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    int pi = e >> 6, ii = e & 63;
    if (p0 + pi < w && i0 + ii < n) tile[pi][ii] = values[(u64)(p0 + pi) * stride + i0 + ii];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    int ii = e >> 6, pi = e & 63;
    if (p0 + pi < w && i0 + ii < n) out[(i0 + ii) * w + p0 + pi] = tile[pi][ii];
  }
}

#define LAUNCH_V(kernel, grid, block, st, ...)                                              \
  do {                                                                                      \
    if (variant == MP2G_POSEIDON2) hipLaunchKernelGGL((kernel<MP2G_POSEIDON2>), grid, block, 0, st, __VA_ARGS__); \
    else if (variant == MP2G_POSEIDON) hipLaunchKernelGGL((kernel<MP2G_POSEIDON>), grid, block, 0, st, __VA_ARGS__); \
    else return hipErrorInvalidValue;                                                       \
  } while (0)
static inline dim3 grid1(u64 n, u32 block, u32 batch = 1) { return dim3((u32)((n + block - 1) / block), batch); }

// permutations queued by leaf_hash_poly_major since the library was loaded (a host-side count: mp2g_stat_leaf_permutations; with the
// kernel's total time from a kernel trace it gives the sponge's rate INSIDE a proving step)
static std::atomic<unsigned long long> g_leaf_perms{0};
unsigned long long leaf_permutations_queued() { return g_leaf_perms.load(std::memory_order_relaxed); }

hipError_t leaf_hash_poly_major(hipStream_t st, int variant, const u64* values, u32 w, u64 stride, u64 n, u64* digests,
                                u32 batch, u64 in_bstride, u64 out_bstride) {
  if (!n || !batch) return hipSuccess;
  if (w > 4) g_leaf_perms.fetch_add((unsigned long long)n * batch * ((w + 7) / 8), std::memory_order_relaxed);
#ifdef MP2G_EXPERIMENT_LEAF_ILP2
  // A/B switch of the two-sponges-per-lane kernels: exists in variant libraries only, the product has one leaf kernel and no switch
  static const int ilp2 = [] { const char* e = getenv("MP2G_LEAF_ILP2"); return e ? atoi(e) : 0; }();
  if (ilp2 && variant == MP2G_POSEIDON2 && w > 4 && (n & 1) == 0) {
    if (ilp2 == 2) hipLaunchKernelGGL(leaf_hash_poly_major_ilp2_w3_kernel, grid1(n >> 1, 256, batch), dim3(256), 0, st, values, w, stride, n, digests, in_bstride, out_bstride);
    else hipLaunchKernelGGL(leaf_hash_poly_major_ilp2_kernel, grid1(n >> 1, 256, batch), dim3(256), 0, st, values, w, stride, n, digests, in_bstride, out_bstride);
    return hipGetLastError();
  }
#endif
  LAUNCH_V(leaf_hash_poly_major_kernel, grid1(n, 256, batch), dim3(256), st, values, w, stride, n, digests, in_bstride, out_bstride);
  return hipGetLastError();
}
hipError_t leaf_hash_row_major(hipStream_t st, int variant, const u64* leaves, u32 len, u64 n, u64* digests) {
  if (!n) return hipSuccess;
  LAUNCH_V(leaf_hash_row_major_kernel, grid1(n, 256), dim3(256), st, leaves, len, n, digests);
  return hipGetLastError();
}
hipError_t leaf_hash_ext_soa(hipStream_t st, int variant, const u64* c0, const u64* c1, u32 arity_bits, u64 n, u64* digests,
                             u32 batch, u64 in_bstride, u64 out_bstride) {
  if (!n || !batch) return hipSuccess;
  LAUNCH_V(leaf_hash_ext_soa_kernel, grid1(n, 256, batch), dim3(256), st, c0, c1, arity_bits, n, digests, in_bstride, out_bstride);
  return hipGetLastError();
}
hipError_t merkle_reduce(hipStream_t st, int variant, u64* levels, u32 log_leaves, u32 cap_h, u32 batch, u64 bstride) {
  u64* cur = levels;
  if (!batch) return hipSuccess;
  for (u32 lv = log_leaves; lv > cap_h; lv--) {
    u64 n_in = (u64)1 << lv;
    u64* nxt = cur + 4 * n_in;
    // below ~2^14 nodes in flight the level is latency-bound: spread each permutation over 16 lanes
    if (variant == MP2G_POSEIDON2 && (n_in / 2) * (u64)batch <= 16384) {
#ifdef MP2G_EXPERIMENT_MERKLE_FUSED  // the remaining levels in as few launches as 6 levels a block allow (measured slower: see the kernel)
      const u32 left = lv - cap_h, launches = (left + 5) / 6, sl = (left + launches - 1) / launches;
      if (sl >= 2) {
        hipLaunchKernelGGL(merkle_subtree_wave_kernel, dim3((u32)(n_in >> sl), batch), dim3(256), 0, st, cur, n_in, sl, bstride);
        for (u32 j = 0; j < sl; j++) { cur += 4 * (n_in >> j); }
        lv -= sl - 1;  // the loop's own step takes the last of the sl levels
        continue;
      }
#endif
      hipLaunchKernelGGL(merkle_level_wave_kernel, grid1(n_in / 2 * 16, 256, batch), dim3(256), 0, st, cur, nxt, n_in / 2, bstride);
    } else
      LAUNCH_V(merkle_level_kernel, grid1(n_in / 2, 256, batch), dim3(256), st, cur, nxt, n_in / 2, bstride);
    cur = nxt;
  }
  return hipGetLastError();
}
hipError_t hash_no_pad_batch(hipStream_t st, int variant, const u64* in, u32 in_len, u64 count, u32 out_len, u64* out) {
  if (!count) return hipSuccess;
  LAUNCH_V(hash_no_pad_batch_kernel, grid1(count, 256), dim3(256), st, in, in_len, count, out_len, out);
  return hipGetLastError();
}
hipError_t merkle_open(hipStream_t st, const u64* levels, u32 log_leaves, u32 cap_h, const u32* idx, u32 n_idx, u64* siblings) {
  u32 total = n_idx * (log_leaves - cap_h) * 4;
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(merkle_open_kernel, grid1(total, 256), dim3(256), 0, st, levels, log_leaves, cap_h, idx, n_idx, siblings);
  return hipGetLastError();
}
hipError_t gather_rows(hipStream_t st, const u64* values, u32 w, u64 stride, const u32* idx, u32 n_idx, u64* out) {
  u32 total = n_idx * w;
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(gather_rows_kernel, grid1(total, 256), dim3(256), 0, st, values, w, stride, idx, n_idx, out);
  return hipGetLastError();
}
hipError_t transpose_to_leaves(hipStream_t st, const u64* values, u32 w, u64 stride, u64 n, u64* out, u32 batch, u64 in_bstride,
                               u64 out_bstride) {
  if (!n || !w || !batch) return hipSuccess;
  hipLaunchKernelGGL(transpose_kernel, dim3((u32)((n + 63) / 64), (w + 63) / 64, batch), dim3(256), 0, st, values, w, stride, n, out,
                     in_bstride, out_bstride);
  return hipGetLastError();
}
}  // namespace mp2g
