// ALU micro-benchmarks for gfx950: what a Goldilocks multiply is made of.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../mapreduce-plonky2_amd/csrc ubench.hip -o ubench
#include "poseidon.cuh"
#include <cstdio>
#include <vector>

#define ITERS 2048
// 64-bit formulation of the weak reduction (compiler picks v_lshl_add_u64 + v_cmp_*_u64)
GLD u64 red64(u64 lo, u64 hi) {
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0;
  bool b = __builtin_sub_overflow(lo, hi_hi, &t0);
  t0 -= b ? GL_EPS : 0;
  u64 t1 = (hi_lo << 32) - hi_lo;
  u64 r;
  bool c = __builtin_add_overflow(t0, t1, &r);
  return r + (c ? GL_EPS : 0);
}
GLD u64 mulw64(u64 a, u64 b) { u64 lo, hi; gl_mul_wide(a, b, lo, hi); return red64(lo, hi); }
GLD u64 add64c(u64 a, u64 b) { u32 c0, c1; u32 s0 = __builtin_addc((u32)a, (u32)b, 0u, &c0); u32 s1 = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c0, &c1); return gl_mk(s0, s1 + c1); }

template <int OP>
__global__ void __launch_bounds__(256) k(u64* out, u64 seed) {
  u64 a[8];
  for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1 + i * 977) + blockIdx.x;
  u32 lo32[8];
  for (int i = 0; i < 8; i++) lo32[i] = (u32)a[i];
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = gl_mul(a[i], a[(i + 1) & 7]);
      if (OP == 1) a[i] = (u64)(u32)a[i] * (u32)(a[(i + 1) & 7]) + a[i];            // v_mad_u64_u32
      if (OP == 2) a[i] = a[i] + a[(i + 1) & 7];                                     // 64-bit add
      if (OP == 3) a[i] = a[i] < a[(i + 1) & 7] ? a[i] + 0xFFFFFFFFull : a[i];        // cmp + select + add
      if (OP == 4) lo32[i] = lo32[i] * lo32[(i + 1) & 7] + 1u;                        // v_mul_lo_u32
      if (OP == 5) lo32[i] = __umulhi(lo32[i], lo32[(i + 1) & 7]) + lo32[i];          // v_mul_hi_u32
      if (OP == 6) a[i] = gl_add(a[i], a[(i + 1) & 7]);
      if (OP == 7) lo32[i] = lo32[i] + lo32[(i + 1) & 7];                              // 32-bit add
      if (OP == 8) a[i] = gl_mul_small(a[i], 7);
      if (OP == 9) a[i] = add64c(a[i], a[(i + 1) & 7]);
      if (OP == 10) lo32[i] = lo32[i] < lo32[(i + 1) & 7] ? lo32[i] + 77u : lo32[(i + 1) & 7];
      if (OP == 11) a[i] = mulw64(a[i], a[(i + 1) & 7]);
      if (OP == 12) a[i] = gl_mulw(a[i], a[(i + 1) & 7]);
      if (OP == 13) { u64 l, h; gl_mul_wide(a[i], a[(i + 1) & 7], l, h); a[i] = l ^ h; }
      if (OP == 14) a[i] = red64(a[i], a[(i + 1) & 7]);
      if (OP == 15) a[i] = gl_reduce128w(a[i], a[(i + 1) & 7]);
      if (OP == 16) a[i] = gl_sub(a[i], a[(i + 1) & 7]);
      if (OP == 17) a[i] = gl_addw(a[i], a[(i + 1) & 7]);
    }
  }
  u64 r = 0;
  for (int i = 0; i < 8; i++) r ^= a[i] ^ lo32[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) kperm(u64* out, u64 seed, int reps) {
  u64 s[12];
  for (int i = 0; i < 12; i++) s[i] = (seed * (threadIdx.x + 1 + i * 977) + blockIdx.x) % GL_P;
  for (int r = 0; r < reps; r++) poseidon2_perm(s);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0];
}
template <class F>
float timeit(F f) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  const int blocks = 256 * 16, threads = 256;
  u64* d;
  hipMalloc(&d, sizeof(u64) * blocks * threads);
  const char* names[] = {"gl_mul", "v_mad_u64_u32", "add64", "cmp+sel+add64", "mul_lo_u32", "mul_hi_u32", "gl_add", "add32", "gl_mul_small", "add_co+addc", "cmp32+sel", "mulw(64bit)", "mulw(32chain)", "mul_wide", "red(64bit)", "red(32chain)", "gl_sub", "gl_addw"};
  double ops = (double)blocks * threads * ITERS * 8;
#define RUN(N) { float ms = timeit([&] { hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(threads), 0, 0, d, 0x9E3779B97F4A7C15ull); }); \
    printf("%-14s %8.3f ms  %8.2f Gop/s (lane-ops)\n", names[N], ms, ops / ms / 1e6); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17)
  {
    int reps = 64;
    float ms = timeit([&] { hipLaunchKernelGGL(kperm, dim3(blocks), dim3(threads), 0, 0, d, 0x9E3779B97F4A7C15ull, reps); });
    printf("poseidon2_perm %8.3f ms  %8.3f Gperm/s\n", ms, (double)blocks * threads * reps / ms / 1e6);
  }
  return 0;
}
