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

// asm formulation: the multiply-add's own carry-out and the subtract chain's borrow drive the two EPS
// corrections (no 64-bit compares)
GLD u64 red_asm(u64 lo, u64 hi) {
  u32 hl = (u32)hi, hh = (u32)(hi >> 32);
  u64 t, c;
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=s"(c) : "v"(hl), "v"(lo));
  u32 t0 = (u32)t, t1 = (u32)(t >> 32), r0, r1, mb, mc;
  asm("v_sub_co_u32 %0, vcc, %4, %6\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "v_cndmask_b32 %2, 0, -1, vcc\n\t"
      "v_cndmask_b32 %3, 0, -1, %7\n\t"
      "v_add_co_u32 %0, vcc, %0, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_sub_co_u32 %0, vcc, %0, %2\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(mb), "=&v"(mc)
      : "v"(t0), "v"(t1), "v"(hh), "s"(c)
      : "vcc");
  return gl_mk(r0, r1);
}
GLD u64 mulw_asm(u64 a, u64 b) { u64 lo, hi; gl_mul_wide(a, b, lo, hi); return red_asm(lo, hi); }
// merged corrections: r += (c - b) * EPS as one signed 64-bit addend
GLD u64 red_asm2(u64 lo, u64 hi) {
  u32 hl = (u32)hi, hh = (u32)(hi >> 32);
  u64 t, c;
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=s"(c) : "v"(hl), "v"(lo));
  u32 t0 = (u32)t, t1 = (u32)(t >> 32), r0, r1, mb, mc;
  asm("v_sub_co_u32 %0, vcc, %4, %6\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "v_cndmask_b32 %2, 0, -1, vcc\n\t"
      "v_cndmask_b32 %3, 0, -1, %7\n\t"
      "v_sub_co_u32 %3, vcc, %3, %2\n\t"          // d0 = mc - mb, borrow when c = 0, b = 1
      "v_subb_co_u32 %2, vcc, 0, 0, vcc\n\t"      // d1 = -borrow
      "v_add_co_u32 %0, vcc, %0, %3\n\t"
      "v_addc_co_u32 %1, vcc, %1, %2, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(mb), "=&v"(mc)
      : "v"(t0), "v"(t1), "v"(hh), "s"(c)
      : "vcc");
  return gl_mk(r0, r1);
}

// corrections as one 64-bit addend built without carry-writing instructions
GLD u64 red_asm3(u64 lo, u64 hi) {
  u32 hl = (u32)hi, hh = (u32)(hi >> 32);
  u64 t, c;
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=s"(c) : "v"(hl), "v"(lo));
  u32 t0 = (u32)t, t1 = (u32)(t >> 32), u0, u1, d0, d1;
  asm("v_sub_co_u32 %0, vcc, %4, %6\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "v_cndmask_b32 %3, 0, -1, vcc\n\t"          // mb
      "v_cndmask_b32 %2, 0, -1, %7\n\t"           // mc
      "v_sub_u32 %2, %2, %3\n\t"                   // d0 = mc - mb
      "s_andn2_b64 vcc, vcc, %7\n\t"               // b and not c
      "v_cndmask_b32 %3, 0, -1, vcc"                 // d1
      : "=&v"(u0), "=&v"(u1), "=&v"(d0), "=&v"(d1)
      : "v"(t0), "v"(t1), "v"(hh), "s"(c)
      : "vcc");
  return gl_mk(u0, u1) + gl_mk(d0, d1);
}
GLD u64 mulw_asm3(u64 a, u64 b) { u64 lo, hi; gl_mul_wide(a, b, lo, hi); return red_asm3(lo, hi); }

GLD u64 m2p24(u64 x) { return gl_canon(gl_reduce96w(x << 24, x >> 40)); }
GLD u64 m2p48(u64 x) { return gl_reduce128(x << 48, x >> 16); }
GLD u64 m2p72(u64 x) { return gl_sub(gl_reduce128(0, x << 8), (x >> 56) << 32); }

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
      if (OP == 18) a[i] = mulw_asm(a[i], a[(i + 1) & 7]);
      if (OP == 19) a[i] = red_asm(a[i], a[(i + 1) & 7]);
      if (OP == 20) a[i] = red_asm2(a[i], a[(i + 1) & 7]);
      if (OP == 21) a[i] = red_asm3(a[i], a[(i + 1) & 7]);
      if (OP == 23) a[i] = m2p24(a[i]) ^ a[(i + 1) & 7];
      if (OP == 24) a[i] = m2p48(a[i]) ^ a[(i + 1) & 7];
      if (OP == 25) a[i] = m2p72(a[i]) ^ a[(i + 1) & 7];
      if (OP == 26) a[i] = gl_canon(a[i] + a[(i + 1) & 7]);
      if (OP == 22) a[i] = mulw_asm3(a[i], a[(i + 1) & 7]);
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
__global__ void __launch_bounds__(256) kperm1(u64* out, u64 seed, int reps) {  // the original Poseidon permutation
  u64 s[12];
  for (int i = 0; i < 12; i++) s[i] = (seed * (threadIdx.x + 1 + i * 977) + blockIdx.x) % GL_P;
  for (int r = 0; r < reps; r++) poseidon_perm(s);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0];
}
__global__ void kcheck(u64* bad, u64 seed) {
  u64 x = seed * (blockIdx.x * 256 + threadIdx.x + 1), y = x * 0xD1342543DE82EF95ull + 12345;
  for (int it = 0; it < 64; it++) {
    x = x * 6364136223846793005ull + 1442695040888963407ull; y = y * 6364136223846793005ull + x;
    u64 lo = x, hi = y;
    if (it % 8 == 0) hi |= 0xFFFFFFFF00000000ull;
    if (it % 8 == 1) lo = 0;
    if (it % 8 == 2) { lo = ~0ull; hi = ~0ull - 0xFFFFFFFFull; }  // largest product hi
    if (it % 8 == 3) hi &= 0xFFFFFFFFull;
    u64 hmax = 0xFFFFFFFE00000001ull;  // (2^64-1)^2 >> 64
    if (hi > hmax) hi = hmax;
    u64 want = gl_canon(red64(lo, hi));  // portable formulation
    u64 h32 = hi & 0xFFFFFFFFull;
    {  // the multiply-reduce forms against 128-bit integer arithmetic reduced the portable way (operands: the same stream, with the extremes)
      u64 a = x, b = y, cadd = lo ^ (hi << 7);
      if (it % 8 == 4) { a = ~0ull; b = ~0ull; cadd = ~0ull; }
      if (it % 8 == 5) { a = 0xFFFFFFFF00000000ull; b = 0xFFFFFFFFull; }
      if (it % 8 == 6) { a = 0xFFFFFFFFull; b = 0xFFFFFFFF00000001ull; cadd = 0xFFFFFFFF00000000ull; }
      unsigned __int128 pr = (unsigned __int128)a * b;
      u64 want_m = gl_canon(red64((u64)pr, (u64)(pr >> 64)));
      unsigned __int128 pa = pr + cadd;
      u64 want_a = gl_canon(red64((u64)pa, (u64)(pa >> 64)));
      if (gl_canon(gl_mulw(a, b)) != want_m || gl_canon(gl_mul_addw(a, b, cadd)) != want_a || gl_mul(a, b) != want_m) atomicAdd((unsigned long long*)bad, 1ull);
    }
    if (gl_canon(gl_reduce128w(lo, hi)) != want || gl_canon(gl_reduce96w(lo, h32)) != gl_canon(red64(lo, h32)) ||
        gl_canon(red_asm(lo, hi)) != want || gl_canon(red_asm2(lo, hi)) != want || gl_canon(red_asm3(lo, hi)) != want)
      atomicAdd((unsigned long long*)bad, 1ull);
  }
}
// ---- single-instruction streams (round 6): what ONE instruction of each class in the leaf sponge's listing costs, in issue slots of
// the 32-bit add, measured as 8 independent streams per lane like the rows above. asm volatile: the instruction named is the
// instruction issued. profiles/r06/leaf_sponge_mix.json prices the kernel's instruction histogram with these rows.
template <int OP>
__global__ void __launch_bounds__(256) kinst(u64* out, u64 seed) {
  u64 a[8];
  u32 x[8];
  for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + 1 + i * 977) + blockIdx.x; x[i] = (u32)(a[i] >> 7); }
  u64 sc = seed | (blockIdx.x & 1);  // a wave-uniform 64-bit mask in SGPRs (v_cndmask's condition)
  u32 ones = ~0u;
  asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x[0]), "v"(x[1]) : "vcc");  // some mask in VCC for the rows that select by it
  asm volatile("" : "+v"(ones));
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int j = (i + 1) & 7;
      if (OP == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(x[j]));
      if (OP == 1) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[j]));
      if (OP == 2) { u64 c; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(a[i]), "=s"(c) : "v"(x[i]), "v"(x[j]), "v"(a[i])); }
      if (OP == 3) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(x[j]), "s"(sc));
      if (OP == 4) { u32 t; asm volatile("v_sub_co_u32 %0, vcc, %2, %3\n\tv_subbrev_co_u32 %1, vcc, 0, %4, vcc\n\tv_subb_co_u32 %0, vcc, %0, %0, vcc"
                                         : "=&v"(x[i]), "=&v"(t) : "v"(x[i]), "v"(x[j]), "v"((u32)a[i]) : "vcc"); x[i] ^= t; }  // three carry-chain instructions (+ one xor)
      if (OP == 5) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(x[j]));
      if (OP == 6) asm volatile("v_bitop3_b32 %0, %1, %2, %1 bitop3:0x30" : "=v"(x[i]) : "v"(x[i]), "v"(x[j]));
      if (OP == 7) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(x[j]));
      if (OP == 8) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(x[j]));
      if (OP == 9) { u64 c; asm volatile("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(a[i]), "=s"(c) : "v"(x[j]), "v"(a[i])); }  // acc += zext(word)
      if (OP == 10) { u64 t = gl_mk(x[j], 0u); asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(a[i]) : "v"(a[i]), "v"(t)); }  // acc += zext(word) as hipcc does it
      if (OP == 11) asm volatile("v_cndmask_b32_e64 %0, %1, %2, vcc" : "=v"(x[i]) : "v"(x[i]), "v"(x[j]) : );       // the mask in VCC, VOP3 encoding
      if (OP == 12) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(x[i]) : "v"(x[i]), "v"(x[j]) : );       // the mask in VCC, VOP2 encoding
      if (OP == 13) asm volatile("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(x[i]) : "s"(sc));                            // constants selected by an SGPR mask (gl_reduce128w's form)
      if (OP == 14) { u64 c; asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\tv_cndmask_b32_e32 %1, 0, %5, vcc" : "=&v"(a[i]), "=&v"(x[i]) : "v"(x[i]), "v"(x[j]), "v"(a[i]), "v"(ones) : "vcc"); (void)c; }  // mad with its carry in VCC + the mask from it (2 instructions)
      if (OP == 15) { u64 c; asm volatile("v_mad_u64_u32 %0, %1, %3, %4, %5\n\ts_nop 1\n\tv_cndmask_b32_e64 %2, 0, -1, %1" : "=&v"(a[i]), "=&s"(c), "=&v"(x[i]) : "v"(x[i]), "v"(x[j]), "v"(a[i])); }  // the same through an SGPR pair (today's form; 2 VALU instructions)
    }
  }
  u64 r = 0;
  for (int i = 0; i < 8; i++) r ^= a[i] ^ x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
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
  {
    u64* bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(kcheck, dim3(4096), dim3(256), 0, 0, bad, 0x9E3779B97F4A7C15ull);
    u64 h = 1; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("asm reduce mismatches: %llu\n", (unsigned long long)h);
  }
  const char* names[] = {"gl_mul", "v_mad_u64_u32", "add64", "cmp+sel+add64", "mul_lo_u32", "mul_hi_u32", "gl_add", "add32", "gl_mul_small", "add_co+addc", "cmp32+sel", "mulw(64bit)", "mulw(32chain)", "mul_wide", "red(64bit)", "red(32chain)", "gl_sub", "gl_addw", "mulw(asm red)", "red(asm)", "red(asm2)", "red(asm3)", "mulw(asm3)", "mul 2^24", "mul 2^48", "mul 2^72", "gl_canon"};
  double ops = (double)blocks * threads * ITERS * 8;
#define RUN(N) { float ms = timeit([&] { hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(threads), 0, 0, d, 0x9E3779B97F4A7C15ull); }); \
    printf("%-14s %8.3f ms  %8.2f Gop/s (lane-ops)\n", names[N], ms, ops / ms / 1e6); }
  // issue slots per operation = (rate of the full-rate 32-bit add, measured first) / (rate of the operation). The dropped reduction
  // variants asm2 / asm3 (rows 20-22 of the round-1 file; the mulw(asm3) row there was a failed launch, not a measurement) are not run.
  float add32_ms = timeit([&] { hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(threads), 0, 0, d, 0x9E3779B97F4A7C15ull); });
#undef RUN
#define RUN(N) { float ms = timeit([&] { hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(threads), 0, 0, d, 0x9E3779B97F4A7C15ull); }); \
    hipError_t e_ = hipGetLastError(); \
    printf("%-14s %8.3f ms  %8.2f Gop/s (lane-ops)  %5.1f slots%s\n", names[N], ms, ops / ms / 1e6, ms / add32_ms, e_ == hipSuccess ? "" : "  LAUNCH FAILED"); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(23) RUN(24) RUN(25) RUN(26)
  {
    const char* inames[] = {"v_mov_b32", "v_lshl_add_u64", "v_mad_u64_u32(asm)", "v_cndmask_b32(sgpr)", "sub_co+subbrev+subb(+xor)", "v_sub_u32", "v_bitop3_b32", "v_add_u32",
                            "v_xor_b32", "mad acc+=zext(w)", "mov,mov,lshl_add acc+=zext(w)", "v_cndmask_b32_e64 vcc", "v_cndmask_b32_e32 vcc", "v_cndmask 0,-1,sgpr",
                            "mad(vcc)+cndmask_e32(vcc)", "mad(sgpr)+cndmask_e64(sgpr)"};
#define RUNI(N) { float ms = timeit([&] { hipLaunchKernelGGL(kinst<N>, dim3(blocks), dim3(threads), 0, 0, d, 0x9E3779B97F4A7C15ull); }); \
    hipError_t e_ = hipGetLastError(); \
    printf("inst %-30s %8.3f ms  %5.2f slots%s\n", inames[N], ms, ms / add32_ms, e_ == hipSuccess ? "" : "  LAUNCH FAILED"); }
    RUNI(0) RUNI(1) RUNI(2) RUNI(3) RUNI(4) RUNI(5) RUNI(6) RUNI(7) RUNI(8) RUNI(9) RUNI(10) RUNI(11) RUNI(12) RUNI(13) RUNI(14) RUNI(15)
  }
  {
    int reps = 64;
    float ms = timeit([&] { hipLaunchKernelGGL(kperm, dim3(blocks), dim3(threads), 0, 0, d, 0x9E3779B97F4A7C15ull, reps); });
    printf("poseidon2_perm %8.3f ms  %8.3f Gperm/s\n", ms, (double)blocks * threads * reps / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(kperm1, dim3(blocks), dim3(threads), 0, 0, d, 0x9E3779B97F4A7C15ull, reps); });
    printf("poseidon_perm  %8.3f ms  %8.3f Gperm/s\n", ms, (double)blocks * threads * reps / ms / 1e6);
  }
  return 0;
}
