// Host-side check of the device arithmetic headers (the GLHD functions) against the C oracle.
// build: hipcc -x hip --offload-arch=gfx950 -O2 -std=c++17 -DMP2G_DEVCONST="static const" \
//        -I../../mapreduce-plonky2_amd/csrc perm_host_test.cpp ../../oracle/liboracle.so -o perm_host_test
#include "poseidon.cuh"
#include <cstdio>
#include <cstdlib>
extern "C" void orc_perm(int variant, uint64_t s[12]);
static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
int main() {
  const uint64_t edge[] = {0, 1, GL_P - 1, GL_P - 2, 0xFFFFFFFFull, 0xFFFFFFFF00000000ull, 0x100000000ull, GL_P >> 1};
  long bad = 0;
  for (int variant = 0; variant < 2; variant++)
    for (int it = 0; it < 20000; it++) {
      uint64_t a[12], b[12];
      for (int i = 0; i < 12; i++) {
        uint64_t v = (it < 4000 && (rnd() & 3)) ? edge[rnd() % 8] : rnd() % GL_P;
        a[i] = b[i] = v;
      }
      if (variant == 0) poseidon2_perm(a); else poseidon_perm(a);
      orc_perm(variant, b);
      for (int i = 0; i < 12; i++) if (a[i] != b[i]) bad++;
    }
  // weak primitives on adversarial non-canonical inputs
  for (int it = 0; it < 2000000; it++) {
    uint64_t x = (rnd() & 1) ? ~0ull - (rnd() & 0xFFFFFFFFull) : rnd();
    uint64_t y = (rnd() & 1) ? ~0ull - (rnd() & 0xFFFFFFFFull) : rnd();
    unsigned __int128 pr = (unsigned __int128)x * y;
    uint64_t want = (uint64_t)(pr % GL_P);
    if (gl_canon(gl_mulw(x, y)) != want) bad++;
    uint64_t yc = y % GL_P;
    if (gl_canon(gl_addw(x, yc)) != (uint64_t)(((unsigned __int128)x + yc) % GL_P)) bad++;
  }
  // canonical primitives on edge and random canonical inputs
  for (int it = 0; it < 2000000; it++) {
    uint64_t x = (it < 4096) ? edge[it & 7] : rnd() % GL_P, y = (it < 4096) ? edge[(it >> 3) & 7] : rnd() % GL_P;
    if (gl_add(x, y) != (uint64_t)(((unsigned __int128)x + y) % GL_P)) bad++;
    if (gl_sub(x, y) != (uint64_t)(((unsigned __int128)x + GL_P - y) % GL_P)) bad++;
    if (gl_mul(x, y) != (uint64_t)(((unsigned __int128)x * y) % GL_P)) bad++;
    uint32_t c = (uint32_t)rnd();
    if (gl_mul_small(x, c) != (uint64_t)(((unsigned __int128)x * c) % GL_P)) bad++;
    uint64_t any = rnd() | ((it & 1) ? 0xFFFFFFFF00000000ull : 0);
    if (gl_canon(any) != any % GL_P) bad++;
  }
  printf(bad ? "FAIL %ld\n" : "host arithmetic ok\n", bad);
  return bad != 0;
}
