// TEST INFRASTRUCTURE -- CPU oracle (see gl.h header). Never linked into the product.
//
// Radix-2 FFT in plonky2's ordering and PolynomialBatch's LDE-to-leaves.
// Follows (absent [dep] sources, restated): plonky2_field field/src/fft.rs (fft: natural-order
// coefficients -> natural-order values v[i] = P(w^i); ifft; ), field/src/polynomial/mod.rs
// (coset_fft: scale coeff i by shift^i then fft; lde: zero-pad), plonky2/src/fri/oracle.rs
// (PolynomialBatch::from_values / from_coeffs: per-poly ifft, lde(rate_bits).coset_fft(g),
// transpose to [n<<rate][w], reverse_index_bits_in_place on rows).  SURVEY App. B "FFT".
#include "gl.h"
#include <stdlib.h>

static inline size_t bitrev(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}
size_t orc_bitrev(size_t x, unsigned bits) { return bitrev(x, bits); }

// in place, natural -> natural. inverse != 0: values -> coefficients.
void orc_fft(gl_t* a, unsigned log_n, int inverse) {
  size_t n = (size_t)1 << log_n;
  for (size_t i = 0; i < n; i++) {
    size_t j = bitrev(i, log_n);
    if (i < j) { gl_t t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  for (unsigned s = 1; s <= log_n; s++) {
    size_t m = (size_t)1 << s, h = m >> 1;
    gl_t wm = gl_root_of_unity(s);
    if (inverse) wm = gl_inv(wm);
    gl_t* tw = malloc(h * sizeof(gl_t));
    tw[0] = 1;
    for (size_t j = 1; j < h; j++) tw[j] = gl_mul(tw[j - 1], wm);
    for (size_t k = 0; k < n; k += m)
      for (size_t j = 0; j < h; j++) {
        gl_t t = gl_mul(tw[j], a[k + j + h]), u = a[k + j];
        a[k + j] = gl_add(u, t);
        a[k + j + h] = gl_sub(u, t);
      }
    free(tw);
  }
  if (inverse) {
    gl_t ninv = gl_inv((gl_t)n % GL_P);
    for (size_t i = 0; i < n; i++) a[i] = gl_mul(a[i], ninv);
  }
}
void orc_coset_fft(gl_t* a, unsigned log_n, gl_t shift) {
  size_t n = (size_t)1 << log_n;
  gl_t s = 1;
  for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], s); s = gl_mul(s, shift); }
  orc_fft(a, log_n, 0);
}
void orc_coset_ifft(gl_t* a, unsigned log_n, gl_t shift) {
  size_t n = (size_t)1 << log_n;
  orc_fft(a, log_n, 1);
  gl_t si = gl_inv(shift), s = 1;
  for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], s); s = gl_mul(s, si); }
}
void orc_fft_batch(gl_t* a, unsigned log_n, size_t batch, int inverse, gl_t coset_shift) {
  size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(dynamic)
  for (size_t b = 0; b < batch; b++) {
    if (coset_shift && !inverse) orc_coset_fft(a + b * n, log_n, coset_shift);
    else if (coset_shift) orc_coset_ifft(a + b * n, log_n, coset_shift);
    else orc_fft(a + b * n, log_n, inverse);
  }
}
// values of each poly on the coset g*<w_{8n}>, natural order, poly-major: out[w][n<<rate]
void orc_lde_values(const gl_t* coeffs, unsigned log_n, size_t w, unsigned rate_bits, gl_t* out) {
  size_t n = (size_t)1 << log_n, N = n << rate_bits;
#pragma omp parallel for schedule(dynamic)
  for (size_t p = 0; p < w; p++) {
    gl_t* buf = out + p * N;
    memcpy(buf, coeffs + p * n, n * sizeof(gl_t));
    memset(buf + n, 0, (N - n) * sizeof(gl_t));
    orc_coset_fft(buf, log_n + rate_bits, GL_MULT_GEN);
  }
}
// leaves[n<<rate][w], row i = evaluations at g * w_{N}^{bitrev(i)}
void orc_lde_leaves(const gl_t* coeffs, unsigned log_n, size_t w, unsigned rate_bits, gl_t* leaves) {
  size_t n = (size_t)1 << log_n, N = n << rate_bits;
  unsigned lg = log_n + rate_bits;
#pragma omp parallel
  {
    gl_t* buf = malloc(N * sizeof(gl_t));
#pragma omp for schedule(dynamic)
    for (size_t p = 0; p < w; p++) {
      memcpy(buf, coeffs + p * n, n * sizeof(gl_t));
      memset(buf + n, 0, (N - n) * sizeof(gl_t));
      orc_coset_fft(buf, lg, GL_MULT_GEN);
      for (size_t i = 0; i < N; i++) leaves[bitrev(i, lg) * w + p] = buf[i];
    }
    free(buf);
  }
}
