// C ABI of libmp2gpu (include/mp2g.h): contexts, device memory, NTT/LDE, hashing, Merkle trees
// and polynomial commitments. Host-side mirror of plonky2's PolynomialBatch / MerkleTree as the
// reference uses them behind `prove()` (recursion-framework/src/circuit_builder.rs:308).
#include "ctx.h"
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

using namespace mp2g;

static thread_local char g_err[512] = "";
int mp2g::fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail("%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define NEED(c, msg) do { if (!(c)) return fail("invalid argument: %s", msg); } while (0)

extern "C" {

const char* mp2g_last_error(void) { return g_err; }

int mp2g_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mp2g_ctx_create(int device, mp2g_ctx** out) {
  NEED(out, "out");
  // No CPU path: the product fails loudly without a GPU (the CPU oracle lives under oracle/
  // and is test infrastructure only).
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail("no HIP device visible: libmp2gpu has no CPU fallback");
  NEED(device >= 0 && device < n, "device index");
  CK(hipSetDevice(device));
  mp2g_ctx* c = new (std::nothrow) mp2g_ctx();
  if (!c) return fail("out of memory");
  c->device = device;
  { const char* sh = getenv("MP2G_SHARE_SCRATCH"); c->share_scratch = !(sh && atoi(sh) == 0); }  // ctx.h: the provers' working memory, shared per context
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) {
    c->own_stream = true;
    c->ntt.stream = c->stream;
#ifdef MP2G_EXPERIMENT_NTT_PRIORITY
    if (const char* pe = getenv("MP2G_NTT_PRIORITY"); pe && atoi(pe) != 0) {
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      if (hipStreamCreateWithPriority(&c->ntt.hi_stream, hipStreamNonBlocking, greatest) != hipSuccess) c->ntt.hi_stream = nullptr;
      if (c->ntt.hi_stream && (hipEventCreateWithFlags(&c->ntt.ev_fork, hipEventDisableTiming) != hipSuccess ||
                               hipEventCreateWithFlags(&c->ntt.ev_join, hipEventDisableTiming) != hipSuccess)) c->ntt.hi_stream = nullptr;
      static bool said = false;
      if (!said) { said = true; fprintf(stderr, "libmp2gpu (variant): NTT on a priority-%d stream (range %d..%d): %s\n", greatest, least, greatest, c->ntt.hi_stream ? "on" : "FAILED"); }
    }
#endif
    e = hipEventCreate(&c->ev0);
  }
  if (e == hipSuccess) e = hipEventCreate(&c->ev1);
  if (e != hipSuccess) {  // nothing half-built survives a failed create
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return fail("mp2g_ctx_create: %s", hipGetErrorString(e));
  }
  *out = c;
  return 0;
}
void mp2g_ctx_destroy(mp2g_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  c->ntt.plans.clear();
  c->ntt.cosets.clear();
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
uint64_t mp2g_stat_leaf_permutations(void) { return mp2g::leaf_permutations_queued(); }
int mp2g_ctx_sync(mp2g_ctx* c) { NEED(c, "ctx"); CK(hipStreamSynchronize(c->stream)); return 0; }
void* mp2g_ctx_stream(mp2g_ctx* c) { return c ? (void*)c->stream : nullptr; }
int mp2g_ctx_set_stream(mp2g_ctx* c, void* stream) {
  NEED(c, "ctx");
  CK(hipStreamSynchronize(c->stream));
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  c->stream = (hipStream_t)stream;
  c->own_stream = false;
  c->ntt.stream = c->stream;
  return 0;
}
int mp2g_dev_alloc(mp2g_ctx* c, size_t bytes, void** d_ptr) {
  NEED(c && d_ptr, "ctx/d_ptr");
  CK(hipSetDevice(c->device));
  CK(hipMalloc(d_ptr, bytes ? bytes : 8));
  return 0;
}
int mp2g_dev_free(mp2g_ctx* c, void* d_ptr) {
  NEED(c, "ctx");
  CK(hipStreamSynchronize(c->stream));
  CK(hipFree(d_ptr));
  return 0;
}
int mp2g_ctx_make_current(mp2g_ctx* c) {
  NEED(c, "ctx");
  CK(hipSetDevice(c->device));
  return 0;
}
int mp2g_ctx_mem_info(mp2g_ctx* c, size_t* free_bytes, size_t* total_bytes) {
  NEED(c && free_bytes && total_bytes, "ctx / outputs");
  CK(hipSetDevice(c->device));
  CK(hipMemGetInfo(free_bytes, total_bytes));
  return 0;
}
int mp2g_h2d(mp2g_ctx* c, void* d_dst, const void* src, size_t bytes) {
  NEED(c, "ctx");
  CK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}
int mp2g_d2h(mp2g_ctx* c, void* dst, const void* d_src, size_t bytes) {
  NEED(c, "ctx");
  CK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}
int mp2g_d2d_2d(mp2g_ctx* c, void* d_dst, size_t dst_pitch, const void* d_src, size_t src_pitch, size_t width_bytes, size_t rows) {
  NEED(c && d_dst && d_src && dst_pitch >= width_bytes && src_pitch >= width_bytes, "ctx / pointers / pitches");
  if (!width_bytes || !rows) return 0;
  CK(hipMemcpy2DAsync(d_dst, dst_pitch, d_src, src_pitch, width_bytes, rows, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}
int mp2g_host_alloc(mp2g_ctx* c, size_t bytes, void** ptr) {
  NEED(c && ptr, "ctx/ptr");
  CK(hipHostMalloc(ptr, bytes ? bytes : 8, hipHostMallocDefault));
  return 0;
}
int mp2g_host_free(mp2g_ctx* c, void* ptr) {
  NEED(c, "ctx");
  CK(hipStreamSynchronize(c->stream));
  CK(hipHostFree(ptr));
  return 0;
}
int mp2g_h2d_async(mp2g_ctx* c, void* d_dst, const void* src, size_t bytes) {
  NEED(c, "ctx");
  CK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  return 0;
}
int mp2g_wires_from_rows_dev(mp2g_ctx* c, const uint64_t* d_rows, uint64_t* d_wires, uint32_t log_n, uint32_t num_wires, uint32_t batch) {
  NEED(c && d_rows && d_wires && log_n >= 1 && log_n <= 24 && num_wires >= 1 && batch >= 1, "ctx / pointers / shape");
  const u64 n = (u64)1 << log_n;
  // rows [n][w] -> polynomials [w][n]: the 64 x 64 LDS tile transpose with the roles of the two indices exchanged
  for (uint32_t b0 = 0; b0 < batch; b0 += 32768) {  // grid z limit
    const uint32_t nb = batch - b0 < 32768 ? batch - b0 : 32768;
    CK(transpose_to_leaves(c->stream, d_rows + (u64)b0 * n * num_wires, (u32)n, num_wires, num_wires, d_wires + (u64)b0 * n * num_wires, nb,
                           n * num_wires, n * num_wires));
  }
  return 0;
}
int mp2g_timer_start(mp2g_ctx* c) { NEED(c, "ctx"); CK(hipEventRecord(c->ev0, c->stream)); return 0; }
int mp2g_timer_stop(mp2g_ctx* c, float* ms) {
  NEED(c && ms, "ctx/ms");
  CK(hipEventRecord(c->ev1, c->stream));
  CK(hipEventSynchronize(c->ev1));
  CK(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}

// ---- NTT -------------------------------------------------------------------------------------
int mp2g_ntt_dev(mp2g_ctx* c, const uint64_t* d_in, uint64_t* d_out, uint32_t log_n, uint32_t batch, int inverse,
                 uint64_t coset_shift, int bitrev_out) {
  NEED(c && d_in && d_out, "ctx/pointers");
  NEED(log_n <= 24, "log_n <= 24");
  NEED(coset_shift < GL_P, "coset_shift canonical");
  if (batch == 0) return 0;
  const u64 n = (u64)1 << log_n;
  if (log_n == 0) {  // size-1 transform is the identity
    if (d_in != d_out) CK(hipMemcpyAsync(d_out, d_in, batch * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
    return 0;
  }
  if (!inverse) {
    CosetTables* pre = nullptr;
    if (coset_shift) CK(c->ntt.coset(log_n, 0, coset_shift, &pre));
    CK(c->ntt.run((const u64*)d_in, (u64*)d_out, log_n, batch, 0, n, n, false, pre, bitrev_out != 0));
  } else {
    NEED(!bitrev_out || !coset_shift, "inverse coset transform with bit-reversed output is not offered");
    CK(c->ntt.run((const u64*)d_in, (u64*)d_out, log_n, batch, 0, n, n, true, nullptr, bitrev_out != 0));
    if (coset_shift) CK(c->ntt.scale_powers((u64*)d_out, log_n, batch, gl_inv(coset_shift), 1));
  }
  return 0;
}
int mp2g_ntt(mp2g_ctx* c, uint64_t* data, uint32_t log_n, uint32_t batch, int inverse, uint64_t coset_shift, int bitrev_out) {
  NEED(c && data, "ctx/data");
  NEED(log_n <= 24, "log_n <= 24");
  if (batch == 0) return 0;
  size_t bytes = ((size_t)batch << log_n) * sizeof(u64);
  DevBuf buf;
  CK(buf.alloc(bytes));
  CK(hipMemcpyAsync(buf.p, data, bytes, hipMemcpyHostToDevice, c->stream));
  int rc = mp2g_ntt_dev(c, buf.p, buf.p, log_n, batch, inverse, coset_shift, bitrev_out);
  if (rc) return rc;
  CK(hipMemcpyAsync(data, buf.p, bytes, hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}
int mp2g_lde_dev(mp2g_ctx* c, const uint64_t* d_coeffs, uint32_t log_n, uint32_t w, uint32_t rate_bits, uint64_t* d_values) {
  NEED(c && d_coeffs && d_values, "ctx/pointers");
  NEED(log_n >= 1 && log_n <= 24 && rate_bits <= 6, "1 <= log_n <= 24, rate_bits <= 6");
  if (w == 0) return 0;
  CosetTables* pre;
  CK(c->ntt.coset(log_n, rate_bits, GL_MULT_GEN, &pre));
  const u64 n = (u64)1 << log_n;
  CK(c->ntt.run((const u64*)d_coeffs, (u64*)d_values, log_n, w, rate_bits, n, n << rate_bits, false, pre, true));
  return 0;
}
int mp2g_lde_leaves(mp2g_ctx* c, const uint64_t* coeffs, uint32_t log_n, uint32_t w, uint32_t rate_bits, uint64_t* leaves) {
  NEED(c && coeffs && leaves, "ctx/pointers");
  if (w == 0) return 0;
  const size_t n = (size_t)1 << log_n, N = n << rate_bits;
  DevBuf dc, dv, dl;
  CK(dc.alloc(w * n * sizeof(u64)));
  CK(dv.alloc(w * N * sizeof(u64)));
  CK(dl.alloc(w * N * sizeof(u64)));
  CK(hipMemcpyAsync(dc.p, coeffs, w * n * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  int rc = mp2g_lde_dev(c, dc.p, log_n, w, rate_bits, dv.p);
  if (rc) return rc;
  CK(transpose_to_leaves(c->stream, dv.p, w, N, N, dl.p));
  CK(hipMemcpyAsync(leaves, dl.p, w * N * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- hashing / Merkle ------------------------------------------------------------------------
int mp2g_hash_no_pad_batch_dev(mp2g_ctx* c, int variant, const uint64_t* d_in, uint32_t in_len, uint32_t count,
                               uint32_t out_len, uint64_t* d_out) {
  NEED(c && d_out, "ctx/out");
  NEED(variant == 0 || variant == 1, "variant");
  NEED(out_len >= 1, "out_len >= 1");
  CK(hash_no_pad_batch(c->stream, variant, (const u64*)d_in, in_len, count, out_len, (u64*)d_out));
  return 0;
}
int mp2g_hash_no_pad_batch(mp2g_ctx* c, int variant, const uint64_t* in, uint32_t in_len, uint32_t count,
                           uint32_t out_len, uint64_t* out) {
  NEED(c && out, "ctx/out");
  if (count == 0) return 0;
  DevBuf di, dout;
  CK(di.alloc((size_t)count * in_len * sizeof(u64)));
  CK(dout.alloc((size_t)count * out_len * sizeof(u64)));
  if (in_len) CK(hipMemcpyAsync(di.p, in, (size_t)count * in_len * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  int rc = mp2g_hash_no_pad_batch_dev(c, variant, di.p, in_len, count, out_len, dout.p);
  if (rc) return rc;
  CK(hipMemcpyAsync(out, dout.p, (size_t)count * out_len * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}

int mp2g_merkle_build(mp2g_ctx* c, int variant, const uint64_t* leaves, uint32_t leaf_len, uint32_t log_leaves,
                      uint32_t cap_height, mp2g_tree** out) {
  NEED(c && out && (leaves || leaf_len == 0), "ctx/out/leaves");
  NEED(variant == 0 || variant == 1, "variant");
  NEED(log_leaves <= 28, "log_leaves <= 28");
  // plonky2 merkle_tree.rs asserts cap_height <= log2(leaves)
  NEED(cap_height <= log_leaves, "cap_height <= log2(leaves)");
  mp2g_tree* t = new (std::nothrow) mp2g_tree();
  if (!t) return fail("out of memory");
  t->ctx = c; t->variant = variant; t->leaf_len = leaf_len; t->log_leaves = log_leaves; t->cap_h = cap_height;
  const size_t L = (size_t)1 << log_leaves;
  hipError_t e = t->leaves.alloc(L * leaf_len * sizeof(u64));
  if (e == hipSuccess) e = t->levels.alloc(merkle_levels_words(log_leaves, cap_height) * sizeof(u64));
  if (e == hipSuccess && leaf_len) e = hipMemcpyAsync(t->leaves.p, leaves, L * leaf_len * sizeof(u64), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = leaf_hash_row_major(c->stream, variant, t->leaves.p, leaf_len, L, t->levels.p);
  if (e == hipSuccess) e = merkle_reduce(c->stream, variant, t->levels.p, log_leaves, cap_height);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { delete t; return fail("merkle_build: %s", hipGetErrorString(e)); }
  *out = t;
  return 0;
}
int mp2g_merkle_cap(const mp2g_tree* t, uint64_t* cap) {
  NEED(t && cap, "tree/cap");
  size_t capw = (size_t)4 << t->cap_h;
  const u64* src = t->levels.p + merkle_levels_words(t->log_leaves, t->cap_h) - capw;
  CK(hipMemcpyAsync(cap, src, capw * sizeof(u64), hipMemcpyDeviceToHost, t->ctx->stream));
  CK(hipStreamSynchronize(t->ctx->stream));
  return 0;
}
int mp2g_merkle_open(const mp2g_tree* t, const uint32_t* idx, uint32_t n_idx, uint64_t* leaves_out, uint64_t* siblings_out) {
  NEED(t && (idx || !n_idx), "tree/idx");
  if (!n_idx) return 0;
  const u32 L = 1u << t->log_leaves;
  for (u32 i = 0; i < n_idx; i++) NEED(idx[i] < L, "leaf index out of range");
  mp2g_ctx* c = t->ctx;
  const u32 depth = t->log_leaves - t->cap_h;
  DevBuf di, dl, ds;
  CK(di.alloc(n_idx * sizeof(u32)));
  CK(hipMemcpyAsync(di.p, idx, n_idx * sizeof(u32), hipMemcpyHostToDevice, c->stream));
  if (leaves_out && t->leaf_len) {
    CK(dl.alloc((size_t)n_idx * t->leaf_len * sizeof(u64)));
    // row gather from the leaf-major copy: leaf q, limb p at leaves[idx*len + p]
    for (u32 q = 0; q < n_idx; q++)
      CK(hipMemcpyAsync(dl.p + (size_t)q * t->leaf_len, t->leaves.p + (size_t)idx[q] * t->leaf_len,
                        t->leaf_len * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
    CK(hipMemcpyAsync(leaves_out, dl.p, (size_t)n_idx * t->leaf_len * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  }
  if (siblings_out && depth) {
    CK(ds.alloc((size_t)n_idx * depth * 4 * sizeof(u64)));
    CK(merkle_open(c->stream, t->levels.p, t->log_leaves, t->cap_h, (const u32*)di.p, n_idx, ds.p));
    CK(hipMemcpyAsync(siblings_out, ds.p, (size_t)n_idx * depth * 4 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  }
  CK(hipStreamSynchronize(c->stream));
  return 0;
}
void mp2g_merkle_free(mp2g_tree* t) {
  if (!t) return;
  (void)hipStreamSynchronize(t->ctx->stream);
  delete t;
}

// ---- commitments -----------------------------------------------------------------------------
static int batch_alloc(mp2g_ctx* c, int variant, uint32_t log_n, uint32_t w, uint32_t rate_bits, uint32_t cap_height, mp2g_batch** out) {
  NEED(c && out, "ctx/out");
  NEED(variant == 0 || variant == 1, "variant");
  NEED(log_n >= 1 && log_n <= 24, "1 <= log_n <= 24");
  NEED(w >= 1, "w >= 1");
  NEED(rate_bits <= 6 && log_n + rate_bits <= 28, "rate_bits");
  NEED(cap_height <= log_n + rate_bits, "cap_height <= log2(leaves)");
  mp2g_batch* b = new (std::nothrow) mp2g_batch();
  if (!b) return fail("out of memory");
  b->ctx = c; b->variant = variant; b->log_n = log_n; b->w = w; b->rate_bits = rate_bits; b->cap_h = cap_height;
  const size_t n = (size_t)1 << log_n, N = n << rate_bits;
  hipError_t e = b->coeffs.alloc(w * n * sizeof(u64));
  if (e == hipSuccess) e = b->values.alloc(w * N * sizeof(u64));
  if (e == hipSuccess) e = b->levels.alloc(merkle_levels_words(log_n + rate_bits, cap_height) * sizeof(u64));
  if (e != hipSuccess) { delete b; return fail("batch alloc: %s", hipGetErrorString(e)); }
  *out = b;
  return 0;
}
// coefficients already in b->coeffs: LDE + leaf hashing + tree
static hipError_t batch_commit(mp2g_batch* b) {
  mp2g_ctx* c = b->ctx;
  const u64 n = (u64)1 << b->log_n, N = n << b->rate_bits;
  CosetTables* pre;
  hipError_t e = c->ntt.coset(b->log_n, b->rate_bits, GL_MULT_GEN, &pre);
  if (e != hipSuccess) return e;
  e = c->ntt.run(b->coeffs.p, b->values.p, b->log_n, b->w, b->rate_bits, n, N, false, pre, true);
  if (e != hipSuccess) return e;
  e = leaf_hash_poly_major(c->stream, b->variant, b->values.p, b->w, N, N, b->levels.p);
  if (e != hipSuccess) return e;
  return merkle_reduce(c->stream, b->variant, b->levels.p, b->log_n + b->rate_bits, b->cap_h);
}
int mp2g_recommit_from_values_dev(mp2g_ctx* c, mp2g_batch* b, const uint64_t* d_values) {
  NEED(c && b && d_values, "ctx/batch/values");
  NEED(b->ctx == c, "batch belongs to another context");
  const u64 n = (u64)1 << b->log_n;
  CK(c->ntt.run((const u64*)d_values, b->coeffs.p, b->log_n, b->w, 0, n, n, true, nullptr, false));
  CK(batch_commit(b));
  return 0;
}
int mp2g_batch_rehash_dev(mp2g_ctx* c, mp2g_batch* b, int parts) {
  NEED(c && b, "ctx/batch");
  NEED(b->ctx == c, "batch belongs to another context");
  NEED(parts >= 1 && parts <= 3, "parts: 1 = leaf sponges, 2 = tree levels, 3 = both");
  const u64 N = ((u64)1 << b->log_n) << b->rate_bits;
  if (parts & 1) CK(leaf_hash_poly_major(c->stream, b->variant, b->values.p, b->w, N, N, b->levels.p));
  if (parts & 2) CK(merkle_reduce(c->stream, b->variant, b->levels.p, b->log_n + b->rate_bits, b->cap_h));
  return 0;
}
int mp2g_commit_from_values_dev(mp2g_ctx* c, int variant, const uint64_t* d_values, uint32_t log_n, uint32_t w,
                                uint32_t rate_bits, uint32_t cap_height, mp2g_batch** out) {
  NEED(d_values, "values");
  mp2g_batch* b;
  int rc = batch_alloc(c, variant, log_n, w, rate_bits, cap_height, &b);
  if (rc) return rc;
  rc = mp2g_recommit_from_values_dev(c, b, d_values);
  if (rc) { delete b; return rc; }
  *out = b;
  return 0;
}
int mp2g_commit_from_coeffs_dev(mp2g_ctx* c, int variant, const uint64_t* d_coeffs, uint32_t log_n, uint32_t w,
                                uint32_t rate_bits, uint32_t cap_height, mp2g_batch** out) {
  NEED(d_coeffs, "coeffs");
  mp2g_batch* b;
  int rc = batch_alloc(c, variant, log_n, w, rate_bits, cap_height, &b);
  if (rc) return rc;
  hipError_t e = hipMemcpyAsync(b->coeffs.p, d_coeffs, ((size_t)w << log_n) * sizeof(u64), hipMemcpyDeviceToDevice, c->stream);
  if (e == hipSuccess) e = batch_commit(b);
  if (e != hipSuccess) { delete b; return fail("commit_from_coeffs: %s", hipGetErrorString(e)); }
  *out = b;
  return 0;
}
int mp2g_commit_from_values(mp2g_ctx* c, int variant, const uint64_t* values, uint32_t log_n, uint32_t w,
                            uint32_t rate_bits, uint32_t cap_height, mp2g_batch** out) {
  NEED(c && values, "ctx/values");
  NEED(log_n <= 24 && w >= 1, "log_n/w");
  DevBuf dv;
  size_t bytes = ((size_t)w << log_n) * sizeof(u64);
  CK(dv.alloc(bytes));
  CK(hipMemcpyAsync(dv.p, values, bytes, hipMemcpyHostToDevice, c->stream));
  int rc = mp2g_commit_from_values_dev(c, variant, dv.p, log_n, w, rate_bits, cap_height, out);
  if (rc) return rc;
  CK(hipStreamSynchronize(c->stream));
  return 0;
}
int mp2g_batch_cap(const mp2g_batch* b, uint64_t* cap) {
  NEED(b && cap, "batch/cap");
  size_t capw = (size_t)4 << b->cap_h;
  const u64* src = b->levels.p + merkle_levels_words(b->log_n + b->rate_bits, b->cap_h) - capw;
  CK(hipMemcpyAsync(cap, src, capw * sizeof(u64), hipMemcpyDeviceToHost, b->ctx->stream));
  CK(hipStreamSynchronize(b->ctx->stream));
  return 0;
}
int mp2g_batch_coeffs(const mp2g_batch* b, uint64_t* coeffs) {
  NEED(b && coeffs, "batch/coeffs");
  CK(hipMemcpyAsync(coeffs, b->coeffs.p, ((size_t)b->w << b->log_n) * sizeof(u64), hipMemcpyDeviceToHost, b->ctx->stream));
  CK(hipStreamSynchronize(b->ctx->stream));
  return 0;
}
int mp2g_batch_open(const mp2g_batch* b, const uint32_t* idx, uint32_t n_idx, uint64_t* leaves_out, uint64_t* siblings_out) {
  NEED(b && (idx || !n_idx), "batch/idx");
  if (!n_idx) return 0;
  const u32 lg = b->log_n + b->rate_bits;
  for (u32 i = 0; i < n_idx; i++) NEED(idx[i] < (1u << lg), "leaf index out of range");
  mp2g_ctx* c = b->ctx;
  const u32 depth = lg - b->cap_h;
  DevBuf di, dl, ds;
  CK(di.alloc(n_idx * sizeof(u32)));
  CK(hipMemcpyAsync(di.p, idx, n_idx * sizeof(u32), hipMemcpyHostToDevice, c->stream));
  if (leaves_out) {
    CK(dl.alloc((size_t)n_idx * b->w * sizeof(u64)));
    CK(gather_rows(c->stream, b->values.p, b->w, (u64)1 << lg, (const u32*)di.p, n_idx, dl.p));
    CK(hipMemcpyAsync(leaves_out, dl.p, (size_t)n_idx * b->w * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  }
  if (siblings_out && depth) {
    CK(ds.alloc((size_t)n_idx * depth * 4 * sizeof(u64)));
    CK(merkle_open(c->stream, b->levels.p, lg, b->cap_h, (const u32*)di.p, n_idx, ds.p));
    CK(hipMemcpyAsync(siblings_out, ds.p, (size_t)n_idx * depth * 4 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  }
  CK(hipStreamSynchronize(c->stream));
  return 0;
}
void mp2g_batch_free(mp2g_batch* b) {
  if (!b) return;
  (void)hipStreamSynchronize(b->ctx->stream);
  delete b;
}

}  // extern "C"
