// C ABI of the Ecgfp5 multiset-digest path (include/mp2g.h), over the kernels of ecgfp5.hip.
#include "ctx.h"
#include "ecgfp5.h"

using namespace mp2g;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail("%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define NEED(c, msg) do { if (!(c)) return fail("invalid argument: %s", msg); } while (0)

static int copy_out(mp2g_ctx* c, const DevBuf& dw, const DevBuf& dwei, size_t count, uint64_t* out_w, uint64_t* out_wei) {
  if (out_w) CK(hipMemcpyAsync(out_w, dw.p, count * 5 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  if (out_wei) CK(hipMemcpyAsync(out_wei, dwei.p, count * 11 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" {

int mp2g_map_to_curve_batch(mp2g_ctx* c, int variant, const uint64_t* in, uint32_t in_len, uint32_t count,
                            uint64_t* out_w, uint64_t* out_wei) {
  NEED(c && (in || !in_len), "ctx/in");
  NEED(variant == 0 || variant == 1, "variant");
  if (!count) return 0;
  DevBuf di, dw, dwei;
  CK(di.alloc((size_t)count * in_len * sizeof(u64)));
  CK(dw.alloc((size_t)count * 5 * sizeof(u64)));
  CK(dwei.alloc((size_t)count * 11 * sizeof(u64)));
  if (in_len) CK(hipMemcpyAsync(di.p, in, (size_t)count * in_len * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(ec_map_to_curve(c->stream, variant, di.p, in_len, count, out_w ? dw.p : nullptr, out_wei ? dwei.p : nullptr, nullptr));
  return copy_out(c, dw, dwei, count, out_w, out_wei);
}

// decode `count` encodings into a fractional-coordinate buffer; errors on invalid encodings
static int decode_host(mp2g_ctx* c, const uint64_t* pts_w, uint32_t count, DevBuf& frac) {
  DevBuf dw, dbad;
  CK(dw.alloc((size_t)count * 5 * sizeof(u64)));
  CK(dbad.alloc(8));
  CK(frac.alloc((size_t)(count ? count : 1) * 20 * sizeof(u64)));
  CK(hipMemsetAsync(dbad.p, 0, 8, c->stream));
  CK(hipMemcpyAsync(dw.p, pts_w, (size_t)count * 5 * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  CK(ec_decode(c->stream, dw.p, count, frac.p, (u32*)dbad.p));
  u32 bad = 0;
  CK(hipMemcpyAsync(&bad, dbad.p, 4, hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  if (bad) return fail("invalid point encoding (Point::decode failed)");
  return 0;
}

int mp2g_curve_sum(mp2g_ctx* c, const uint64_t* pts_w, uint32_t count, uint64_t out_w[5], uint64_t out_wei[11]) {
  NEED(c && (pts_w || !count), "ctx/pts");
  DevBuf frac, scratch, dw, dwei;
  int rc = decode_host(c, pts_w, count, frac);
  if (rc) return rc;
  CK(scratch.alloc(20 * 1025 * sizeof(u64)));
  CK(dw.alloc(5 * sizeof(u64)));
  CK(dwei.alloc(11 * sizeof(u64)));
  CK(ec_sum(c->stream, frac.p, count, scratch.p));
  CK(ec_emit(c->stream, scratch.p, 1, out_w ? dw.p : nullptr, out_wei ? dwei.p : nullptr));
  return copy_out(c, dw, dwei, 1, out_w, out_wei);
}

int mp2g_curve_sum_ranges(mp2g_ctx* c, const uint64_t* pts_w, uint32_t count, const uint32_t* ranges, uint32_t n_ranges,
                          uint64_t* out_w, uint64_t* out_wei) {
  NEED(c && (pts_w || !count) && (ranges || !n_ranges), "ctx/pts/ranges");
  for (uint32_t i = 0; i < n_ranges; i++) NEED(ranges[2 * i] <= ranges[2 * i + 1] && ranges[2 * i + 1] <= count, "range outside the points");
  if (!n_ranges) return 0;
  DevBuf frac, dr, fout, dw, dwei;
  int rc = decode_host(c, pts_w, count, frac);
  if (rc) return rc;
  CK(dr.alloc((size_t)n_ranges * 8));
  CK(fout.alloc((size_t)n_ranges * 20 * sizeof(u64)));
  CK(dw.alloc((size_t)n_ranges * 5 * sizeof(u64)));
  CK(dwei.alloc((size_t)n_ranges * 11 * sizeof(u64)));
  CK(hipMemcpyAsync(dr.p, ranges, (size_t)n_ranges * 8, hipMemcpyHostToDevice, c->stream));
  CK(ec_sum_ranges(c->stream, frac.p, (const u32*)dr.p, n_ranges, fout.p));
  CK(ec_emit(c->stream, fout.p, n_ranges, out_w ? dw.p : nullptr, out_wei ? dwei.p : nullptr));
  return copy_out(c, dw, dwei, n_ranges, out_w, out_wei);
}

int mp2g_scalar_mul_batch(mp2g_ctx* c, const uint64_t* pts_w, const uint32_t* scalars, uint32_t count, uint64_t* out_w,
                          uint64_t* out_wei) {
  NEED(c && ((pts_w && scalars) || !count), "ctx/pts/scalars");
  if (!count) return 0;
  DevBuf frac, fout, ds, dw, dwei;
  int rc = decode_host(c, pts_w, count, frac);
  if (rc) return rc;
  CK(fout.alloc((size_t)count * 20 * sizeof(u64)));
  CK(ds.alloc((size_t)count * 16));
  CK(dw.alloc((size_t)count * 5 * sizeof(u64)));
  CK(dwei.alloc((size_t)count * 11 * sizeof(u64)));
  CK(hipMemcpyAsync(ds.p, scalars, (size_t)count * 16, hipMemcpyHostToDevice, c->stream));
  CK(ec_scalar_mul(c->stream, frac.p, (const u32*)ds.p, count, fout.p));
  CK(ec_emit(c->stream, fout.p, count, out_w ? dw.p : nullptr, out_wei ? dwei.p : nullptr));
  return copy_out(c, dw, dwei, count, out_w, out_wei);
}

int mp2g_field_hashed_scalar_mul(mp2g_ctx* c, int variant, const uint64_t* inputs, uint32_t n_inputs, const uint64_t base_w[5],
                                 uint64_t out_w[5], uint64_t out_wei[11]) {
  NEED(c && base_w && (inputs || !n_inputs), "ctx/pointers");
  uint64_t h[4];
  int rc = mp2g_hash_no_pad_batch(c, variant, inputs, n_inputs, 1, 4, h);
  if (rc) return rc;
  // hash_to_int_value: e0 + e1 * 2^64 as little-endian u32 limbs
  uint32_t k[4] = {(uint32_t)h[0], (uint32_t)(h[0] >> 32), (uint32_t)h[1], (uint32_t)(h[1] >> 32)};
  return mp2g_scalar_mul_batch(c, base_w, k, 1, out_w, out_wei);
}

int mp2g_row_digest_batch_dev(mp2g_ctx* c, int variant, const uint64_t* d_col_ids, uint32_t n_cols, const uint32_t* d_values,
                              const uint32_t* d_unique, uint32_t n_unique, uint32_t rows, uint64_t* d_frac_out,
                              uint64_t out_w[5], uint64_t out_wei[11]) {
  NEED(c && (d_col_ids || !n_cols) && (d_values || !n_cols || !rows), "ctx/pointers");
  NEED(variant == 0 || variant == 1, "variant");
  DevBuf frac, scratch, dw, dwei;
  CK(frac.alloc((size_t)(rows ? rows : 1) * 20 * sizeof(u64)));
  CK(scratch.alloc(20 * 1025 * sizeof(u64)));
  CK(dw.alloc(5 * sizeof(u64)));
  CK(dwei.alloc(11 * sizeof(u64)));
  CK(ec_row_digest(c->stream, variant, (const u64*)d_col_ids, n_cols, d_values, d_unique, n_unique, rows, frac.p));
  CK(ec_sum(c->stream, frac.p, rows, scratch.p));
  if (d_frac_out) CK(hipMemcpyAsync(d_frac_out, scratch.p, 20 * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
  if (out_w || out_wei) CK(ec_emit(c->stream, scratch.p, 1, out_w ? dw.p : nullptr, out_wei ? dwei.p : nullptr));
  return copy_out(c, dw, dwei, 1, out_w, out_wei);
}

int mp2g_row_digests(mp2g_ctx* c, int variant, const uint64_t* col_ids, uint32_t n_cols, const uint32_t* values, const uint32_t* unique,
                     uint32_t n_unique, uint32_t rows, uint64_t* out_w, uint64_t* out_wei) {
  NEED(c && (col_ids || !n_cols), "ctx/col_ids");
  NEED(variant == 0 || variant == 1, "variant");
  NEED((values || !n_cols || !rows) && (unique || !n_unique || !rows), "values/unique");
  if (!rows) return 0;
  DevBuf dc, dv, du, frac, dw, dwei;
  CK(dc.alloc((size_t)n_cols * sizeof(u64)));
  CK(dv.alloc((size_t)rows * n_cols * 32));
  CK(du.alloc((size_t)rows * n_unique * 32));
  CK(frac.alloc((size_t)rows * 20 * sizeof(u64)));
  CK(dw.alloc((size_t)rows * 5 * sizeof(u64)));
  CK(dwei.alloc((size_t)rows * 11 * sizeof(u64)));
  if (n_cols) CK(hipMemcpyAsync(dc.p, col_ids, (size_t)n_cols * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  if (n_cols) CK(hipMemcpyAsync(dv.p, values, (size_t)rows * n_cols * 32, hipMemcpyHostToDevice, c->stream));
  if (n_unique) CK(hipMemcpyAsync(du.p, unique, (size_t)rows * n_unique * 32, hipMemcpyHostToDevice, c->stream));
  CK(ec_row_digest(c->stream, variant, (const u64*)dc.p, n_cols, (const u32*)dv.p, (const u32*)du.p, n_unique, rows, frac.p));
  CK(ec_emit(c->stream, frac.p, rows, out_w ? dw.p : nullptr, out_wei ? dwei.p : nullptr));
  return copy_out(c, dw, dwei, rows, out_w, out_wei);
}

int mp2g_row_digest_batch(mp2g_ctx* c, int variant, const uint64_t* col_ids, uint32_t n_cols, const uint32_t* values,
                          const uint32_t* unique, uint32_t n_unique, uint32_t rows, uint64_t out_w[5], uint64_t out_wei[11]) {
  NEED(c && (col_ids || !n_cols), "ctx/col_ids");
  NEED((values || !n_cols || !rows) && (unique || !n_unique || !rows), "values/unique");
  DevBuf dc, dv, du;
  CK(dc.alloc((size_t)n_cols * sizeof(u64)));
  CK(dv.alloc((size_t)rows * n_cols * 32));
  CK(du.alloc((size_t)rows * n_unique * 32));
  if (n_cols) CK(hipMemcpyAsync(dc.p, col_ids, (size_t)n_cols * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  if (rows && n_cols) CK(hipMemcpyAsync(dv.p, values, (size_t)rows * n_cols * 32, hipMemcpyHostToDevice, c->stream));
  if (rows && n_unique) CK(hipMemcpyAsync(du.p, unique, (size_t)rows * n_unique * 32, hipMemcpyHostToDevice, c->stream));
  return mp2g_row_digest_batch_dev(c, variant, dc.p, n_cols, (const u32*)dv.p, (const u32*)du.p, n_unique, rows, nullptr, out_w, out_wei);
}

}  // extern "C"
