// Arrow C Data Interface front end (include/olap_arrow.h): imports the callers' arrays by
// their data buffers, runs the operators of olap_gpu.h on host operands (pinned staging ring,
// stage.hip) and exports the results as arrays that share one allocation.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "olap_arrow.h"

namespace {

// One malloc'ed block shared by the arrays exported from one call.
struct Block {
  void* base;
  std::atomic<int64_t> refs;
};

struct Exported {
  Block* block;
  const void* buffers[2];
};

void release_exported(struct ArrowArray* a) {
  if (!a || !a->release) return;
  Exported* e = static_cast<Exported*>(a->private_data);
  if (e) {
    if (e->block && e->block->refs.fetch_sub(1) == 1) {
      std::free(e->block->base);
      delete e->block;
    }
    delete e;
  }
  a->release = nullptr;
  a->private_data = nullptr;
}

void export_array(struct ArrowArray* out, Block* block, const void* data, int64_t length) {
  Exported* e = new Exported{block, {nullptr, data}};
  block->refs.fetch_add(1);
  out->length = length;
  out->null_count = 0;
  out->offset = 0;
  out->n_buffers = 2;
  out->n_children = 0;
  out->buffers = e->buffers;  // validity (none), data
  out->children = nullptr;
  out->dictionary = nullptr;
  out->release = release_exported;
  out->private_data = e;
}

Block* new_block(uint64_t bytes) {
  void* p = nullptr;
  if (posix_memalign(&p, 64, bytes ? bytes : 64) != 0) return nullptr;  // pages are committed when touched
  return new Block{p, {0}};
}

void drop_if_unused(Block* b) {
  if (b && b->refs.load() == 0) {
    std::free(b->base);
    delete b;
  }
}

int dtype_of(olap_ctx* ctx, const struct ArrowSchema* type, int* dtype) {
  if (!type || !type->format) return ctx->fail(OLAP_ERR_INVALID, "null Arrow schema");
  const char* f = type->format;
  if (!std::strcmp(f, "I")) *dtype = OLAP_U32;
  else if (!std::strcmp(f, "i")) *dtype = OLAP_I32;
  else if (!std::strcmp(f, "L")) *dtype = OLAP_U64;
  else if (!std::strcmp(f, "l")) *dtype = OLAP_I64;
  else return ctx->fail(OLAP_ERR_INVALID, "Arrow format '%s' is not a 32/64-bit integer column", f);
  return OLAP_OK;
}

int width_of(int dtype) { return (dtype == OLAP_U32 || dtype == OLAP_I32) ? 4 : 8; }

// data pointers and lengths of the batches of one column
int import_column(olap_ctx* ctx, const struct ArrowArray* const* batches, uint64_t n, int width, std::vector<const void*>* ptrs,
                  std::vector<uint64_t>* rows) {
  if (n && !batches) return ctx->fail(OLAP_ERR_INVALID, "null batch list");
  ptrs->resize(n);
  rows->resize(n);
  for (uint64_t b = 0; b < n; ++b) {
    const struct ArrowArray* a = batches[b];
    if (!a || !a->release) return ctx->fail(OLAP_ERR_INVALID, "batch %llu is null or released", (unsigned long long)b);
    if (a->n_buffers != 2 || a->n_children != 0 || a->dictionary)
      return ctx->fail(OLAP_ERR_INVALID, "batch %llu is not a primitive array", (unsigned long long)b);
    if (a->length < 0 || a->offset < 0) return ctx->fail(OLAP_ERR_INVALID, "batch %llu has a negative length or offset", (unsigned long long)b);
    int64_t nulls = a->null_count;
    if (nulls < 0 && a->buffers[0]) {  // -1 = "not computed" (legal in the C Data Interface): count the cleared validity bits
      const uint8_t* bits = static_cast<const uint8_t*>(a->buffers[0]);
      nulls = 0;
      for (int64_t i = a->offset; i < a->offset + a->length; ++i) nulls += !((bits[i >> 3] >> (i & 7)) & 1);
    }
    if (nulls > 0) return ctx->fail(OLAP_ERR_INVALID, "batch %llu has nulls: the path is defined on non-nullable columns", (unsigned long long)b);
    if (a->length && !a->buffers[1]) return ctx->fail(OLAP_ERR_INVALID, "batch %llu has no data buffer", (unsigned long long)b);
    (*ptrs)[b] = a->length ? static_cast<const char*>(a->buffers[1]) + a->offset * width : nullptr;
    (*rows)[b] = (uint64_t)a->length;
  }
  return OLAP_OK;
}

}  // namespace

extern "C" {

int olap_arrow_filter_lt(olap_ctx* ctx, const struct ArrowSchema* type, const struct ArrowArray* const* batches,
                         uint64_t n_batches, const void* threshold, struct ArrowArray* out_batches, uint64_t* out_total) {
  if (!ctx) return OLAP_ERR_INVALID;
  int dtype;
  OLAP_TRY(dtype_of(ctx, type, &dtype));
  if (n_batches && !out_batches) return ctx->fail(OLAP_ERR_INVALID, "null output array list");
  if (!threshold) return ctx->fail(OLAP_ERR_INVALID, "null threshold");
  const int w = width_of(dtype);
  std::vector<const void*> ptrs;
  std::vector<uint64_t> rows;
  OLAP_TRY(import_column(ctx, batches, n_batches, w, &ptrs, &rows));
  // worst-case layout (like the reference sizes its pull, filter_dpu.cc:67-83): batch b's results start
  // where its input rows start; only the pages that receive results are ever touched
  std::vector<uint64_t> first(n_batches + 1, 0);
  for (uint64_t b = 0; b < n_batches; ++b) first[b + 1] = first[b] + rows[b];
  Block* block = new_block(first[n_batches] * w);
  if (!block) return ctx->fail(OLAP_ERR_NOMEM, "host allocation of %llu bytes failed", (unsigned long long)(first[n_batches] * w));
  std::vector<void*> outs(n_batches);
  std::vector<uint64_t> out_rows(n_batches, 0);
  for (uint64_t b = 0; b < n_batches; ++b) outs[b] = static_cast<char*>(block->base) + first[b] * w;
  const int rc = olap_filter_lt(ctx, dtype, ptrs.data(), rows.data(), n_batches, OLAP_MEM_HOST, threshold, outs.data(), OLAP_MEM_HOST,
                                out_rows.data(), out_total);
  if (rc != OLAP_OK) {
    drop_if_unused(block);
    return rc;
  }
  for (uint64_t b = 0; b < n_batches; ++b) export_array(&out_batches[b], block, outs[b], (int64_t)out_rows[b]);
  drop_if_unused(block);  // n_batches == 0
  return OLAP_OK;
}

int olap_arrow_sum(olap_ctx* ctx, const struct ArrowSchema* type, const struct ArrowArray* const* batches, uint64_t n_batches,
                   uint64_t* out_sum) {
  if (!ctx) return OLAP_ERR_INVALID;
  int dtype;
  OLAP_TRY(dtype_of(ctx, type, &dtype));
  std::vector<const void*> ptrs;
  std::vector<uint64_t> rows;
  OLAP_TRY(import_column(ctx, batches, n_batches, width_of(dtype), &ptrs, &rows));
  return olap_sum(ctx, dtype, ptrs.data(), rows.data(), n_batches, OLAP_MEM_HOST, out_sum);
}

int olap_arrow_take(olap_ctx* ctx, const struct ArrowSchema* data_type, const struct ArrowArray* const* data,
                    const struct ArrowArray* const* indices, uint64_t n_batches, struct ArrowArray* out_batches) {
  if (!ctx) return OLAP_ERR_INVALID;
  int dtype;
  OLAP_TRY(dtype_of(ctx, data_type, &dtype));
  if (n_batches && !out_batches) return ctx->fail(OLAP_ERR_INVALID, "null output array list");
  const int w = width_of(dtype);
  std::vector<const void*> dptrs, iptrs;
  std::vector<uint64_t> drows, irows;
  OLAP_TRY(import_column(ctx, data, n_batches, w, &dptrs, &drows));
  OLAP_TRY(import_column(ctx, indices, n_batches, 4, &iptrs, &irows));
  std::vector<uint64_t> first(n_batches + 1, 0);
  for (uint64_t b = 0; b < n_batches; ++b) first[b + 1] = first[b] + irows[b];
  Block* block = new_block(first[n_batches] * w);
  if (!block) return ctx->fail(OLAP_ERR_NOMEM, "host allocation of %llu bytes failed", (unsigned long long)(first[n_batches] * w));
  std::vector<void*> outs(n_batches);
  for (uint64_t b = 0; b < n_batches; ++b) outs[b] = static_cast<char*>(block->base) + first[b] * w;
  const int rc = olap_take(ctx, dtype, dptrs.data(), drows.data(), reinterpret_cast<const uint32_t* const*>(iptrs.data()), irows.data(),
                           n_batches, OLAP_MEM_HOST, outs.data(), OLAP_MEM_HOST);
  if (rc != OLAP_OK) {
    drop_if_unused(block);
    return rc;
  }
  for (uint64_t b = 0; b < n_batches; ++b) export_array(&out_batches[b], block, outs[b], (int64_t)irows[b]);
  drop_if_unused(block);
  return OLAP_OK;
}

int olap_arrow_join_u32(olap_ctx* ctx, const struct ArrowArray* const* left_fk, const struct ArrowArray* const* left_y,
                        uint64_t n_left_batches, const struct ArrowArray* const* right_pk, const struct ArrowArray* const* right_x,
                        uint64_t n_right_batches, struct ArrowArray* out_columns, olap_join_stats* stats) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (!out_columns) return ctx->fail(OLAP_ERR_INVALID, "null output array list");
  std::vector<const void*> lk, ly, rk, rx;
  std::vector<uint64_t> lrows, lyrows, rrows, rxrows;
  OLAP_TRY(import_column(ctx, left_fk, n_left_batches, 4, &lk, &lrows));
  OLAP_TRY(import_column(ctx, right_pk, n_right_batches, 4, &rk, &rrows));
  if (left_y) {
    OLAP_TRY(import_column(ctx, left_y, n_left_batches, 4, &ly, &lyrows));
    if (lyrows != lrows) return ctx->fail(OLAP_ERR_INVALID, "left key and payload batches differ in length");
  }
  if (right_x) {
    OLAP_TRY(import_column(ctx, right_x, n_right_batches, 4, &rx, &rxrows));
    if (rxrows != rrows) return ctx->fail(OLAP_ERR_INVALID, "right key and payload batches differ in length");
  }
  uint64_t n_left = 0;
  for (uint64_t r : lrows) n_left += r;
  Block* block = new_block(3 * n_left * 4);
  if (!block) return ctx->fail(OLAP_ERR_NOMEM, "host allocation of %llu bytes failed", (unsigned long long)(3 * n_left * 4));
  uint32_t* o = static_cast<uint32_t*>(block->base);
  uint64_t out_rows = 0;
  auto pp = [](const std::vector<const void*>& v) { return reinterpret_cast<const uint32_t* const*>(v.data()); };
  const int rc = olap_join_u32(ctx, pp(lk), left_y ? pp(ly) : nullptr, lrows.data(), n_left_batches, pp(rk), right_x ? pp(rx) : nullptr,
                               rrows.data(), n_right_batches, OLAP_MEM_HOST, o, o + n_left, o + 2 * n_left, n_left, OLAP_MEM_HOST,
                               &out_rows, stats);
  if (rc != OLAP_OK) {
    drop_if_unused(block);
    return rc;
  }
  for (int c = 0; c < 3; ++c) export_array(&out_columns[c], block, o + (uint64_t)c * n_left, (int64_t)out_rows);
  return OLAP_OK;
}

// 64-bit keys (olap_join_u64): fk / pk as uint64 / int64 arrays, payloads uint32 / int32 (or NULL: selection vectors).
// out_columns[0] = fk as 64-bit values, [1] = y, [2] = x (32-bit), all of left-row length (the rows of out_columns[0] first in
// the block: 8-byte aligned).
int olap_arrow_join_u64(olap_ctx* ctx, const struct ArrowArray* const* left_fk, const struct ArrowArray* const* left_y,
                        uint64_t n_left_batches, const struct ArrowArray* const* right_pk, const struct ArrowArray* const* right_x,
                        uint64_t n_right_batches, struct ArrowArray* out_columns, olap_join_stats* stats) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (!out_columns) return ctx->fail(OLAP_ERR_INVALID, "null output array list");
  std::vector<const void*> lk, ly, rk, rx;
  std::vector<uint64_t> lrows, lyrows, rrows, rxrows;
  OLAP_TRY(import_column(ctx, left_fk, n_left_batches, 8, &lk, &lrows));
  OLAP_TRY(import_column(ctx, right_pk, n_right_batches, 8, &rk, &rrows));
  if (left_y) {
    OLAP_TRY(import_column(ctx, left_y, n_left_batches, 4, &ly, &lyrows));
    if (lyrows != lrows) return ctx->fail(OLAP_ERR_INVALID, "left key and payload batches differ in length");
  }
  if (right_x) {
    OLAP_TRY(import_column(ctx, right_x, n_right_batches, 4, &rx, &rxrows));
    if (rxrows != rrows) return ctx->fail(OLAP_ERR_INVALID, "right key and payload batches differ in length");
  }
  uint64_t n_left = 0;
  for (uint64_t r : lrows) n_left += r;
  Block* block = new_block(n_left * 16);
  if (!block) return ctx->fail(OLAP_ERR_NOMEM, "host allocation of %llu bytes failed", (unsigned long long)(n_left * 16));
  uint64_t* ok = static_cast<uint64_t*>(block->base);
  uint32_t* o32 = reinterpret_cast<uint32_t*>(ok + n_left);
  uint64_t out_rows = 0;
  auto k64 = [](const std::vector<const void*>& v) { return reinterpret_cast<const uint64_t* const*>(v.data()); };
  auto p32 = [](const std::vector<const void*>& v) { return reinterpret_cast<const uint32_t* const*>(v.data()); };
  const int rc = olap_join_u64(ctx, k64(lk), left_y ? p32(ly) : nullptr, lrows.data(), n_left_batches, k64(rk), right_x ? p32(rx) : nullptr,
                               rrows.data(), n_right_batches, OLAP_MEM_HOST, ok, o32, o32 + n_left, n_left, OLAP_MEM_HOST, &out_rows, stats);
  if (rc != OLAP_OK) {
    drop_if_unused(block);
    return rc;
  }
  export_array(&out_columns[0], block, ok, (int64_t)out_rows);
  export_array(&out_columns[1], block, o32, (int64_t)out_rows);
  export_array(&out_columns[2], block, o32 + n_left, (int64_t)out_rows);
  return OLAP_OK;
}

}  // extern "C"
