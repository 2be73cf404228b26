// ctx.hip -- context, arena, timers, column staging of libolap_gpu.so.
// Takes the place of dpu::DpuSet (reference host/dpuext/dpuext.hpp:148-927) and
// of the Arrow-buffer <-> DPU copies (host/dpuext/arrow_utils.cc:25-266).
#include "common.h"

#include <cstring>
#include <new>

namespace olap {

const char* const kPhaseNames[PH_COUNT] = {
    "copy-to-gpu",    "copy-from-gpu",  "sum",
    "filter-count",   "filter-compact", "take",
    "partition-hist", "partition-scan", "partition-scatter",
    "partition-scatter-claim", "partition-scatter-capacity",
    "join-probe",     "gen"};

int ws_begin(olap_ctx* ctx, uint64_t bytes) {
  bytes = align_up(bytes + 4096, 1 << 20);
  if (bytes > ctx->ws_bytes) {
    // Growing means hipFree + hipMalloc (seconds at 100 GB): when an arena exists already, grow with 3 % of
    // headroom so that a run of calls whose sizes creep up (the per-group joins of the multi-GPU path differ
    // by a fraction of a percent) reallocates once, not every time.  If the padded size does not fit, the
    // exact size is tried.
    uint64_t padded = ctx->ws ? align_up(bytes + bytes / 32, 64 << 20) : bytes;
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) {
      OLAP_HIP(ctx, hipFree(ctx->ws));
      ctx->ws = nullptr;
      ctx->ws_bytes = 0;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, padded);
    if (e != hipSuccess && padded != bytes) {
      (void)hipGetLastError();
      padded = bytes;
      e = hipMalloc(&p, padded);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return ctx->fail(OLAP_ERR_NOMEM, "workspace hipMalloc(%llu bytes) failed: %s",
                       (unsigned long long)bytes, hipGetErrorString(e));
    }
    ctx->ws = (char*)p;
    ctx->ws_bytes = padded;
  }
  ctx->ws_used = 0;
  return OLAP_OK;
}

void* ws_alloc(olap_ctx* ctx, uint64_t bytes) {
  uint64_t off = align_up(ctx->ws_used, 256);
  if (off + bytes > ctx->ws_bytes) return nullptr;
  ctx->ws_used = off + bytes;
  return ctx->ws + off;
}

int pinned_begin(olap_ctx* ctx, uint64_t bytes) {
  bytes = align_up(bytes + 4096, 1 << 16);
  if (bytes > ctx->pinned_bytes) {
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pinned) {
      OLAP_HIP(ctx, hipHostFree(ctx->pinned));
      ctx->pinned = nullptr;
      ctx->pinned_bytes = 0;
    }
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return ctx->fail(OLAP_ERR_NOMEM, "pinned hipHostMalloc(%llu bytes) failed: %s",
                       (unsigned long long)bytes, hipGetErrorString(e));
    }
    ctx->pinned = (char*)p;
    ctx->pinned_bytes = bytes;
  }
  ctx->pinned_used = 0;
  return OLAP_OK;
}

void* pinned_alloc(olap_ctx* ctx, uint64_t bytes) {
  uint64_t off = align_up(ctx->pinned_used, 64);
  if (off + bytes > ctx->pinned_bytes) return nullptr;
  ctx->pinned_used = off + bytes;
  return ctx->pinned + off;
}

static hipEvent_t get_event(olap_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void phase_begin(olap_ctx* ctx, int phase) {
  if (!ctx->profiling) return;
  olap_ctx::Pending p{phase, get_event(ctx), get_event(ctx)};
  if (!p.a || !p.b) return;
  (void)hipEventRecord(p.a, ctx->stream);
  ctx->pending.push_back(p);
}

void phase_end(olap_ctx* ctx, int phase) {
  if (!ctx->profiling) return;
  for (size_t i = ctx->pending.size(); i-- > 0;) {
    if (ctx->pending[i].phase == phase) {
      (void)hipEventRecord(ctx->pending[i].b, ctx->stream);
      return;
    }
  }
}

int resolve_timers(olap_ctx* ctx) {
  for (auto& p : ctx->pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ctx->timers[p.phase].ms += ms;
      ctx->timers[p.phase].launches += 1;
    } else {
      (void)hipGetLastError();
    }
    ctx->event_pool.push_back(p.a);
    ctx->event_pool.push_back(p.b);
  }
  ctx->pending.clear();
  return OLAP_OK;
}

uint64_t column_set_bytes(const uint64_t* rows, uint64_t nb, int ncols, const uint32_t* elem_bytes,
                          int in_mem) {
  uint64_t bytes = (uint64_t)ncols * align_up(nb * 8, 256) + 2 * align_up((nb + 1) * 8, 256) + 1024;
  if (in_mem == OLAP_MEM_HOST) {
    for (int c = 0; c < ncols; ++c)
      for (uint64_t b = 0; b < nb; ++b) bytes += align_up(rows[b] * elem_bytes[c], 256);
  }
  return bytes;
}

// NOTE: callers must have sized the arena (ws_begin) and the pinned scratch
// (pinned_begin) to cover column_set_bytes() / the tables before calling this.
int build_column_set(olap_ctx* ctx, const void* const* const* cols, int ncols,
                     const uint32_t* elem_bytes, const uint64_t* rows, uint64_t nb, int in_mem,
                     uint32_t unit_rows, ColumnSet* out) {
  if (nb > 0xFFFFFFF0ull) return ctx->fail(OLAP_ERR_INVALID, "too many batches");
  uint64_t* h_unit = (uint64_t*)pinned_alloc(ctx, (nb + 1) * 8);
  uint64_t* h_row = (uint64_t*)pinned_alloc(ctx, (nb + 1) * 8);
  if (!h_unit || !h_row) return ctx->fail(OLAP_ERR_NOMEM, "pinned scratch exhausted");
  // (hundreds of thousands of batches per call at SF=2048: one division per DISTINCT batch size in a row, not per batch)
  bool uniform = nb > 0;
  h_unit[0] = 0;
  h_row[0] = 0;
  bool full = nb > 0;
  uint64_t last_rows = ~0ull, last_units = 0;
  bool last_whole = false;
  for (uint64_t b = 0; b < nb; ++b) {
    if (rows[b] != last_rows) {
      last_rows = rows[b];
      last_units = (rows[b] + unit_rows - 1) / unit_rows;
      last_whole = last_units * unit_rows == rows[b];
      if (b) uniform = false;
    }
    full = full && last_whole;
    h_row[b + 1] = h_row[b] + last_rows;
    h_unit[b + 1] = h_unit[b] + last_units;
  }
  if (uniform && (rows[0] == 0 || rows[0] > 0xFFFFFFFFull)) uniform = false;
  uint64_t* d_unit = (uint64_t*)ws_alloc(ctx, (nb + 1) * 8);
  uint64_t* d_row = (uint64_t*)ws_alloc(ctx, (nb + 1) * 8);
  if (!d_unit || !d_row) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (tables)");
  OLAP_HIP(ctx, hipMemcpyAsync(d_unit, h_unit, (nb + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  OLAP_HIP(ctx, hipMemcpyAsync(d_row, h_row, (nb + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  ColView& v = out->view;
  v.unit_start = d_unit;
  v.row_start = d_row;
  v.total_units = h_unit[nb];
  v.total_rows = h_row[nb];
  v.nb = (uint32_t)nb;
  v.unit_rows = unit_rows;
  v.units_per_batch = uniform ? (uint32_t)((rows[0] + unit_rows - 1) / unit_rows) : 0;
  v.uniform_rows = uniform ? (uint32_t)rows[0] : 0;
  out->dev_ptr_tables.clear();
  out->h_ptrs.assign((size_t)ncols, std::vector<const void*>());
  out->h_rows.assign(rows, rows + nb);
  out->h_elem.assign(elem_bytes, elem_bytes + ncols);
  v.row0 = nullptr;
  bool contiguous = nb > 0;
  for (int c = 0; c < ncols; ++c) {
    const void** h_ptrs = (const void**)pinned_alloc(ctx, (nb ? nb : 1) * 8);
    const void** d_ptrs = (const void**)ws_alloc(ctx, (nb ? nb : 1) * 8);
    if (!h_ptrs || !d_ptrs) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (ptr tables)");
    if (in_mem == OLAP_MEM_HOST) {
      // staged copies of the batches, back to back in the arena: one DMA per pinned slot
      std::vector<StageItem> items;
      items.reserve(nb);
      for (uint64_t b = 0; b < nb; ++b) {
        uint64_t bytes = rows[b] * elem_bytes[c];
        char* d = (char*)ws_alloc(ctx, bytes ? bytes : 1);
        if (!d) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (staging)");
        if (bytes) {
          if (!cols[c][b]) return ctx->fail(OLAP_ERR_INVALID, "null batch pointer (column %d batch %llu)", c, (unsigned long long)b);
          items.push_back(StageItem{d, const_cast<void*>(cols[c][b]), bytes});
        }
        h_ptrs[b] = d;
      }
      PhaseScope ps(ctx, PH_COPY_TO);
      OLAP_TRY(stage_h2d(ctx, items));
    } else {
      for (uint64_t b = 0; b < nb; ++b) {
        if (rows[b] && !cols[c][b])
          return ctx->fail(OLAP_ERR_INVALID, "null batch pointer (column %d batch %llu)", c, (unsigned long long)b);
        if (((uintptr_t)cols[c][b]) & (uintptr_t)(elem_bytes[c] - 1))  // (element sizes are powers of two)
          return ctx->fail(OLAP_ERR_INVALID, "batch pointer not aligned to its element size (column %d batch %llu)", c, (unsigned long long)b);
        h_ptrs[b] = cols[c][b];
      }
    }
    for (uint64_t b = 0; b < nb; ++b) full = full && (((uintptr_t)h_ptrs[b]) & 15) == 0;
    for (uint64_t b = 0; b + 1 < nb; ++b) contiguous = contiguous && (const char*)h_ptrs[b + 1] == (const char*)h_ptrs[b] + rows[b] * elem_bytes[c];
    out->h_ptrs[(size_t)c].assign(h_ptrs, h_ptrs + nb);
    OLAP_HIP(ctx, hipMemcpyAsync(d_ptrs, h_ptrs, (nb ? nb : 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    out->dev_ptr_tables.push_back((const void* const*)d_ptrs);
  }
  out->full_units = full;
  out->contiguous = full && contiguous;
  return OLAP_OK;
}

uint64_t split_column_set_bytes(uint64_t nb, int ncols) { return 2 * (align_up((nb + 2) * 8, 256) * (3 + (uint64_t)ncols)) + 1024; }

// One derived view: entries (pointers per column, rows, global first row) -> device tables
static int make_piece_view(olap_ctx* ctx, const std::vector<std::vector<const void*>>& ptrs, const std::vector<uint64_t>& rows,
                           const std::vector<uint64_t>& row0, uint32_t unit_rows, bool full, ColumnSet* out) {
  const uint64_t nb = rows.size();
  const int ncols = (int)ptrs.size();
  uint64_t* h_unit = (uint64_t*)pinned_alloc(ctx, (nb + 1) * 8);
  uint64_t* h_row = (uint64_t*)pinned_alloc(ctx, (nb + 1) * 8);
  uint64_t* h_row0 = (uint64_t*)pinned_alloc(ctx, (nb + 1) * 8);
  uint64_t* d_unit = (uint64_t*)ws_alloc(ctx, (nb + 1) * 8);
  uint64_t* d_row = (uint64_t*)ws_alloc(ctx, (nb + 1) * 8);
  uint64_t* d_row0 = (uint64_t*)ws_alloc(ctx, (nb + 1) * 8);
  if (!h_unit || !h_row || !h_row0 || !d_unit || !d_row || !d_row0) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (piece views)");
  h_unit[0] = h_row[0] = 0;
  for (uint64_t b = 0; b < nb; ++b) {
    h_row[b + 1] = h_row[b] + rows[b];
    h_unit[b + 1] = h_unit[b] + (rows[b] + unit_rows - 1) / unit_rows;
    h_row0[b] = row0[b];
  }
  h_row0[nb] = 0;
  OLAP_HIP(ctx, hipMemcpyAsync(d_unit, h_unit, (nb + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  OLAP_HIP(ctx, hipMemcpyAsync(d_row, h_row, (nb + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  OLAP_HIP(ctx, hipMemcpyAsync(d_row0, h_row0, (nb + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  ColView& v = out->view;
  v.unit_start = d_unit;
  v.row_start = d_row;
  v.row0 = d_row0;
  v.total_units = h_unit[nb];
  v.total_rows = h_row[nb];
  v.nb = (uint32_t)nb;
  v.unit_rows = unit_rows;
  bool uniform = nb > 0 && rows[0] > 0 && rows[0] <= 0xFFFFFFFFull;
  for (uint64_t b = 1; b < nb && uniform; ++b) uniform = rows[b] == rows[0];
  v.units_per_batch = uniform ? (uint32_t)((rows[0] + unit_rows - 1) / unit_rows) : 0;  // equal pieces: a division instead of a search
  v.uniform_rows = uniform ? (uint32_t)rows[0] : 0;
  out->dev_ptr_tables.clear();
  out->h_ptrs = ptrs;
  out->h_rows = rows;
  for (int c = 0; c < ncols; ++c) {
    const void** h_p = (const void**)pinned_alloc(ctx, (nb ? nb : 1) * 8);
    const void** d_p = (const void**)ws_alloc(ctx, (nb ? nb : 1) * 8);
    if (!h_p || !d_p) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (piece views)");
    for (uint64_t b = 0; b < nb; ++b) h_p[b] = ptrs[(size_t)c][b];
    OLAP_HIP(ctx, hipMemcpyAsync(d_p, h_p, (nb ? nb : 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    out->dev_ptr_tables.push_back((const void* const*)d_p);
  }
  out->full_units = full && nb > 0;
  return OLAP_OK;
}

int split_column_set(olap_ctx* ctx, const ColumnSet& cs, ColumnSet* whole, ColumnSet* rest) {
  const uint64_t nb = cs.h_rows.size();
  const int ncols = (int)cs.h_ptrs.size();
  const uint32_t unit = cs.view.unit_rows;
  std::vector<std::vector<const void*>> pw((size_t)ncols), pr((size_t)ncols);
  std::vector<uint64_t> rw, rr, zw, zr;
  uint64_t row0 = 0;
  for (uint64_t b = 0; b < nb; ++b) {
    const uint64_t rows = cs.h_rows[b];
    // rows in front of the first 16-byte boundary: an Arrow slice moves every column of a batch by the same number of ELEMENTS,
    // so columns of one element size share it -- those rows go to `rest`, what follows is aligned
    const uint32_t e0 = cs.h_elem[0];
    const uint64_t mis = ((uintptr_t)cs.h_ptrs[0][b]) & 15;
    bool same = mis % e0 == 0;
    for (int c = 1; c < ncols; ++c) same = same && cs.h_elem[(size_t)c] == e0 && (((uintptr_t)cs.h_ptrs[(size_t)c][b]) & 15) == mis;
    const uint64_t head = same ? std::min<uint64_t>(rows, ((16 - mis) & 15) / e0) : rows;  // (different misalignments: all of it to `rest`)
    const uint64_t full_rows = (rows - head) - (rows - head) % unit;
    auto piece = [&](std::vector<std::vector<const void*>>& pp, std::vector<uint64_t>& rws, std::vector<uint64_t>& zs, uint64_t first, uint64_t n) {
      if (!n) return;
      for (int c = 0; c < ncols; ++c) pp[(size_t)c].push_back((const char*)cs.h_ptrs[(size_t)c][b] + first * cs.h_elem[(size_t)c]);
      rws.push_back(n);
      zs.push_back(row0 + first);
    };
    piece(pr, rr, zr, 0, head);
    piece(pw, rw, zw, head, full_rows);
    piece(pr, rr, zr, head + full_rows, rows - head - full_rows);
    row0 += rows;
  }
  OLAP_TRY(make_piece_view(ctx, pw, rw, zw, unit, true, whole));
  OLAP_TRY(make_piece_view(ctx, pr, rr, zr, unit, false, rest));
  whole->h_elem = rest->h_elem = cs.h_elem;
  return OLAP_OK;
}

}  // namespace olap

using namespace olap;

extern "C" {

int olap_abi_version(void) { return OLAP_ABI_VERSION; }

int olap_create(int device_id, olap_ctx** out_ctx) {
  if (!out_ctx) return OLAP_ERR_INVALID;
  *out_ctx = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return OLAP_ERR_DEVICE;
  if (device_id < 0 || device_id >= n) return OLAP_ERR_INVALID;
  if (hipSetDevice(device_id) != hipSuccess) return OLAP_ERR_DEVICE;
  olap_ctx* ctx = new (std::nothrow) olap_ctx();
  if (!ctx) return OLAP_ERR_NOMEM;
  ctx->device = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
    delete ctx;
    return OLAP_ERR_DEVICE;
  }
  ctx->compute_units = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return OLAP_ERR_DEVICE;
  }
  ctx->stream = ctx->own_stream;
  *out_ctx = ctx;
  return OLAP_OK;
}

void olap_destroy(olap_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  resolve_timers(ctx);
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  stager_destroy(ctx);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

const char* olap_last_error(const olap_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int olap_device_info(olap_ctx* ctx, char* name, size_t name_cap, int* compute_units,
                     uint64_t* total_mem_bytes, uint64_t* free_mem_bytes) {
  if (!ctx) return OLAP_ERR_INVALID;
  OLAP_HIP(ctx, hipSetDevice(ctx->device));
  hipDeviceProp_t prop;
  OLAP_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (name && name_cap) {
    std::snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  size_t fr = 0, tot = 0;
  OLAP_HIP(ctx, hipMemGetInfo(&fr, &tot));
  if (total_mem_bytes) *total_mem_bytes = tot;
  if (free_mem_bytes) *free_mem_bytes = fr;
  return OLAP_OK;
}

int olap_set_stream(olap_ctx* ctx, void* hip_stream) {
  if (!ctx) return OLAP_ERR_INVALID;
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  resolve_timers(ctx);
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return OLAP_OK;
}

void* olap_get_stream(olap_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int olap_synchronize(olap_ctx* ctx) {
  if (!ctx) return OLAP_ERR_INVALID;
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return resolve_timers(ctx);
}

int olap_reserve_workspace(olap_ctx* ctx, uint64_t bytes) {
  if (!ctx) return OLAP_ERR_INVALID;
  OLAP_HIP(ctx, hipSetDevice(ctx->device));
  if (bytes <= ctx->ws_bytes) return OLAP_OK;
  return ws_begin(ctx, bytes);
}

uint64_t olap_workspace_bytes(const olap_ctx* ctx) { return ctx ? ctx->ws_bytes : 0; }

int olap_malloc(olap_ctx* ctx, uint64_t bytes, void** out_dptr) {
  if (!ctx || !out_dptr) return OLAP_ERR_INVALID;
  OLAP_HIP(ctx, hipSetDevice(ctx->device));
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return ctx->fail(OLAP_ERR_NOMEM, "hipMalloc(%llu bytes) failed: %s", (unsigned long long)bytes,
                     hipGetErrorString(e));
  }
  *out_dptr = p;
  return OLAP_OK;
}

int olap_free(olap_ctx* ctx, void* dptr) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (!dptr) return OLAP_OK;
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  OLAP_HIP(ctx, hipFree(dptr));
  return OLAP_OK;
}

int olap_memcpy_h2d(olap_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return OLAP_ERR_INVALID;
  if (!bytes) return OLAP_OK;
  OLAP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return OLAP_OK;
}

int olap_memcpy_d2h(olap_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return OLAP_ERR_INVALID;
  if (!bytes) return OLAP_OK;
  OLAP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return OLAP_OK;
}

int olap_memset(olap_ctx* ctx, void* dst, int byte_value, uint64_t bytes) {
  if (!ctx || (bytes && !dst)) return OLAP_ERR_INVALID;
  if (!bytes) return OLAP_OK;
  OLAP_HIP(ctx, hipMemsetAsync(dst, byte_value, bytes, ctx->stream));
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return OLAP_OK;
}

const char* olap_phase_kernel(olap_ctx* ctx, const char* phase_name) {
  if (!ctx || !phase_name) return "";
  for (int p = 0; p < PH_COUNT; ++p)
    if (!std::strcmp(kPhaseNames[p], phase_name)) return ctx->phase_kernel[p].c_str();
  return "";
}

int olap_set_profiling(olap_ctx* ctx, int enabled) {
  if (!ctx) return OLAP_ERR_INVALID;
  ctx->profiling = enabled != 0;
  return OLAP_OK;
}

int olap_reset_timers(olap_ctx* ctx) {
  if (!ctx) return OLAP_ERR_INVALID;
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  resolve_timers(ctx);
  for (auto& t : ctx->timers) t = PhaseTimer{};
  return OLAP_OK;
}

int olap_get_timers(olap_ctx* ctx, const char** names, double* total_ms, uint64_t* launches, int cap,
                    int* out_n) {
  if (!ctx || !out_n) return OLAP_ERR_INVALID;
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  resolve_timers(ctx);
  int n = 0;
  for (int p = 0; p < PH_COUNT; ++p) {
    if (!ctx->timers[p].launches) continue;
    if (n < cap) {
      if (names) names[n] = kPhaseNames[p];
      if (total_ms) total_ms[n] = ctx->timers[p].ms;
      if (launches) launches[n] = ctx->timers[p].launches;
    }
    ++n;
  }
  *out_n = n;
  return OLAP_OK;
}

}  // extern "C"
