// gen.hip -- device-side synthetic column generator and order-independent
// digests.  Distributions follow the reference's generator
// (host/generator/generator.cc:22-71: uniform u32 payloads, pk = 0,1,2,...,
// fk of left batch i uniform in [i*R, (i+1)*R-1]); the bit stream is our own
// counter-based one (common.h gen64/gen32, mirrored in dpu_olap_amd/datagen.py)
// so SF=2048 columns are produced directly in HBM and can be re-derived anywhere.
#include "common.h"

namespace olap {

constexpr int kGBlock = 256;

__global__ __launch_bounds__(kGBlock) void k_gen_uniform_u32(uint32_t* out, uint64_t n, uint64_t seed,
                                                             uint32_t stream, uint64_t first_row) {
  for (uint64_t i = (uint64_t)blockIdx.x * kGBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kGBlock)
    out[i] = gen32(seed, stream, first_row + i);
}

__global__ __launch_bounds__(kGBlock) void k_gen_uniform_u64(uint64_t* out, uint64_t n, uint64_t seed,
                                                             uint32_t stream, uint64_t first_row) {
  for (uint64_t i = (uint64_t)blockIdx.x * kGBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kGBlock)
    out[i] = gen64(seed, stream, first_row + i);
}

__global__ __launch_bounds__(kGBlock) void k_gen_index_u32(uint32_t* out, uint64_t n, uint32_t first) {
  for (uint64_t i = (uint64_t)blockIdx.x * kGBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kGBlock)
    out[i] = first + (uint32_t)i;
}

__global__ __launch_bounds__(kGBlock) void k_gen_fk_u32(uint32_t* out, uint64_t n, uint64_t seed, uint32_t stream,
                                                        uint64_t first_row, uint64_t lrows, uint64_t rrows) {
  for (uint64_t i = (uint64_t)blockIdx.x * kGBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kGBlock) {
    const uint64_t r = first_row + i;
    const uint64_t batch = r / lrows;
    const uint64_t u = ((uint64_t)gen32(seed, stream, r) * rrows) >> 32;  // uniform in [0, rrows)
    out[i] = (uint32_t)(batch * rrows + u);
  }
}

__global__ __launch_bounds__(kGBlock) void k_gen_uniform_u32_at(uint32_t* out, const uint32_t* rows, uint64_t n,
                                                                uint64_t seed, uint32_t stream) {
  for (uint64_t i = (uint64_t)blockIdx.x * kGBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kGBlock)
    out[i] = gen32(seed, stream, rows[i]);
}

__device__ __forceinline__ void block_fold3(uint64_t s, uint64_t x, uint64_t m, unsigned long long* out) {
  __shared__ unsigned long long sh[3][kGBlock / kWave];
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    s += __shfl_down(s, o, kWave);
    x ^= __shfl_down(x, o, kWave);
    m += __shfl_down(m, o, kWave);
  }
  if (lane_id() == 0) {
    sh[0][threadIdx.x / kWave] = s;
    sh[1][threadIdx.x / kWave] = x;
    sh[2][threadIdx.x / kWave] = m;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long S = 0, X = 0, M = 0;
    for (int i = 0; i < kGBlock / kWave; ++i) {
      S += sh[0][i];
      X ^= sh[1][i];
      M += sh[2][i];
    }
    atomicAdd(&out[0], S);
    atomicXor(&out[1], X);
    atomicAdd(&out[2], M);
  }
}

template <typename T>
__global__ __launch_bounds__(kGBlock) void k_checksum(const T* col, uint64_t n, unsigned long long* out) {
  uint64_t s = 0, x = 0, m = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kGBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kGBlock) {
    const uint64_t v = (uint64_t)col[i];
    s += v;
    x ^= v;
    m += mix64(v);
  }
  block_fold3(s, x, m, out);
}

__global__ __launch_bounds__(kGBlock) void k_checksum_rows3(const uint32_t* a, const uint32_t* b, const uint32_t* c,
                                                            uint64_t n, unsigned long long* out) {
  uint64_t s = 0, x = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kGBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kGBlock) {
    const uint64_t h = mix64(mix64(((uint64_t)a[i] << 32) | b[i]) ^ (uint64_t)c[i]);
    s += h;
    x ^= h;
  }
  block_fold3(s, x, 0, out);
}

static unsigned gen_grid(const olap_ctx* ctx, uint64_t n) {
  uint64_t g = (n + kGBlock - 1) / kGBlock;
  const uint64_t cap = (uint64_t)ctx->compute_units * 16;
  if (g > cap) g = cap;
  return (unsigned)(g ? g : 1);
}

static int gen_prolog(olap_ctx* ctx, const void* out, uint64_t n) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (n && !out) return ctx->fail(OLAP_ERR_INVALID, "null output");
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
  return OLAP_OK;
}

static int gen_epilog(olap_ctx* ctx) {
  OLAP_HIP(ctx, hipGetLastError());
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return resolve_timers(ctx);
}

}  // namespace olap

using namespace olap;

extern "C" {

int olap_gen_uniform_u32(olap_ctx* ctx, uint32_t* out, uint64_t n, uint64_t seed, uint32_t stream_id,
                         uint64_t first_row) {
  OLAP_TRY(gen_prolog(ctx, out, n));
  if (!n) return OLAP_OK;
  {
    PhaseScope ps(ctx, PH_GEN);
    hipLaunchKernelGGL(k_gen_uniform_u32, dim3(gen_grid(ctx, n)), dim3(kGBlock), 0, ctx->stream, out, n, seed,
                       stream_id, first_row);
  }
  return gen_epilog(ctx);
}

int olap_gen_uniform_u64(olap_ctx* ctx, uint64_t* out, uint64_t n, uint64_t seed, uint32_t stream_id,
                         uint64_t first_row) {
  OLAP_TRY(gen_prolog(ctx, out, n));
  if (!n) return OLAP_OK;
  {
    PhaseScope ps(ctx, PH_GEN);
    hipLaunchKernelGGL(k_gen_uniform_u64, dim3(gen_grid(ctx, n)), dim3(kGBlock), 0, ctx->stream, out, n, seed,
                       stream_id, first_row);
  }
  return gen_epilog(ctx);
}

int olap_gen_index_u32(olap_ctx* ctx, uint32_t* out, uint64_t n, uint32_t first_value) {
  OLAP_TRY(gen_prolog(ctx, out, n));
  if (!n) return OLAP_OK;
  {
    PhaseScope ps(ctx, PH_GEN);
    hipLaunchKernelGGL(k_gen_index_u32, dim3(gen_grid(ctx, n)), dim3(kGBlock), 0, ctx->stream, out, n, first_value);
  }
  return gen_epilog(ctx);
}

int olap_gen_fk_u32(olap_ctx* ctx, uint32_t* out, uint64_t n, uint64_t seed, uint32_t stream_id, uint64_t first_row,
                    uint64_t left_batch_rows, uint64_t right_batch_rows) {
  OLAP_TRY(gen_prolog(ctx, out, n));
  if (!left_batch_rows || !right_batch_rows || right_batch_rows > (1ull << 32))
    return ctx->fail(OLAP_ERR_INVALID, "bad batch rows");
  if (!n) return OLAP_OK;
  {
    PhaseScope ps(ctx, PH_GEN);
    hipLaunchKernelGGL(k_gen_fk_u32, dim3(gen_grid(ctx, n)), dim3(kGBlock), 0, ctx->stream, out, n, seed, stream_id,
                       first_row, left_batch_rows, right_batch_rows);
  }
  return gen_epilog(ctx);
}

int olap_gen_uniform_u32_at(olap_ctx* ctx, uint32_t* out, const uint32_t* rows_device, uint64_t n, uint64_t seed,
                            uint32_t stream_id) {
  OLAP_TRY(gen_prolog(ctx, out, n));
  if (n && !rows_device) return ctx->fail(OLAP_ERR_INVALID, "null rows");
  if (!n) return OLAP_OK;
  {
    PhaseScope ps(ctx, PH_GEN);
    hipLaunchKernelGGL(k_gen_uniform_u32_at, dim3(gen_grid(ctx, n)), dim3(kGBlock), 0, ctx->stream, out, rows_device,
                       n, seed, stream_id);
  }
  return gen_epilog(ctx);
}

int olap_checksum(olap_ctx* ctx, const void* col, uint64_t n, int elem_bytes, uint64_t out[3]) {
  OLAP_TRY(gen_prolog(ctx, col, n));
  if (!out || (elem_bytes != 4 && elem_bytes != 8)) return ctx->fail(OLAP_ERR_INVALID, "bad checksum arguments");
  OLAP_TRY(ws_begin(ctx, 4096));
  unsigned long long* d = (unsigned long long*)ws_alloc(ctx, 24);
  if (!d) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted");
  OLAP_HIP(ctx, hipMemsetAsync(d, 0, 24, ctx->stream));
  if (n) {
    PhaseScope ps(ctx, PH_GEN);
    if (elem_bytes == 4)
      hipLaunchKernelGGL((k_checksum<uint32_t>), dim3(gen_grid(ctx, n)), dim3(kGBlock), 0, ctx->stream,
                         (const uint32_t*)col, n, d);
    else
      hipLaunchKernelGGL((k_checksum<uint64_t>), dim3(gen_grid(ctx, n)), dim3(kGBlock), 0, ctx->stream,
                         (const uint64_t*)col, n, d);
    OLAP_HIP(ctx, hipGetLastError());
  }
  unsigned long long h[3];
  OLAP_HIP(ctx, hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, ctx->stream));
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  out[0] = h[0];
  out[1] = h[1];
  out[2] = h[2];
  return resolve_timers(ctx);
}

int olap_checksum_rows3_u32(olap_ctx* ctx, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint64_t n,
                            uint64_t out[2]) {
  OLAP_TRY(gen_prolog(ctx, a, n));
  if (!out || (n && (!b || !c))) return ctx->fail(OLAP_ERR_INVALID, "bad checksum arguments");
  OLAP_TRY(ws_begin(ctx, 4096));
  unsigned long long* d = (unsigned long long*)ws_alloc(ctx, 24);
  if (!d) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted");
  OLAP_HIP(ctx, hipMemsetAsync(d, 0, 24, ctx->stream));
  if (n) {
    PhaseScope ps(ctx, PH_GEN);
    hipLaunchKernelGGL(k_checksum_rows3, dim3(gen_grid(ctx, n)), dim3(kGBlock), 0, ctx->stream, a, b, c, n, d);
    OLAP_HIP(ctx, hipGetLastError());
  }
  unsigned long long h[3];
  OLAP_HIP(ctx, hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, ctx->stream));
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  out[0] = h[0];
  out[1] = h[1];
  return resolve_timers(ctx);
}

}  // extern "C"
