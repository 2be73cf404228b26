// copy_rate.hip -- what does a plain copy of 32 GiB -> 32 GiB reach on this chip, and with which shape?
// (the radix passes move exactly these bytes: the ceiling their 12.8-13.8 ms are measured against)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/copy_rate scripts/ubench/copy_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define G __attribute__((address_space(1)))
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// each workgroup copies a contiguous run of tiles (RUN=1) or tiles dealt round-robin (RUN=0); U vectors in flight per thread
template <int U, int NTL, int NTS, int RUN>
__global__ __launch_bounds__(1024) void k_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint64_t nvec) {
  const uint64_t tile = (uint64_t)blockDim.x * U;
  const uint64_t tiles = nvec / tile;
  const uint64_t per = (tiles + gridDim.x - 1) / gridDim.x;
  uint64_t t = RUN ? blockIdx.x * per : blockIdx.x;
  const uint64_t tend = RUN ? (t + per < tiles ? t + per : tiles) : tiles;
  const uint64_t step = RUN ? 1 : gridDim.x;
  for (; t < tend; t += step) {
    const G u32x4* p = (const G u32x4*)in + t * tile + threadIdx.x;
    G u32x4* q = (G u32x4*)out + t * tile + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = NTL ? __builtin_nontemporal_load(p + (uint64_t)j * blockDim.x) : p[(uint64_t)j * blockDim.x];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (NTS) __builtin_nontemporal_store(v[j], q + (uint64_t)j * blockDim.x);
      else q[(uint64_t)j * blockDim.x] = v[j];
    }
  }
}

template <int U, int NTL, int NTS, int RUN>
static float run(const u32x4* in, u32x4* out, uint64_t nvec, int grid, int block) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_copy<U, NTL, NTS, RUN>), dim3(grid), dim3(block), 0, 0, in, out, nvec);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep && ms < best) best = ms;
  }
  return best;
}

int main() {
  const uint64_t bytes = 32ull << 30, nvec = bytes / 16;
  u32x4 *in, *out;
  CHECK(hipMalloc(&in, bytes)); CHECK(hipMalloc(&out, bytes + (1 << 21)));
  CHECK(hipMemset(in, 1, bytes)); CHECK(hipMemset(out, 2, bytes));
  printf("copy of 32 GiB -> 32 GiB (best of 3 after a warm-up); GB/s counts read + written bytes\n");
  printf("%-44s %8s %8s\n", "variant", "ms", "GB/s");
  auto rep = [&](const char* name, float ms) { printf("%-44s %8.2f %8.0f\n", name, ms, 2.0 * bytes / ms / 1e6); };
  {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float best = 1e9f;
    for (int r = 0; r < 4; ++r) { hipEventRecord(a); hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (r && ms < best) best = ms; }
    rep("hipMemcpyAsync D2D", best);
  }
  char name[128];
  for (int grid : {256, 512, 1024, 2048, 8192}) {
    for (int block : {256, 512, 1024}) {
      if ((uint64_t)grid * block > 256ull * 2048 * 4 && grid > 2048) continue;
      snprintf(name, sizeof name, "grid %5d block %4d U4 contiguous runs", grid, block); rep(name, run<4, 0, 0, 1>(in, out, nvec, grid, block));
      snprintf(name, sizeof name, "grid %5d block %4d U4 round-robin tiles", grid, block); rep(name, run<4, 0, 0, 0>(in, out, nvec, grid, block));
    }
  }
  for (int grid : {512, 2048}) {
    snprintf(name, sizeof name, "grid %5d block  512 U8 round-robin", grid); rep(name, run<8, 0, 0, 0>(in, out, nvec, grid, 512));
    snprintf(name, sizeof name, "grid %5d block  512 U2 round-robin", grid); rep(name, run<2, 0, 0, 0>(in, out, nvec, grid, 512));
    snprintf(name, sizeof name, "grid %5d block  512 U4 rr nt-load", grid); rep(name, run<4, 1, 0, 0>(in, out, nvec, grid, 512));
    snprintf(name, sizeof name, "grid %5d block  512 U4 rr nt-store", grid); rep(name, run<4, 0, 1, 0>(in, out, nvec, grid, 512));
    snprintf(name, sizeof name, "grid %5d block  512 U4 rr nt-both", grid); rep(name, run<4, 1, 1, 0>(in, out, nvec, grid, 512));
  }
  // skewed destination (not a power of two away from the source)
  rep("grid 2048 block 512 U4 rr, out + 1 MiB + 4352 B", run<4, 0, 0, 0>(in, (u32x4*)((char*)out + (1 << 20) + 4352), nvec - (1 << 17), 2048, 512));
  return 0;
}
