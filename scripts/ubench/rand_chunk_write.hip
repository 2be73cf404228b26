// Microbenchmark: HBM write bandwidth for scattered chunks of S bytes (what a radix
// scatter produces) vs chunk size, with 8-byte and 16-byte stores per lane, and with a
// streaming read running beside it (read 8 B + write 8 B per element = scatter mix).
// Build: hipcc --offload-arch=gfx950 -O3 -o rand_chunk_write rand_chunk_write.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// each thread moves 8 bytes (one tuple); lanes_per_chunk = S / 8
template <bool READ>
__global__ __launch_bounds__(256) void k_w8(uint2* dst, const uint2* src, uint64_t n_elems, uint32_t lanes_per_chunk,
                                            uint64_t n_chunks_mask, uint64_t salt) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_elems; i += (uint64_t)gridDim.x * 256) {
    const uint64_t chunk = i / lanes_per_chunk, within = i % lanes_per_chunk;
    const uint64_t pos = mix64(chunk ^ salt) & n_chunks_mask;
    uint2 v = make_uint2((uint32_t)i, (uint32_t)(i >> 32));
    if (READ) v = src[i];
    dst[pos * lanes_per_chunk + within] = v;
  }
}

template <bool READ>
__global__ __launch_bounds__(256) void k_w16(uint4* dst, const uint4* src, uint64_t n_elems, uint32_t lanes_per_chunk,
                                             uint64_t n_chunks_mask, uint64_t salt) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_elems; i += (uint64_t)gridDim.x * 256) {
    const uint64_t chunk = i / lanes_per_chunk, within = i % lanes_per_chunk;
    const uint64_t pos = mix64(chunk ^ salt) & n_chunks_mask;
    uint4 v = make_uint4((uint32_t)i, (uint32_t)(i >> 32), 1, 2);
    if (READ) v = src[i];
    dst[pos * lanes_per_chunk + within] = v;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const uint64_t bytes = 16ull << 30;
  void *dst, *src;
  CK(hipMalloc(&dst, bytes));
  CK(hipMalloc(&src, bytes));
  CK(hipMemset(dst, 0, bytes));
  CK(hipMemset(src, 1, bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const int grid = 256 * 8;
  printf("%-6s %-6s %-10s %-10s\n", "store", "chunkB", "write GB/s", "rd+wr GB/s(total)");
  for (int width : {8, 16}) {
    for (uint32_t S : {64u, 128u, 256u, 512u, 1024u, 4096u, 65536u}) {
      if (S < (uint32_t)width) continue;
      const uint32_t lpc = S / width;
      const uint64_t n_elems = bytes / width;
      const uint64_t mask = bytes / S - 1;
      float ms[2] = {0, 0};
      for (int rd = 0; rd < 2; ++rd) {
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(a));
          if (width == 8) {
            if (rd) hipLaunchKernelGGL(k_w8<true>, dim3(grid), dim3(256), 0, 0, (uint2*)dst, (const uint2*)src, n_elems, lpc, mask, 77ull + rep);
            else hipLaunchKernelGGL(k_w8<false>, dim3(grid), dim3(256), 0, 0, (uint2*)dst, (const uint2*)src, n_elems, lpc, mask, 77ull + rep);
          } else {
            if (rd) hipLaunchKernelGGL(k_w16<true>, dim3(grid), dim3(256), 0, 0, (uint4*)dst, (const uint4*)src, n_elems, lpc, mask, 77ull + rep);
            else hipLaunchKernelGGL(k_w16<false>, dim3(grid), dim3(256), 0, 0, (uint4*)dst, (const uint4*)src, n_elems, lpc, mask, 77ull + rep);
          }
          CK(hipEventRecord(b));
          CK(hipEventSynchronize(b));
          float t;
          CK(hipEventElapsedTime(&t, a, b));
          if (rep == 0 || t < ms[rd]) ms[rd] = t;
        }
      }
      printf("%-6d %-6u %-10.0f %-10.0f\n", width, S, bytes / (ms[0] * 1e-3) / 1e9, 2.0 * bytes / (ms[1] * 1e-3) / 1e9);
      fflush(stdout);
    }
  }
  return 0;
}
