// Microbenchmark: does it matter WHERE the 512 output streams of a radix pass lie?
// Every workgroup (512 threads, 2 per CU, 512 workgroups) reads its contiguous slice of the
// input (16 B per lane) and writes it out as LINE-byte lines dealt round-robin to K = 512
// streams -- the memory behaviour of the write-combining scatter without its LDS work.
//   near        stream s of workgroup w starts at w * WINDOW + s * (WINDOW / K)      (pass 1: private window)
//   far_exact   stream s of workgroup w starts at s * BUCKET + w * (BUCKET / nWG)    (exact pass 0)
//   far_claim   bucket s is shared: 1 KiB allotments interleaved over the workgroups  (claiming pass 0)
// and the reverse for reads (gather): pieces of P bytes fetched from a different window
// each time, written sequentially.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_fanout stream_fanout.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int kBlock = 512, kWG = 512, K = 512;

enum { NEAR = 0, FAR_EXACT = 1, FAR_CLAIM = 2 };

// lanes_per_line = LINE / 16; one iteration of a workgroup moves kBlock * 16 B = 8 KiB
__global__ __launch_bounds__(kBlock) void k_fanout(const uint4* in, uint4* out, uint64_t bytes_per_wg, int mode,
                                                    uint32_t line_bytes, uint64_t pad) {
  const uint32_t lpl = line_bytes / 16;                       // lanes per line
  const uint32_t lines_per_iter = kBlock / lpl;               // lines one iteration writes
  const uint64_t iters = bytes_per_wg / (kBlock * 16);
  const uint64_t window = bytes_per_wg, bucket = bytes_per_wg;  // K == kWG: a bucket is as large as a window
  const uint4* src = in + (uint64_t)blockIdx.x * (bytes_per_wg / 16);
  const uint32_t l = threadIdx.x / lpl, within = threadIdx.x % lpl;
  for (uint64_t it = 0; it < iters; ++it) {
    const uint4 v = src[it * kBlock + threadIdx.x];
    const uint64_t line = it * lines_per_iter + l;            // running line number of this workgroup
    const uint64_t s = line % K, n = line / K;                // stream, n-th line of that stream
    uint64_t off;
    // pad: extra bytes between consecutive streams, so their addresses do not differ by a power of two
    if (mode == NEAR) off = (uint64_t)blockIdx.x * (window + K * pad) + s * (window / K + pad) + n * line_bytes;
    else if (mode == FAR_EXACT) off = s * (bucket + pad) + (uint64_t)blockIdx.x * (bucket / kWG) + n * line_bytes;
    else {
      const uint64_t lines_per_allot = 1024 / line_bytes;
      off = s * (bucket + pad) + ((n / lines_per_allot) * kWG + blockIdx.x) * 1024 + (n % lines_per_allot) * line_bytes;
    }
    out[off / 16 + within] = v;
  }
}

// gather: piece j of workgroup w comes from window (w + j) % kWG, slot j / kWG ... of that window
__global__ __launch_bounds__(kBlock) void k_gather(const uint4* in, uint4* out, uint64_t bytes_per_wg, uint32_t piece_bytes,
                                                    int far) {
  const uint64_t iters = bytes_per_wg / (kBlock * 16);
  const uint64_t pieces_per_wg = bytes_per_wg / piece_bytes;
  uint4* dst = out + (uint64_t)blockIdx.x * (bytes_per_wg / 16);
  for (uint64_t it = 0; it < iters; ++it) {
    const uint64_t byte = it * kBlock * 16 + threadIdx.x * 16;
    const uint64_t piece = byte / piece_bytes, within = byte % piece_bytes;
    uint64_t off;
    if (far) {
      const uint64_t win = (blockIdx.x + piece) % kWG;                    // another window every piece
      const uint64_t slot = (piece / kWG) * kWG + blockIdx.x;             // distinct slots per reader
      off = win * bytes_per_wg + (slot % pieces_per_wg) * piece_bytes + within;
    } else {
      off = (uint64_t)blockIdx.x * bytes_per_wg + byte;
    }
    dst[it * kBlock + threadIdx.x] = in[off / 16];
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const uint64_t bytes = 32ull << 30, per_wg = bytes / kWG;
  void *src, *dst;
  CK(hipMalloc(&src, bytes));
  const uint64_t dst_bytes = bytes + (4ull << 30);  // room for the padded layouts
  CK(hipMalloc(&dst, dst_bytes));
  CK(hipMemset(src, 1, bytes));
  CK(hipMemset(dst, 0, dst_bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  auto timed = [&](auto&& launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(a);
      launch();
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float t;
      (void)hipEventElapsedTime(&t, a, b);
      if (t < best) best = t;
    }
    return best;
  };
  const char* names[3] = {"near", "far_exact", "far_claim"};
  printf("fan-out writes (32 GiB in, 32 GiB out, 512 streams per workgroup)\n%-10s %-8s %-6s %-8s %-8s\n", "layout", "pad", "line", "ms", "GB/s(rd+wr)");
  for (int mode = 0; mode < 3; ++mode)
    for (uint64_t pad : {0ull, 4352ull, 69632ull, 1114112ull}) {   // 17 x 256 B, 17 x 4 KiB, 17 x 64 KiB
      if (mode == NEAR && pad > 4352) continue;                  // (K * pad per window must fit the spare room)
      for (uint32_t line : {64u, 128u}) {
        const float ms = timed([&] { hipLaunchKernelGGL(k_fanout, dim3(kWG), dim3(kBlock), 0, 0, (const uint4*)src, (uint4*)dst, per_wg, mode, line, pad); });
        printf("%-10s %-8llu %-6u %-8.2f %-8.0f\n", names[mode], (unsigned long long)pad, line, ms, 2.0 * bytes / (ms * 1e-3) / 1e9);
      }
    }
  printf("gather reads (pieces from another window each time, sequential writes)\n%-10s %-8s %-8s %-8s\n", "layout", "piece", "ms", "GB/s(rd+wr)");
  {
    const float ms = timed([&] { hipLaunchKernelGGL(k_gather, dim3(kWG), dim3(kBlock), 0, 0, (const uint4*)src, (uint4*)dst, per_wg, 8192u, 0); });
    printf("%-10s %-8s %-8.2f %-8.0f\n", "near", "-", ms, 2.0 * bytes / (ms * 1e-3) / 1e9);
  }
  for (uint32_t piece : {1024u, 8192u, 32768u, 131072u}) {
    const float ms = timed([&] { hipLaunchKernelGGL(k_gather, dim3(kWG), dim3(kBlock), 0, 0, (const uint4*)src, (uint4*)dst, per_wg, piece, 1); });
    printf("%-10s %-8u %-8.2f %-8.0f\n", "far", piece, ms, 2.0 * bytes / (ms * 1e-3) / 1e9);
  }
  return 0;
}
