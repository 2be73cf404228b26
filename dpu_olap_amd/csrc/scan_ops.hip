// scan_ops.hip -- sum, filter (order-preserving stream compaction) and take
// (gather) over batched Arrow integer columns, hand-written for gfx950.
//
// Reference behaviour (upmem/dpu_olap):
//   sum    dpu/shared/kernels/aggr.c:16-33 + dpu/aggr/main.c:44-50,76-88
//   filter dpu/shared/kernels/filter.c:25,57-177
//   take   dpu/shared/kernels/take.c:12-47
// All three are HBM-bandwidth bound: 16-byte-per-lane coalesced loads, wave64
// ballot / popcount ranks for the compaction, one u64 atomic per workgroup for
// the sum.  No MFMA.
#include "common.h"

namespace olap {

constexpr int kBlock = 256;          // 4 waves
constexpr int kVecPerThread = 4;     // 16-B vectors per thread per tile
constexpr int kTileBytes = kBlock * kVecPerThread * 16;  // 16 KiB per tile

template <typename T>
struct Vec16;
template <>
struct Vec16<uint32_t> { using type = uint4; };
template <>
struct Vec16<int32_t> { using type = int4; };
template <>
struct Vec16<uint64_t> { using type = ulonglong2; };
template <>
struct Vec16<int64_t> { using type = longlong2; };

template <typename T>
__device__ __forceinline__ T vec_get(const typename Vec16<T>::type& v, int k);
template <>
__device__ __forceinline__ uint32_t vec_get<uint32_t>(const uint4& v, int k) {
  return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
}
template <>
__device__ __forceinline__ int32_t vec_get<int32_t>(const int4& v, int k) {
  return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
}
template <>
__device__ __forceinline__ uint64_t vec_get<uint64_t>(const ulonglong2& v, int k) {
  return k == 0 ? v.x : v.y;
}
template <>
__device__ __forceinline__ int64_t vec_get<int64_t>(const longlong2& v, int k) {
  return k == 0 ? v.x : v.y;
}

// ------------------------------------------------------------------ sum
// One tile = 16 KiB of one batch.  Grid-stride over tiles; per-lane 64-bit
// accumulator; wave shuffle reduce; LDS across the 4 waves; one atomic per block.
constexpr int kSumVec = 8;  // 16-B vectors in flight per thread in k_sum (32 KiB tiles)

template <typename T, typename ACC>
__global__ __launch_bounds__(kBlock) void k_sum(ColView v, const void* const* ptrs,
                                                unsigned long long* out) {
  constexpr int E = 16 / sizeof(T);
  constexpr int kVecPerThread = kSumVec;
  constexpr uint32_t TILE = kBlock * kVecPerThread * E;
  using V = typename Vec16<T>::type;
  ACC acc = 0;
  for (uint64_t tile = blockIdx.x; tile < v.total_units; tile += gridDim.x) {
    Loc l = locate(v, tile);
    const uint64_t rows = batch_rows(v, l.batch);
    const uint64_t r0 = l.local_unit * TILE;
    const uint32_t n = (uint32_t)min((uint64_t)TILE, rows - r0);
    const T* p = (const T*)ptrs[l.batch] + r0;
    if (n == TILE && (((uintptr_t)p) & 15) == 0) {
      const V* pv = (const V*)p;
      V x[kVecPerThread];
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j) x[j] = gload_vec16(pv + j * kBlock + threadIdx.x);
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j)
#pragma unroll
        for (int k = 0; k < E; ++k) acc += (ACC)vec_get<T>(x[j], k);
    } else {
      // a batch's last (or only) tile, or a buffer that is only element-aligned: 16-byte loads over the aligned
      // middle, scalars at both ends (batches shorter than a tile are all "last tiles")
      const uint32_t head = min(n, (uint32_t)(((16 - (((uintptr_t)p) & 15)) & 15) / sizeof(T)));
      const uint32_t nvec = (n - head) / E;
      const V* pv = (const V*)(p + head);
      for (uint32_t i = threadIdx.x; i < head; i += kBlock) acc += (ACC)gload_t(p + i);
      for (uint32_t i = threadIdx.x; i < nvec; i += kBlock) {
        const V x = gload_vec16(pv + i);
#pragma unroll
        for (int k = 0; k < E; ++k) acc += (ACC)vec_get<T>(x, k);
      }
      for (uint32_t i = head + nvec * E + threadIdx.x; i < n; i += kBlock) acc += (ACC)gload_t(p + i);
    }
  }
  __shared__ unsigned long long s_part[kBlock / kWave];
  unsigned long long w = wave_reduce_add((unsigned long long)acc);
  if (lane_id() == 0) s_part[threadIdx.x / kWave] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int i = 0; i < kBlock / kWave; ++i) t += s_part[i];
    atomicAdd(out, t);
  }
}

// --------------------------------------------------------------- filter
// Work unit = segment: up to seg_rows consecutive rows of one batch, processed
// by ONE workgroup tile after tile with a running output offset, so a batch
// that is a single segment (the reference's 64 Ki-row batches) needs no
// inter-workgroup communication at all.  Batches longer than a segment use a
// count pass + per-batch scan of the segment counts first.
//
// Rank of an element inside a tile (order preserving):
//   round j (kBlock vectors = kBlock*E consecutive rows), wave w, lane l:
//   c = number of passing elements among the lane's E elements (0..E);
//   exclusive lane prefix from three ballots (bit planes of c) + popcount of the
//   lanes below; wave totals of all rounds go through LDS once per tile.
template <typename T>
__device__ __forceinline__ void filter_tile_flags(const T* p, uint32_t n, uint32_t TILE, bool fast,
                                                  T thr, T (&val)[kVecPerThread][16 / sizeof(T)],
                                                  uint32_t (&flags)[kVecPerThread]) {
  constexpr int E = 16 / sizeof(T);
  using V = typename Vec16<T>::type;
  if (fast) {
    const V* pv = (const V*)p;
    V x[kVecPerThread];
#pragma unroll
    for (int j = 0; j < kVecPerThread; ++j) x[j] = gload_vec16(pv + j * kBlock + threadIdx.x);
#pragma unroll
    for (int j = 0; j < kVecPerThread; ++j) {
      uint32_t f = 0;
#pragma unroll
      for (int k = 0; k < E; ++k) {
        val[j][k] = vec_get<T>(x[j], k);
        f |= (val[j][k] < thr ? 1u : 0u) << k;
      }
      flags[j] = f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < kVecPerThread; ++j) {
      uint32_t f = 0;
#pragma unroll
      for (int k = 0; k < E; ++k) {
        uint32_t idx = (j * kBlock + threadIdx.x) * E + k;
        T x = idx < n ? gload_t(p + idx) : thr;  // thr itself never passes `x < thr`
        val[j][k] = x;
        f |= ((idx < n && x < thr) ? 1u : 0u) << k;
      }
      flags[j] = f;
    }
  }
  (void)TILE;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_filter_count(ColView sv, const void* const* in_ptrs, T thr,
                                                         uint32_t* seg_count) {
  constexpr int E = 16 / sizeof(T);
  constexpr uint32_t TILE = kBlock * kVecPerThread * E;
  __shared__ uint32_t s_part[kBlock / kWave];
  for (uint64_t seg = blockIdx.x; seg < sv.total_units; seg += gridDim.x) {
    Loc l = locate(sv, seg);
    const uint64_t rows = batch_rows(sv, l.batch);
    const uint64_t r0 = l.local_unit * sv.unit_rows;
    const uint32_t n_seg = (uint32_t)min((uint64_t)sv.unit_rows, rows - r0);
    const T* in = (const T*)in_ptrs[l.batch] + r0;
    // a buffer that is only element-aligned (an Arrow slice): a short first tile up to the first 16-byte boundary,
    // whole aligned tiles after it
    const uint32_t head = min(n_seg, (uint32_t)(((16 - (((uintptr_t)in) & 15)) & 15) / sizeof(T)));
    uint32_t cnt = 0;
    for (uint32_t t0 = 0, step = head ? head : TILE; t0 < n_seg; t0 += step, step = TILE) {
      const uint32_t n = min(step, n_seg - t0);
      T val[kVecPerThread][E];
      uint32_t flags[kVecPerThread];
      filter_tile_flags<T>(in + t0, n, TILE, n == TILE, thr, val, flags);
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j) cnt += __popc(flags[j]);
    }
    uint32_t w = (uint32_t)wave_reduce_add((uint64_t)cnt);
    __syncthreads();  // s_part reuse across segments
    if (lane_id() == 0) s_part[threadIdx.x / kWave] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t t = 0;
#pragma unroll
      for (int i = 0; i < kBlock / kWave; ++i) t += s_part[i];
      seg_count[seg] = t;
    }
  }
}

// One workgroup per batch: exclusive scan of that batch's segment counts.
__global__ __launch_bounds__(kBlock) void k_filter_segscan(ColView sv, const uint32_t* seg_count,
                                                           uint64_t* seg_base) {
  __shared__ uint64_t s_wave[kBlock / kWave];
  __shared__ uint64_t s_carry;
  for (uint32_t b = blockIdx.x; b < sv.nb; b += gridDim.x) {
    const uint64_t s0 = sv.units_per_batch ? (uint64_t)b * sv.units_per_batch : sv.unit_start[b];
    const uint64_t s1 = sv.units_per_batch ? s0 + sv.units_per_batch : sv.unit_start[b + 1];
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t base = s0; base < s1; base += kBlock) {
      const uint64_t i = base + threadIdx.x;
      uint64_t c = i < s1 ? seg_count[i] : 0;
      uint64_t incl = c;  // inclusive wave scan
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        uint64_t y = __shfl_up(incl, o, kWave);
        if ((int)lane_id() >= o) incl += y;
      }
      if (lane_id() == kWave - 1) s_wave[threadIdx.x / kWave] = incl;
      __syncthreads();
      uint64_t off = s_carry;
      for (unsigned w = 0; w < threadIdx.x / kWave; ++w) off += s_wave[w];
      if (i < s1) seg_base[i] = off + incl - c;
      __syncthreads();
      if (threadIdx.x == kBlock - 1) s_carry = off + incl;
      __syncthreads();
    }
  }
}

#ifndef OLAP_FILTER_WAVES
#define OLAP_FILTER_WAVES 1  // (A/B builds: minimum waves per SIMD of the compacting kernels -- 82 VGPRs = 5 waves as the compiler leaves it)
#endif
template <typename T>
__global__ __launch_bounds__(kBlock, OLAP_FILTER_WAVES) void k_filter_compact(ColView sv, const void* const* in_ptrs,
                                                           void* const* out_ptrs, T thr,
                                                           const uint64_t* seg_base,
                                                           unsigned long long* out_rows) {
  constexpr int E = 16 / sizeof(T);
  constexpr uint32_t TILE = kBlock * kVecPerThread * E;
  constexpr int NW = kBlock / kWave;
  __shared__ uint32_t s_tot[2][kVecPerThread][NW];
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x / kWave;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int buf = 0;
  for (uint64_t seg = blockIdx.x; seg < sv.total_units; seg += gridDim.x) {
    Loc l = locate(sv, seg);
    const uint64_t rows = batch_rows(sv, l.batch);
    const uint64_t r0 = l.local_unit * sv.unit_rows;
    const uint32_t n_seg = (uint32_t)min((uint64_t)sv.unit_rows, rows - r0);
    const T* in = (const T*)in_ptrs[l.batch] + r0;
    T* out = (T*)out_ptrs[l.batch];
    const uint32_t head = min(n_seg, (uint32_t)(((16 - (((uintptr_t)in) & 15)) & 15) / sizeof(T)));  // (as in k_filter_count)
    uint64_t base = seg_base ? seg_base[seg] : 0;
    for (uint32_t t0 = 0, step = head ? head : TILE; t0 < n_seg; t0 += step, step = TILE) {
      const uint32_t n = min(step, n_seg - t0);
      T val[kVecPerThread][E];
      uint32_t flags[kVecPerThread];
      filter_tile_flags<T>(in + t0, n, TILE, n == TILE, thr, val, flags);
      uint32_t rank[kVecPerThread];
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j) {
        const uint32_t c = __popc(flags[j]);
        uint32_t r = 0, tot = 0;
#pragma unroll
        for (int bit = 0; bit < 3; ++bit) {  // c <= 4 -> three bit planes
          const unsigned long long m = __ballot((c >> bit) & 1u);
          r += (uint32_t)__popcll(m & lt_mask) << bit;
          tot += (uint32_t)__popcll(m) << bit;
        }
        rank[j] = r;
        if (lane == 0) s_tot[buf][j][wave] = tot;
      }
      __syncthreads();
      uint32_t tile_total = 0;
      uint32_t my_off[kVecPerThread];
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          if ((unsigned)w == wave) my_off[j] = tile_total;
          tile_total += s_tot[buf][j][w];
        }
      }
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j) {
        uint64_t o = base + my_off[j] + rank[j];
#pragma unroll
        for (int k = 0; k < E; ++k) {
          if (flags[j] & (1u << k)) gstore_t(out + o++, val[j][k]);
        }
      }
      base += tile_total;
      buf ^= 1;  // double-buffered totals: one barrier per tile
    }
    const bool last_seg = (l.local_unit + 1) * (uint64_t)sv.unit_rows >= rows;
    if (last_seg && threadIdx.x == 0) out_rows[l.batch] = base;
  }
}

// k_filter_compact with the survivors of DENSE tiles (at least half of the rows pass) staged through LDS: they are written, in
// order, into a 16 KiB strip (ds_write_b32 / b64 per survivor instead of one predicated 4- or 8-byte global store each) and
// leave as whole 16-byte vectors, lane-contiguous (2^30 rows, all passing: 1.84 -> 1.67 ms; int32 at threshold 2^30 = 75 %:
// 23.0 -> 21.0 ms per 2^34 rows).  Sparse tiles keep the direct stores and their single barrier (the strip costs a second one:
// at 25 % it gains nothing, below that it loses 2-3 %).  The output position is kept as (16-byte vector index, elements already in that vector): what does
// not fill a vector is carried into the next tile, so every vector is written exactly once; only the first vector of a segment
// that starts inside a vector (its front belongs to the previous segment) and the last, partial one use scalar stores.  Two
// barriers per tile.  Needs 16-byte aligned output buffers (else: k_filter_compact).
template <typename T>
__global__ __launch_bounds__(kBlock, OLAP_FILTER_WAVES) void k_filter_compact_lds(ColView sv, const void* const* in_ptrs, void* const* out_ptrs, T thr,
                                                               const uint64_t* seg_base, unsigned long long* out_rows) {
  constexpr int E = 16 / sizeof(T);
  constexpr uint32_t TILE = kBlock * kVecPerThread * E;
  constexpr int NW = kBlock / kWave;
  using V = typename Vec16<T>::type;
  __shared__ uint32_t s_tot[2][kVecPerThread][NW];
  __shared__ __attribute__((aligned(16))) T s_stage[TILE + E];
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x / kWave;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int buf = 0;
  for (uint64_t seg = blockIdx.x; seg < sv.total_units; seg += gridDim.x) {
    Loc l = locate(sv, seg);
    const uint64_t rows = batch_rows(sv, l.batch);
    const uint64_t r0 = l.local_unit * sv.unit_rows;
    const uint32_t n_seg = (uint32_t)min((uint64_t)sv.unit_rows, rows - r0);
    const T* in = (const T*)in_ptrs[l.batch] + r0;
    T* out = (T*)out_ptrs[l.batch];
    const uint32_t head = min(n_seg, (uint32_t)(((16 - (((uintptr_t)in) & 15)) & 15) / sizeof(T)));  // (as in k_filter_count)
    const uint64_t base0 = seg_base ? seg_base[seg] : 0;
    uint64_t vbase = base0 / E * E;           // first element of the vector the next output element falls into
    uint32_t c = (uint32_t)(base0 - vbase);   // elements of that vector in front of the next output element ...
    uint32_t foreign = c;                     // ... of which this many are another segment's (never written from here)
    T carry[E];                               // thread 0: the elements [foreign, c) of the open vector
#pragma unroll
    for (int k = 0; k < E; ++k) carry[k] = T(0);
    for (uint32_t t0 = 0, step = head ? head : TILE; t0 < n_seg; t0 += step, step = TILE) {
      const uint32_t n = min(step, n_seg - t0);
      T val[kVecPerThread][E];
      uint32_t flags[kVecPerThread];
      filter_tile_flags<T>(in + t0, n, TILE, n == TILE, thr, val, flags);
      uint32_t rank[kVecPerThread];
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j) {
        const uint32_t cc = __popc(flags[j]);
        uint32_t r = 0, tot = 0;
#pragma unroll
        for (int bit = 0; bit < 3; ++bit) {  // cc <= 4 -> three bit planes
          const unsigned long long m = __ballot((cc >> bit) & 1u);
          r += (uint32_t)__popcll(m & lt_mask) << bit;
          tot += (uint32_t)__popcll(m) << bit;
        }
        rank[j] = r;
        if (lane == 0) s_tot[buf][j][wave] = tot;
      }
      __syncthreads();  // ---- totals known; the previous tile's strip has been read
      uint32_t tile_total = 0;
      uint32_t my_off[kVecPerThread];
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          if ((unsigned)w == wave) my_off[j] = tile_total;
          tile_total += s_tot[buf][j][w];
        }
      }
      if (tile_total * 2u < n) {
        // Few survivors (selectivity below one half): every survivor its own store, as k_filter_compact -- one barrier for this tile.
        // The open vector's elements go out first; what this tile writes into the front of the next open vector is then "not ours".
        if (threadIdx.x == 0) {
#pragma unroll
          for (int k = 0; k < E; ++k)
            if ((uint32_t)k >= foreign && (uint32_t)k < c) gstore_t(out + vbase + k, carry[k]);
        }
        const uint64_t base = vbase + c;
#pragma unroll
        for (int j = 0; j < kVecPerThread; ++j) {
          uint64_t o = base + my_off[j] + rank[j];
#pragma unroll
          for (int k = 0; k < E; ++k)
            if (flags[j] & (1u << k)) gstore_t(out + o++, val[j][k]);
        }
        const uint64_t nb = base + tile_total;
        vbase = nb / E * E;
        c = foreign = (uint32_t)(nb - vbase);
        buf ^= 1;
        continue;
      }
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j) {
        uint32_t o = c + my_off[j] + rank[j];
#pragma unroll
        for (int k = 0; k < E; ++k)
          if (flags[j] & (1u << k)) s_stage[o++] = val[j][k];
      }
      if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < E; ++k)
          if ((uint32_t)k >= foreign && (uint32_t)k < c) s_stage[k] = carry[k];
      }
      __syncthreads();  // ---- the strip holds [foreign, c + tile_total)
      const uint32_t filled = c + tile_total, nvec = filled / E, rem = filled - nvec * E;
      for (uint32_t v = threadIdx.x; v < nvec; v += kBlock) {
        const V x = *(const V*)(s_stage + v * E);
        if (v == 0 && foreign) {  // this vector's front was written before (another segment's, or a sparse tile's)
#pragma unroll
          for (int k = 0; k < E; ++k)
            if ((uint32_t)k >= foreign) gstore_t(out + vbase + k, vec_get<T>(x, k));
        } else {
          gstore_vec16((V*)(out + vbase) + v, x);
        }
      }
      if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < E; ++k)
          if ((uint32_t)k < rem) carry[k] = s_stage[nvec * E + k];
      }
      if (nvec) foreign = 0;
      vbase += (uint64_t)nvec * E;
      c = rem;
      buf ^= 1;
    }
    if (threadIdx.x == 0) {  // the open vector's elements
#pragma unroll
      for (int k = 0; k < E; ++k)
        if ((uint32_t)k >= foreign && (uint32_t)k < c) gstore_t(out + vbase + k, carry[k]);
    }
    const bool last_seg = (l.local_unit + 1) * (uint64_t)sv.unit_rows >= rows;
    if (last_seg && threadIdx.x == 0) out_rows[l.batch] = vbase + c;
  }
}

// ----------------------------------------------------------------- take
// Tiles over the index stream of each batch; indices loaded 16 B per lane,
// values gathered (the batch's data buffer is L2 / Infinity-Cache resident while
// its tiles are in flight), results stored 16 B per lane.
template <typename T, int VPT>
__global__ __launch_bounds__(kBlock) void k_take(ColView iv, const void* const* data_ptrs,
                                                 const void* const* idx_ptrs, void* const* out_ptrs) {
  constexpr int kVecPerThread = VPT;
  constexpr uint32_t TILE = kBlock * kVecPerThread * 4;  // indices per tile
  for (uint64_t tile = blockIdx.x; tile < iv.total_units; tile += gridDim.x) {
    Loc l = locate(iv, tile);
    const uint64_t rows = batch_rows(iv, l.batch);
    const uint64_t r0 = l.local_unit * TILE;
    const uint32_t n = (uint32_t)min((uint64_t)TILE, rows - r0);
    const uint32_t* idx = (const uint32_t*)idx_ptrs[l.batch] + r0;
    const T* data = (const T*)data_ptrs[l.batch];
    T* out = (T*)out_ptrs[l.batch] + r0;
    if (n == TILE && (((uintptr_t)idx) & 15) == 0 && (((uintptr_t)out) & 15) == 0) {
      uint4 ix[kVecPerThread];
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j) ix[j] = gload4((const uint4*)idx + j * kBlock + threadIdx.x);
      T g[kVecPerThread][4];
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j) {
        g[j][0] = gload_t(data + ix[j].x);
        g[j][1] = gload_t(data + ix[j].y);
        g[j][2] = gload_t(data + ix[j].z);
        g[j][3] = gload_t(data + ix[j].w);
      }
#pragma unroll
      for (int j = 0; j < kVecPerThread; ++j) {
        T* o = out + (size_t)(j * kBlock + threadIdx.x) * 4;
        if constexpr (sizeof(T) == 4) {
          gstore4(o, make_uint4((uint32_t)g[j][0], (uint32_t)g[j][1], (uint32_t)g[j][2], (uint32_t)g[j][3]));
        } else {
          gstore_vec16((ulonglong2*)o, make_ulonglong2((uint64_t)g[j][0], (uint64_t)g[j][1]));
          gstore_vec16((ulonglong2*)o + 1, make_ulonglong2((uint64_t)g[j][2], (uint64_t)g[j][3]));
        }
      }
    } else {
      for (uint32_t i = threadIdx.x; i < n; i += kBlock) gstore_t(out + i, gload_t(data + gload1(idx + i)));
    }
  }
}

// Small batches (the reference's 64 Ki-row shape): the random 4-byte gathers would pull a
// whole 64-byte line each from HBM.  Instead one workgroup owns a batch, streams its data
// through LDS window by window (coalesced 16 B/lane loads) and serves the gathers from LDS;
// indices and results of the workgroup stay in registers, so indices are read once and the
// output is written once, coalesced.  Two workgroups of 512 per CU (64 KiB window each).
constexpr int kTBlock = 512;
constexpr int kTWindowBytes = 64 << 10;
constexpr int kTIdxPerThread = 16;  // up to 8192 indices per batch are held in registers

template <typename T>
__global__ __launch_bounds__(kTBlock, 4) void k_take_lds(ColView dv, ColView iv, uint32_t nb,
                                                         const void* const* data_ptrs,
                                                         const void* const* idx_ptrs, void* const* out_ptrs) {
  __shared__ __attribute__((aligned(16))) unsigned char s_win[kTWindowBytes];
  constexpr uint32_t W = kTWindowBytes / sizeof(T);  // elements per window
  T* win = reinterpret_cast<T*>(s_win);
  for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
    const uint32_t rows = (uint32_t)batch_rows(dv, b), m = (uint32_t)batch_rows(iv, b);
    const T* data = (const T*)data_ptrs[b];
    const uint32_t* idx = (const uint32_t*)idx_ptrs[b];
    T* out = (T*)out_ptrs[b];
    uint32_t ix[kTIdxPerThread];
    T val[kTIdxPerThread];
#pragma unroll
    for (int k = 0; k < kTIdxPerThread; ++k) {
      const uint32_t i = k * kTBlock + threadIdx.x;
      ix[k] = i < m ? gload1(idx + i) : 0xFFFFFFFFu;
      val[k] = 0;
    }
    const bool vec_ok = (((uintptr_t)data) & 15) == 0;
    for (uint32_t w0 = 0; w0 < rows; w0 += W) {
      const uint32_t wn = min(W, rows - w0);
      __syncthreads();  // previous window fully consumed
      if (vec_ok && wn == W) {
        const uint4* src = (const uint4*)(data + w0);
#pragma unroll
        for (int j = 0; j < kTWindowBytes / 16 / kTBlock; ++j) ((uint4*)s_win)[j * kTBlock + threadIdx.x] = gload4(src + j * kTBlock + threadIdx.x);
      } else {
        // a partial window, or a buffer that is only element-aligned (an Arrow slice): 16-byte loads over the aligned
        // middle (stored element by element: the LDS side is then not 16-byte aligned), scalars at both ends
        constexpr uint32_t E = 16 / sizeof(T);
        const T* src = data + w0;
        const uint32_t head = min(wn, (uint32_t)(((16 - (((uintptr_t)src) & 15)) & 15) / sizeof(T)));
        const uint32_t nvec = (wn - head) / E;
        for (uint32_t i = threadIdx.x; i < head; i += kTBlock) win[i] = gload_t(src + i);
        const typename Vec16<T>::type* sv = (const typename Vec16<T>::type*)(src + head);
        for (uint32_t i = threadIdx.x; i < nvec; i += kTBlock) {
          const typename Vec16<T>::type v = gload_vec16(sv + i);
#pragma unroll
          for (uint32_t k = 0; k < E; ++k) win[head + i * E + k] = vec_get<T>(v, (int)k);
        }
        for (uint32_t i = head + nvec * E + threadIdx.x; i < wn; i += kTBlock) win[i] = gload_t(src + i);
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < kTIdxPerThread; ++k) {
        const uint32_t rel = ix[k] - w0;  // unsigned: out-of-window (and the 0xFFFFFFFF filler) fail the test
        if (rel < wn) val[k] = win[rel];
      }
    }
#pragma unroll
    for (int k = 0; k < kTIdxPerThread; ++k) {
      const uint32_t i = k * kTBlock + threadIdx.x;
      if (i < m) gstore_t(out + i, val[k]);
    }
  }
}

// Large batches (data buffer far beyond LDS, e.g. BM_TakeDpu's 4 Mi-row shape): a direct gather moves one
// 64-byte sector per 4-byte value.  An alternative of two streaming kernels (opt-in, take_variant = 4; slower than
// the gather on the benchmark shape, see run_take):
//   k_take_bucket  one workgroup per batch buckets the batch's (index, position) pairs by 64 KiB window of the data
//                  buffer (LDS histogram -> scan -> scatter of 8-byte pairs inside the batch's own pair region);
//   k_take_window  one workgroup per (batch, window): the window is read once, coalesced, into LDS, and the
//                  window's pairs are served from it.
// Bytes per index: 4 (index) + 8 + 8 (pair out and back) + sizeof(T) (result) + the data buffer read once.
constexpr int kTbBlock = 1024;
constexpr uint32_t kTbMaxWin = 1024;  // windows per batch the bucket kernel keeps counters for

// f(index value, position) over one batch's indices: two 16-byte loads in flight per thread
template <class F>
__device__ __forceinline__ void tb_for_each(const uint32_t* idx, uint32_t m, F&& f) {
  const unsigned t = threadIdx.x;
  uint32_t done = 0;
  if ((((uintptr_t)idx) & 15) == 0) {
    const uint4* v = (const uint4*)idx;
    const uint32_t nvec = m >> 2;
    uint32_t i = t;
    for (; i + kTbBlock < nvec; i += 2 * kTbBlock) {
      const uint4 a = gload4(v + i), b = gload4(v + i + kTbBlock);
      f(a.x, 4 * i), f(a.y, 4 * i + 1), f(a.z, 4 * i + 2), f(a.w, 4 * i + 3);
      const uint32_t j = 4 * (i + kTbBlock);
      f(b.x, j), f(b.y, j + 1), f(b.z, j + 2), f(b.w, j + 3);
    }
    for (; i < nvec; i += kTbBlock) {
      const uint4 a = gload4(v + i);
      f(a.x, 4 * i), f(a.y, 4 * i + 1), f(a.z, 4 * i + 2), f(a.w, 4 * i + 3);
    }
    done = nvec << 2;
  }
  for (uint32_t i = done + t; i < m; i += kTbBlock) f(gload1(idx + i), i);
}

__global__ __launch_bounds__(kTbBlock) void k_take_bucket(ColView iv, ColView dv, uint32_t nb,
                                                          const void* const* idx_ptrs, uint32_t wshift,
                                                          uint32_t nwin_max, uint2* pairs, uint32_t* win_off) {
  __shared__ uint32_t s_cnt[kTbMaxWin];
  __shared__ uint32_t s_wave[kTbBlock / kWave];
  const unsigned t = threadIdx.x, lane = lane_id(), wave = t / kWave;
  for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
    const uint32_t m = (uint32_t)batch_rows(iv, b);
    const uint64_t rows = batch_rows(dv, b);
    const uint32_t last = rows ? (uint32_t)((rows - 1) >> wshift) : 0;  // last window, < nwin_max (host)
    const uint32_t* idx = (const uint32_t*)idx_ptrs[b];
    uint2* pr = pairs + batch_row0(iv, b);
    uint32_t* off = win_off + (uint64_t)b * (nwin_max + 1);
    s_cnt[t] = 0;
    __syncthreads();
    tb_for_each(idx, m, [&](uint32_t ix, uint32_t) { atomicAdd(&s_cnt[min(ix >> wshift, last)], 1u); });
    __syncthreads();
    // exclusive scan of the counters (thread t owns window t)
    const uint32_t c = s_cnt[t];
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o, kWave);
      if ((int)lane >= o) incl += up;
    }
    if (lane == kWave - 1) s_wave[wave] = incl;
    __syncthreads();
    uint32_t excl = incl - c;
    for (unsigned w = 0; w < wave; ++w) excl += s_wave[w];
    s_cnt[t] = excl;  // now the window's cursor
    if (t < nwin_max) off[t] = t <= last ? excl : m;
    if (t == 0) off[nwin_max] = m;
    __syncthreads();
    tb_for_each(idx, m, [&](uint32_t ix, uint32_t pos) {
      const uint32_t slot = atomicAdd(&s_cnt[min(ix >> wshift, last)], 1u);
      gstore2(pr + slot, make_uint2(ix, pos));
    });
    __syncthreads();
  }
}

// Work item = (batch, window).  The scattered 4-byte result stores of one batch must meet in ONE L2 to leave it as
// whole lines (measured: spread over the 8 XCDs they reach HBM as 32-byte partial writes, 8x the bytes), so batch b
// belongs to XCD b % 8 -- workgroups are dealt to the XCDs round-robin by blockIdx -- and the streams that pass
// through (window, pairs) are loaded non-temporal so they do not push the half-written result lines out.
template <typename T>
__global__ __launch_bounds__(kTBlock, 4) void k_take_window(ColView dv, ColView iv, uint32_t nb, uint32_t wshift,
                                                            uint32_t nwin_max, const void* const* data_ptrs,
                                                            const uint2* pairs, const uint32_t* win_off,
                                                            void* const* out_ptrs) {
  __shared__ __attribute__((aligned(16))) unsigned char s_win[kTWindowBytes];
  constexpr uint32_t W = kTWindowBytes / sizeof(T);  // elements per window = 1 << wshift
  constexpr int kQ = 4;                              // pairs in flight per thread
  T* win = reinterpret_cast<T*>(s_win);
  const uint32_t xcd = blockIdx.x % kXcds, slot = blockIdx.x / kXcds, nslots = gridDim.x / kXcds;  // gridDim % 8 == 0
  const uint32_t my_batches = nb > xcd ? (nb - xcd + kXcds - 1) / kXcds : 0;
  const uint64_t items = (uint64_t)my_batches * nwin_max;
  for (uint64_t item = slot; item < items; item += nslots) {
    const uint32_t b = (uint32_t)(item / nwin_max) * kXcds + xcd, w = (uint32_t)(item % nwin_max);
    const uint32_t* off = win_off + (uint64_t)b * (nwin_max + 1) + w;
    const uint32_t lo = off[0], hi = off[1];
    if (lo == hi) continue;  // nobody asked for this window (uniform across the workgroup)
    const uint64_t rows = batch_rows(dv, b);
    const uint32_t w0 = w << wshift;
    const uint32_t wn = (uint32_t)min((uint64_t)W, rows - w0);
    const T* data = (const T*)data_ptrs[b] + w0;
    const uint2* pr = pairs + batch_row0(iv, b);
    T* out = (T*)out_ptrs[b];
    uint2 q[kQ];  // the first pairs travel together with the window
#pragma unroll
    for (int k = 0; k < kQ; ++k) q[k] = gload2_nt(pr + min(lo + k * kTBlock + threadIdx.x, hi - 1));
    __syncthreads();  // previous window fully consumed
    if (wn == W && (((uintptr_t)data) & 15) == 0) {
#pragma unroll
      for (int j = 0; j < kTWindowBytes / 16 / kTBlock; ++j)
        ((uint4*)s_win)[j * kTBlock + threadIdx.x] = gload4_nt((const uint4*)data + j * kTBlock + threadIdx.x);
    } else {
      for (uint32_t i = threadIdx.x; i < wn; i += kTBlock) win[i] = gload_t(data + i);
    }
    __syncthreads();
    for (uint32_t p0 = lo;;) {
#pragma unroll
      for (int k = 0; k < kQ; ++k) {
        const uint32_t rel = q[k].x - w0;
        // an index beyond the batch reads as 0, like the LDS kernel
        if (p0 + k * kTBlock + threadIdx.x < hi) gstore_t(out + q[k].y, rel < wn ? win[rel] : (T)0);
      }
      p0 += kQ * kTBlock;
      if (p0 >= hi) break;
#pragma unroll
      for (int k = 0; k < kQ; ++k) q[k] = gload2_nt(pr + min(p0 + k * kTBlock + threadIdx.x, hi - 1));
    }
  }
}

static int grid_for(const olap_ctx* ctx, uint64_t units, int blocks_per_cu) {
  uint64_t cap = (uint64_t)ctx->compute_units * blocks_per_cu;
  uint64_t g = units < cap ? units : cap;
  return (int)(g ? g : 1);
}

// Splits [0, nb) into chunks of batches whose staged bytes fit `budget`.
static uint64_t next_chunk(const uint64_t* rows, uint64_t nb, uint64_t b0, uint64_t bytes_per_row,
                           uint64_t budget) {
  uint64_t b = b0, acc = 0;
  while (b < nb) {
    uint64_t add = rows[b] * bytes_per_row + 512;
    if (b > b0 && acc + add > budget) break;
    acc += add;
    ++b;
  }
  return b;
}

constexpr uint64_t kHostChunkBudget = 1ull << 30;  // staging bytes per round for host inputs

template <typename T, typename ACC>
static int run_sum(olap_ctx* ctx, const void* const* ptrs, const uint64_t* rows, uint64_t nb, int in_mem,
                   uint64_t* out_sum) {
  constexpr uint32_t TILE = kBlock * kSumVec * 16 / sizeof(T);
  const uint32_t eb = sizeof(T);
  unsigned long long total = 0;
  uint64_t b0 = 0;
  while (b0 < nb) {
    const uint64_t b1 = in_mem == OLAP_MEM_HOST ? next_chunk(rows, nb, b0, eb, kHostChunkBudget) : nb;
    const uint64_t cnt = b1 - b0;
    const void* const* cols[1] = {ptrs + b0};
    OLAP_TRY(pinned_begin(ctx, (cnt + 1) * 8 * 3 + 4096));
    OLAP_TRY(ws_begin(ctx, column_set_bytes(rows + b0, cnt, 1, &eb, in_mem) + 4096));
    ColumnSet cs;
    OLAP_TRY(build_column_set(ctx, cols, 1, &eb, rows + b0, cnt, in_mem, TILE, &cs));
    unsigned long long* d_out = (unsigned long long*)ws_alloc(ctx, 8);
    if (!d_out) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted");
    OLAP_HIP(ctx, hipMemsetAsync(d_out, 0, 8, ctx->stream));
    if (cs.view.total_units) {
      PhaseScope ps(ctx, PH_SUM);
      hipLaunchKernelGGL((k_sum<T, ACC>), dim3(grid_for(ctx, cs.view.total_units, 8)), dim3(kBlock), 0,
                         ctx->stream, cs.view, cs.dev_ptr_tables[0], d_out);
      OLAP_HIP(ctx, hipGetLastError());
    }
    unsigned long long part = 0;
    OLAP_HIP(ctx, hipMemcpyAsync(&part, d_out, 8, hipMemcpyDeviceToHost, ctx->stream));
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    total += part;  // host add of the per-round partials (host/aggr/aggr_dpu.cc:82-84)
    b0 = b1;
  }
  *out_sum = total;
  return resolve_timers(ctx);
}

// Packs per-batch results (src[b][0 .. off[b+1]-off[b])) back to back: dense[off[b] + i].
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack(const void* const* src, const uint64_t* off, uint32_t nb, T* dense) {
  for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
    const T* in = static_cast<const T*>(src[b]);
    const uint64_t o = off[b], n = off[b + 1] - o;
    for (uint64_t i = threadIdx.x; i < n; i += kBlock) dense[o + i] = in[i];
  }
}

template <typename T>
static int run_filter(olap_ctx* ctx, const void* const* ptrs, const uint64_t* rows, uint64_t nb, int in_mem,
                      const void* threshold, void* const* out_ptrs, int out_mem, uint64_t* out_rows,
                      uint64_t* out_total) {
  constexpr uint32_t TILE = kTileBytes / sizeof(T);
  const uint32_t eb = sizeof(T);
  T thr;
  std::memcpy(&thr, threshold, sizeof(T));
  uint64_t grand = 0;
  uint64_t b0 = 0;
  while (b0 < nb) {
    const bool staged = in_mem == OLAP_MEM_HOST || out_mem == OLAP_MEM_HOST;
    const uint64_t b1 = staged ? next_chunk(rows, nb, b0, 2 * eb, kHostChunkBudget) : nb;
    const uint64_t cnt = b1 - b0;
    uint64_t total_rows = 0, max_rows = 0;
    for (uint64_t b = b0; b < b1; ++b) {
      total_rows += rows[b];
      if (rows[b] > max_rows) max_rows = rows[b];
    }
    // segment size: a multiple of the tile; whole 64 Ki-row batches stay one segment
    uint32_t seg_rows = 16 * TILE;
    const uint64_t want_units = (uint64_t)ctx->compute_units * 8;
    if (cnt < want_units && total_rows / want_units < seg_rows) {
      uint64_t s = total_rows / want_units;
      s = align_up(s ? s : 1, TILE);
      seg_rows = (uint32_t)(s < TILE ? TILE : (s > 16 * TILE ? 16 * TILE : s));
    }
    // Enough LONG batches to occupy the chip on their own (filter_whole_batches per CU, default 2): one workgroup walks a
    // whole batch with its running offset -- one pass over the data (5 B per row at s = 0.25) instead of count + scan +
    // compact (9 B).  2^30 rows in 512 batches: 1.15 ms against 1.84 ms.
    if (max_rows > seg_rows && ctx->filter_whole_batches > 0 && cnt >= (uint64_t)ctx->compute_units * ctx->filter_whole_batches &&
        max_rows < (1ull << 31))
      seg_rows = (uint32_t)align_up(max_rows, TILE);
    const bool multi_seg = max_rows > seg_rows;
    const void* const* cols[1] = {ptrs + b0};
    uint64_t n_seg_est = 0;
    for (uint64_t b = b0, last_rows = ~0ull, last_segs = 0; b < b1; ++b) {  // (one division per distinct batch size in a row)
      if (rows[b] != last_rows) {
        last_rows = rows[b];
        last_segs = (rows[b] + seg_rows - 1) / seg_rows;
      }
      n_seg_est += last_segs;
    }
    uint64_t need = column_set_bytes(rows + b0, cnt, 1, &eb, in_mem) + align_up(cnt * 8, 256) * 3 +
                    align_up(n_seg_est * 12, 256) * 2 + 8192;
    if (out_mem == OLAP_MEM_HOST) {
      for (uint64_t b = b0; b < b1; ++b) need += align_up(rows[b] * eb, 256);
      need += align_up(total_rows * eb + 16, 256) + align_up((cnt + 1) * 8, 256);  // packed copy + its offsets
    }
    OLAP_TRY(pinned_begin(ctx, (cnt + 1) * 8 * 8 + 4096));
    OLAP_TRY(ws_begin(ctx, need));
    ColumnSet cs;
    OLAP_TRY(build_column_set(ctx, cols, 1, &eb, rows + b0, cnt, in_mem, seg_rows, &cs));
    // output pointer table
    void** h_out = (void**)pinned_alloc(ctx, (cnt ? cnt : 1) * 8);
    void** d_out_tbl = (void**)ws_alloc(ctx, (cnt ? cnt : 1) * 8);
    unsigned long long* d_rows = (unsigned long long*)ws_alloc(ctx, (cnt ? cnt : 1) * 8);
    unsigned long long* h_rows = (unsigned long long*)pinned_alloc(ctx, (cnt ? cnt : 1) * 8);
    if (!h_out || !d_out_tbl || !d_rows || !h_rows) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted");
    for (uint64_t b = 0; b < cnt; ++b) {
      if (out_mem == OLAP_MEM_HOST) {
        h_out[b] = ws_alloc(ctx, rows[b0 + b] * eb ? rows[b0 + b] * eb : 1);
        if (!h_out[b]) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (filter output staging)");
      } else {
        if (rows[b0 + b] && !out_ptrs[b0 + b]) return ctx->fail(OLAP_ERR_INVALID, "null output pointer");
        if (((uintptr_t)out_ptrs[b0 + b]) & (uintptr_t)(eb - 1)) return ctx->fail(OLAP_ERR_INVALID, "output pointer misaligned");
        h_out[b] = out_ptrs[b0 + b];
      }
    }
    OLAP_HIP(ctx, hipMemcpyAsync(d_out_tbl, h_out, (cnt ? cnt : 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    OLAP_HIP(ctx, hipMemsetAsync(d_rows, 0, (cnt ? cnt : 1) * 8, ctx->stream));
    uint64_t* d_seg_base = nullptr;
    if (cs.view.total_units) {
      const int grid = grid_for(ctx, cs.view.total_units, 8);
      if (multi_seg) {
        uint32_t* d_seg_count = (uint32_t*)ws_alloc(ctx, cs.view.total_units * 4);
        d_seg_base = (uint64_t*)ws_alloc(ctx, cs.view.total_units * 8);
        if (!d_seg_count || !d_seg_base) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (segments)");
        {
          PhaseScope ps(ctx, PH_FILTER_COUNT);
          hipLaunchKernelGGL((k_filter_count<T>), dim3(grid), dim3(kBlock), 0, ctx->stream, cs.view,
                             cs.dev_ptr_tables[0], thr, d_seg_count);
          hipLaunchKernelGGL(k_filter_segscan, dim3(grid_for(ctx, cnt, 8)), dim3(kBlock), 0, ctx->stream,
                             cs.view, d_seg_count, d_seg_base);
          OLAP_HIP(ctx, hipGetLastError());
        }
      }
      PhaseScope ps(ctx, PH_FILTER_COMPACT);
      // Dense tiles through an LDS strip and 16-byte stores (k_filter_compact_lds): for segments of the benchmark's size -- short
      // batches pay for its longer prologue (1024-row batches: 3.0 -> 3.6 ms per 2^30 rows), and a workgroup that walks a whole
      // long batch alone is bound by its own latency chain (512 x 2 Mi rows: 1.15 -> 1.24 ms).  Needs 16-byte aligned output buffers.
      bool lds_stores = ctx->filter_stores >= 2 ||
                        (ctx->filter_stores == 1 && seg_rows == 16 * TILE && total_rows / cs.view.total_units >= 8 * (uint64_t)TILE);
      for (uint64_t b = 0; b < cnt && lds_stores; ++b) lds_stores = (((uintptr_t)h_out[b]) & 15) == 0;
      if (lds_stores)
        hipLaunchKernelGGL((k_filter_compact_lds<T>), dim3(grid), dim3(kBlock), 0, ctx->stream, cs.view, cs.dev_ptr_tables[0],
                           (void* const*)d_out_tbl, thr, (const uint64_t*)d_seg_base, d_rows);
      else
        hipLaunchKernelGGL((k_filter_compact<T>), dim3(grid), dim3(kBlock), 0, ctx->stream, cs.view,
                           cs.dev_ptr_tables[0], (void* const*)d_out_tbl, thr, (const uint64_t*)d_seg_base, d_rows);
      OLAP_HIP(ctx, hipGetLastError());
    }
    OLAP_HIP(ctx, hipMemcpyAsync(h_rows, d_rows, (cnt ? cnt : 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint64_t b = 0; b < cnt; ++b) {
      out_rows[b0 + b] = h_rows[b];
      grand += h_rows[b];
    }
    if (out_mem == OLAP_MEM_HOST && cnt) {
      // the per-batch results sit in worst-case-sized staging: pack them back to back so
      // the way out is a few large DMAs instead of one per batch
      uint64_t* h_off = (uint64_t*)pinned_alloc(ctx, (cnt + 1) * 8);
      uint64_t* d_off = (uint64_t*)ws_alloc(ctx, (cnt + 1) * 8);
      if (!h_off || !d_off) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (pack offsets)");
      h_off[0] = 0;
      for (uint64_t b = 0; b < cnt; ++b) h_off[b + 1] = h_off[b] + h_rows[b];
      T* dense = (T*)ws_alloc(ctx, h_off[cnt] * eb + 16);
      if (!dense) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (packed filter output)");
      std::vector<StageItem> items;
      items.reserve(cnt);
      for (uint64_t b = 0; b < cnt; ++b)
        if (h_rows[b]) items.push_back(StageItem{dense + h_off[b], out_ptrs[b0 + b], h_rows[b] * eb});
      PhaseScope ps(ctx, PH_COPY_FROM);
      if (h_off[cnt]) {
        OLAP_HIP(ctx, hipMemcpyAsync(d_off, h_off, (cnt + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL((k_pack<T>), dim3((unsigned)std::min<uint64_t>(cnt, 65535u * 16u)), dim3(kBlock), 0, ctx->stream,
                           (const void* const*)d_out_tbl, (const uint64_t*)d_off, (uint32_t)cnt, dense);
        OLAP_HIP(ctx, hipGetLastError());
        OLAP_TRY(stage_d2h(ctx, items));
      }
    }
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    b0 = b1;
  }
  if (out_total) *out_total = grand;
  return resolve_timers(ctx);
}

template <typename T>
static int run_take(olap_ctx* ctx, const void* const* data_ptrs, const uint64_t* data_rows,
                    const uint32_t* const* idx_ptrs, const uint64_t* idx_rows, uint64_t nb, int in_mem,
                    void* const* out_ptrs, int out_mem) {
  constexpr uint32_t TILE = kBlock * kVecPerThread * 4;
  const uint32_t eb = sizeof(T), ib = 4;
  uint64_t b0 = 0;
  while (b0 < nb) {
    uint64_t b1 = nb;
    const bool staged = in_mem == OLAP_MEM_HOST || out_mem == OLAP_MEM_HOST;
    if (staged) {  // chunk by data + idx + out bytes
      uint64_t acc = 0;
      b1 = b0;
      while (b1 < nb) {
        uint64_t add = data_rows[b1] * eb + idx_rows[b1] * (4 + eb) + 1024;
        if (b1 > b0 && acc + add > kHostChunkBudget) break;
        acc += add;
        ++b1;
      }
    }
    const uint64_t cnt = b1 - b0;
    const void* const* dcols[1] = {data_ptrs + b0};
    const void* const* icols[1] = {(const void* const*)(idx_ptrs + b0)};
    uint64_t need = column_set_bytes(data_rows + b0, cnt, 1, &eb, in_mem) +
                    column_set_bytes(idx_rows + b0, cnt, 1, &ib, in_mem) + align_up(cnt * 8, 256) + 8192;
    if (out_mem == OLAP_MEM_HOST)
      for (uint64_t b = b0; b < b1; ++b) need += align_up(idx_rows[b] * eb, 256);
    // choose the kernel by batch shape
    uint64_t max_data = 0, max_idx = 0, sum_idx = 0;
    for (uint64_t b = b0; b < b1; ++b) {
      if (data_rows[b] > max_data) max_data = data_rows[b];
      if (idx_rows[b] > max_idx) max_idx = idx_rows[b];
      sum_idx += idx_rows[b];
    }
    const bool lds_path = ctx->take_variant != 1 && max_idx <= (uint64_t)kTBlock * kTIdxPerThread &&
                          max_data * eb <= 8ull * kTWindowBytes && cnt >= (uint64_t)ctx->compute_units;
    // window-bucketed path (take_variant = 4 only): reads the data buffers once (sum_data * eb) plus ~24 B per index
    // instead of one 64-byte sector per index -- and loses: at SF=2048 (2048 x 4 Mi rows, 512 Ki indices each) it takes
    // 6.9 + 17.6 ms against 20-22 ms for the direct gather, because the scattered 4-byte result stores leave the L2
    // as 32-byte partial writes (WRITE_SIZE 38.5 GB for 4.3 GB of results), with or without keeping a batch on one
    // XCD.  Kept as a tested alternative; profiles/r2_ab_experiments.md has the numbers.
    const uint32_t wshift = eb == 4 ? 14 : 13;  // 64 KiB windows
    const uint64_t nwin_max = max_data ? ((max_data - 1) >> wshift) + 1 : 1;
    const bool bucket_path = ctx->take_variant == 4 && !lds_path && nwin_max <= kTbMaxWin && max_idx < (1ull << 32) && sum_idx > 0;
    if (bucket_path) need += align_up(sum_idx * 8, 256) + align_up(cnt * (nwin_max + 1) * 4, 256);
    OLAP_TRY(pinned_begin(ctx, (cnt + 1) * 8 * 8 + 4096));
    OLAP_TRY(ws_begin(ctx, need));
    // large batches, experiment knob (take_variant = 3): small tiles keep fewer batches in
    // flight; measured no gain at SF=2048 -- every random gather moves a whole 64-byte sector
    // (2^30 gathers in 21.9 ms = 49 G sectors/s, the random-sector rate of the memory), so this stays off.
    const bool small_tiles = !lds_path && max_data * eb > (1ull << 20) && ctx->take_variant == 3;
    const uint32_t tile_sel = small_tiles ? kBlock * 4 : TILE;
    ColumnSet ds, is;
    OLAP_TRY(build_column_set(ctx, dcols, 1, &eb, data_rows + b0, cnt, in_mem, tile_sel, &ds));
    OLAP_TRY(build_column_set(ctx, icols, 1, &ib, idx_rows + b0, cnt, in_mem, tile_sel, &is));
    void** h_out = (void**)pinned_alloc(ctx, (cnt ? cnt : 1) * 8);
    void** d_out_tbl = (void**)ws_alloc(ctx, (cnt ? cnt : 1) * 8);
    if (!h_out || !d_out_tbl) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted");
    for (uint64_t b = 0; b < cnt; ++b) {
      if (out_mem == OLAP_MEM_HOST) {
        h_out[b] = ws_alloc(ctx, idx_rows[b0 + b] * eb ? idx_rows[b0 + b] * eb : 1);
        if (!h_out[b]) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (take output staging)");
      } else {
        if (idx_rows[b0 + b] && !out_ptrs[b0 + b]) return ctx->fail(OLAP_ERR_INVALID, "null output pointer");
        if (((uintptr_t)out_ptrs[b0 + b]) & (uintptr_t)(eb - 1)) return ctx->fail(OLAP_ERR_INVALID, "output pointer misaligned");
        h_out[b] = out_ptrs[b0 + b];
      }
    }
    OLAP_HIP(ctx, hipMemcpyAsync(d_out_tbl, h_out, (cnt ? cnt : 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (is.view.total_units) {
      PhaseScope ps(ctx, PH_TAKE);
      if (bucket_path) {
        uint2* d_pairs = (uint2*)ws_alloc(ctx, sum_idx * 8);
        uint32_t* d_win_off = (uint32_t*)ws_alloc(ctx, cnt * (nwin_max + 1) * 4);
        if (!d_pairs || !d_win_off) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (take pairs)");
        hipLaunchKernelGGL(k_take_bucket, dim3(grid_for(ctx, cnt, 1)), dim3(kTbBlock), 0, ctx->stream, is.view, ds.view,
                           (uint32_t)cnt, is.dev_ptr_tables[0], wshift, (uint32_t)nwin_max, d_pairs, d_win_off);
        const int wgrid = (int)align_up((uint64_t)grid_for(ctx, cnt * nwin_max, 2), kXcds);
        hipLaunchKernelGGL((k_take_window<T>), dim3(wgrid), dim3(kTBlock), 0, ctx->stream, ds.view, is.view, (uint32_t)cnt,
                           wshift, (uint32_t)nwin_max, ds.dev_ptr_tables[0], (const uint2*)d_pairs,
                           (const uint32_t*)d_win_off, (void* const*)d_out_tbl);
      } else if (lds_path) {
        hipLaunchKernelGGL((k_take_lds<T>), dim3(grid_for(ctx, cnt, 2)), dim3(kTBlock), 0, ctx->stream, ds.view, is.view,
                           (uint32_t)cnt, ds.dev_ptr_tables[0], is.dev_ptr_tables[0], (void* const*)d_out_tbl);
      } else if (small_tiles) {
        hipLaunchKernelGGL((k_take<T, 1>), dim3(grid_for(ctx, is.view.total_units, 8)), dim3(kBlock), 0, ctx->stream,
                           is.view, ds.dev_ptr_tables[0], is.dev_ptr_tables[0], (void* const*)d_out_tbl);
      } else {
        hipLaunchKernelGGL((k_take<T, 4>), dim3(grid_for(ctx, is.view.total_units, 8)), dim3(kBlock), 0, ctx->stream,
                           is.view, ds.dev_ptr_tables[0], is.dev_ptr_tables[0], (void* const*)d_out_tbl);
      }
      OLAP_HIP(ctx, hipGetLastError());
    }
    if (out_mem == OLAP_MEM_HOST) {
      std::vector<StageItem> items;
      items.reserve(cnt);
      for (uint64_t b = 0; b < cnt; ++b)
        if (idx_rows[b0 + b]) items.push_back(StageItem{h_out[b], out_ptrs[b0 + b], idx_rows[b0 + b] * eb});
      PhaseScope ps(ctx, PH_COPY_FROM);
      OLAP_TRY(stage_d2h(ctx, items));
    }
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    b0 = b1;
  }
  return resolve_timers(ctx);
}

static int check_common(olap_ctx* ctx, int dtype, const void* p, const uint64_t* rows, uint64_t nb, int mem) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (!dtype_ok(dtype)) return ctx->fail(OLAP_ERR_INVALID, "unknown dtype %d", dtype);
  if (nb && (!p || !rows)) return ctx->fail(OLAP_ERR_INVALID, "null batch table");
  if (mem != OLAP_MEM_HOST && mem != OLAP_MEM_DEVICE) return ctx->fail(OLAP_ERR_INVALID, "bad memory space %d", mem);
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
  return OLAP_OK;
}

}  // namespace olap

using namespace olap;

extern "C" {

int olap_sum(olap_ctx* ctx, int dtype, const void* const* ptrs, const uint64_t* rows, uint64_t n_batches,
             int in_mem, uint64_t* out_sum) {
  OLAP_TRY(check_common(ctx, dtype, ptrs, rows, n_batches, in_mem));
  if (!out_sum) return ctx->fail(OLAP_ERR_INVALID, "out_sum is null");
  switch (dtype) {
    case OLAP_U32: return run_sum<uint32_t, unsigned long long>(ctx, ptrs, rows, n_batches, in_mem, out_sum);
    case OLAP_I32: return run_sum<int32_t, long long>(ctx, ptrs, rows, n_batches, in_mem, out_sum);
    case OLAP_U64: return run_sum<uint64_t, unsigned long long>(ctx, ptrs, rows, n_batches, in_mem, out_sum);
    default: return run_sum<int64_t, unsigned long long>(ctx, ptrs, rows, n_batches, in_mem, out_sum);
  }
}

int olap_filter_lt(olap_ctx* ctx, int dtype, const void* const* ptrs, const uint64_t* rows, uint64_t n_batches,
                   int in_mem, const void* threshold, void* const* out_ptrs, int out_mem, uint64_t* out_rows,
                   uint64_t* out_total) {
  OLAP_TRY(check_common(ctx, dtype, ptrs, rows, n_batches, in_mem));
  if (!threshold) return ctx->fail(OLAP_ERR_INVALID, "threshold is null");
  if (n_batches && (!out_ptrs || !out_rows)) return ctx->fail(OLAP_ERR_INVALID, "null output table");
  if (out_mem != OLAP_MEM_HOST && out_mem != OLAP_MEM_DEVICE) return ctx->fail(OLAP_ERR_INVALID, "bad out_mem");
  switch (dtype) {
    case OLAP_U32: return run_filter<uint32_t>(ctx, ptrs, rows, n_batches, in_mem, threshold, out_ptrs, out_mem, out_rows, out_total);
    case OLAP_I32: return run_filter<int32_t>(ctx, ptrs, rows, n_batches, in_mem, threshold, out_ptrs, out_mem, out_rows, out_total);
    case OLAP_U64: return run_filter<uint64_t>(ctx, ptrs, rows, n_batches, in_mem, threshold, out_ptrs, out_mem, out_rows, out_total);
    default: return run_filter<int64_t>(ctx, ptrs, rows, n_batches, in_mem, threshold, out_ptrs, out_mem, out_rows, out_total);
  }
}

int olap_take(olap_ctx* ctx, int dtype, const void* const* data_ptrs, const uint64_t* data_rows,
              const uint32_t* const* idx_ptrs, const uint64_t* idx_rows, uint64_t n_batches, int in_mem,
              void* const* out_ptrs, int out_mem) {
  OLAP_TRY(check_common(ctx, dtype, data_ptrs, data_rows, n_batches, in_mem));
  if (n_batches && (!idx_ptrs || !idx_rows || !out_ptrs)) return ctx->fail(OLAP_ERR_INVALID, "null index/output table");
  if (out_mem != OLAP_MEM_HOST && out_mem != OLAP_MEM_DEVICE) return ctx->fail(OLAP_ERR_INVALID, "bad out_mem");
  if (dtype_bytes(dtype) == 4)
    return run_take<uint32_t>(ctx, data_ptrs, data_rows, idx_ptrs, idx_rows, n_batches, in_mem, out_ptrs, out_mem);
  return run_take<uint64_t>(ctx, data_ptrs, data_rows, idx_ptrs, idx_rows, n_batches, in_mem, out_ptrs, out_mem);
}

}  // extern "C"
