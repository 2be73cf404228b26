// radix64.h -- device code of the PK-FK join on 64-bit keys (olap_join_u64; included by join64.hip only).
//
// Reference: the hash table's 64-bit key mode, dpu/shared/hashtable/hashtable.h:13-18 (HT_64BIT_KEYS:
// ht_key_t = uint64_t, values stay uint32_t) and hashtable.c:23-26 (its hash: (key >> 33) ^ key ^ (key << 11),
// truncated to 32 bits), with the table semantics of hashtable.c:89-192 (open addressing, an equal key overwrites, first
// cell = hash & (n_buckets - 1); the reference then probes linearly, these tables step by double hashing -- table_put /
// table_get in probe_kernels.h: where a key sits never shows in the rows).  The reference's partition kernel is written for
// T = uint32_t only (dpu/shared/kernels/partition.c:20-28,45-46); for 64-bit keys the bucket is taken, the
// same way, from the top bits of the Wang mix -- of the table hash: bucket = wang_hash(ht_hash64(key)) >> shift.
//
// Tuples are 16 bytes {key lo, key hi, payload, 0} through the radix passes (exact layout: histogram + scan +
// tile-staged scatter); the LDS table keeps the 64-bit keys (no direct-address form: that needs the 32-bit
// hash bijection).  Bytes per row pair: read 12 + 12, per pass 32 + 32, probe 32 read + 16 written.
#pragma once
#include "radix_kernels.h"

namespace olap {

constexpr int k64Block = 512;
constexpr int k64Tile = 4096;                 // tuples per tile: 64 KiB of staging
constexpr int k64Tpt = k64Tile / k64Block;    // 8
constexpr int k64MaxFBits = 9;                // <= 512 buckets per pass (two workgroups per CU)
constexpr int k64MaxF = 1 << k64MaxFBits;
constexpr uint32_t k64ScatterLds = k64Tile * 16 + k64MaxF * 8 + (k64MaxF + 4) * 4 + 64;
constexpr int k64Slots = 4096;                // LDS table slots per partition: 32 KiB keys + 16 KiB values
constexpr uint32_t k64TableCap = k64Slots / 4 * 3;
constexpr uint32_t k64TargetBuildRows = k64Slots / 2;
constexpr unsigned long long k64Empty = ~0ull;

// hashtable.c:23-26 (HT_64BIT_KEYS)
__host__ __device__ __forceinline__ uint32_t ht_hash64(uint64_t key) { return (uint32_t)((key >> 33) ^ key ^ (key << 11)); }
__host__ __device__ __forceinline__ uint32_t part_hash64(uint64_t key) { return wang_hash(ht_hash64(key)); }

struct Tile64 {
  uint64_t key[k64Tpt];
  uint32_t pay[k64Tpt];
  uint32_t mask;
};

// One tile of (key, payload): the u64 key column + u32 payload column of the batches (SRC_COLS; payload null: global row
// index) or the 16-byte tuple array of the previous pass (SRC_TUPLES, PassSrc::tuples reinterpreted).  Thread t holds
// rows 2 * (j * 512 + t) and + 1 of the tile, j = 0..3.
template <int SRC, bool WANT_PAYLOAD>
__device__ __forceinline__ void load_tile64(const PassSrc& s, const TileRange& r, uint64_t tile, Tile64& t) {
  t.mask = 0;
  if (SRC == SRC_COLS) {
    const Loc l = locate(s.v, tile);
    const uint64_t* kp = (const uint64_t*)uniform_u64((uint64_t)s.key_ptrs[l.batch]);
    const uint32_t* pp = s.pay_ptrs ? (const uint32_t*)uniform_u64((uint64_t)s.pay_ptrs[l.batch]) : nullptr;
    const uint64_t rows = batch_rows(s.v, l.batch), r0 = l.local_unit * k64Tile;
    const uint32_t n = (uint32_t)min((uint64_t)k64Tile, rows - r0);
    const uint32_t rowid0 = (uint32_t)(batch_row0(s.v, l.batch) + r0);
    kp += r0;
    if (pp) pp += r0;
    const bool fast = n == k64Tile && (((uintptr_t)kp) & 15) == 0 && (((uintptr_t)pp) & 7) == 0;
#pragma unroll
    for (int j = 0; j < k64Tpt / 2; ++j) {
      const uint32_t e = (j * k64Block + threadIdx.x) * 2;
      if (fast) {
        const uint4 kv = gload4(kp + e);
        t.key[2 * j] = ((uint64_t)kv.y << 32) | kv.x;
        t.key[2 * j + 1] = ((uint64_t)kv.w << 32) | kv.z;
        if (WANT_PAYLOAD) {
          if (pp) {
            const uint2 pv = gload2(pp + e);
            t.pay[2 * j] = pv.x;
            t.pay[2 * j + 1] = pv.y;
          } else {
            t.pay[2 * j] = rowid0 + e;
            t.pay[2 * j + 1] = rowid0 + e + 1;
          }
        }
        t.mask |= 3u << (2 * j);
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool ok = e + h < n;
          t.key[2 * j + h] = ok ? gload1_u64(kp + e + h) : 0ull;
          if (WANT_PAYLOAD) t.pay[2 * j + h] = ok ? (pp ? gload1(pp + e + h) : rowid0 + e + h) : 0u;
          t.mask |= (ok ? 1u : 0u) << (2 * j + h);
        }
      }
    }
  } else {
    const uint4* tv = (const uint4*)s.tuples;
    const uint64_t base = r.a + tile * (uint64_t)k64Tile;
#pragma unroll
    for (int j = 0; j < k64Tpt; ++j) {
      const uint64_t i = base + (uint64_t)j * k64Block + threadIdx.x;
      const bool ok = i >= r.lo && i < r.hi;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) v = tv[i];
      t.key[j] = ((uint64_t)v.y << 32) | v.x;
      t.pay[j] = v.z;
      t.mask |= (ok ? 1u : 0u) << j;
    }
  }
}

// ------------------------------------------------------------------ hist (build_histogram, partition.c:67-92)
template <int SRC>
__global__ __launch_bounds__(k64Block) void k64_hist(PassSrc s, uint32_t shift, uint32_t fbits, uint32_t* hist) {
  __shared__ uint32_t cl[k64MaxF];
  const uint32_t F = 1u << fbits;
  for (uint32_t b = threadIdx.x; b < F; b += k64Block) cl[b] = 0;
  __syncthreads();
  const TileRange r = block_tiles<SRC, k64Tile>(s);
  for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
    Tile64 t;
    load_tile64<SRC, false>(s, r, tile, t);
#pragma unroll
    for (int i = 0; i < k64Tpt; ++i)
      if (t.mask & (1u << i)) atomicAdd(&cl[(part_hash64(t.key[i]) >> shift) & (F - 1)], 1u);
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < F; b += k64Block) hist[(uint64_t)blockIdx.x * F + b] = cl[b];
}

// --------------------------------------------------------------- scatter (partition_array, partition.c:167-294)
// Tile-staged: rank inside the bucket by an LDS atomic, exclusive scan of the bucket counts, the tile's tuples
// staged grouped by bucket, copied out as runs (consecutive lanes -> consecutive 16-byte tuples of a run).
template <int SRC>
__global__ __launch_bounds__(k64Block) void k64_scatter(PassSrc s, uint32_t shift, uint32_t fbits, const uint32_t* hist,
                                                         const uint64_t* seg_off_out, uint4* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* stage = (uint4*)smem;
  uint64_t* gcur = (uint64_t*)(smem + k64Tile * 16);
  uint32_t* cl = (uint32_t*)(smem + k64Tile * 16 + k64MaxF * 8);
  uint32_t* wtot = cl + (k64MaxF + 4);
  const uint32_t F = 1u << fbits;
  const TileRange r = block_tiles<SRC, k64Tile>(s);
  const uint32_t b_own = threadIdx.x;  // one bucket per thread (F <= 512 = block size)
  if (b_own < F) {
    gcur[b_own] = seg_off_out[(uint64_t)r.seg * F + b_own] + hist[(uint64_t)blockIdx.x * F + b_own];
    cl[b_own] = 0;
  }
  __syncthreads();
  const unsigned lane = lane_id(), wave = threadIdx.x / kWave;
  for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
    Tile64 t;
    load_tile64<SRC, true>(s, r, tile, t);
    uint32_t br[k64Tpt];
#pragma unroll
    for (int i = 0; i < k64Tpt; ++i) {
      br[i] = 0;
      if (t.mask & (1u << i)) {
        const uint32_t b = (part_hash64(t.key[i]) >> shift) & (F - 1);
        br[i] = (b << 16) | atomicAdd(&cl[b], 1u);
      }
    }
    __syncthreads();
    const uint32_t cnt = b_own < F ? cl[b_own] : 0u;
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o, kWave);
      if ((int)lane >= o) incl += y;
    }
    if (lane == kWave - 1) wtot[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - cnt, n_tile = 0;
#pragma unroll
    for (int w = 0; w < k64Block / kWave; ++w) {
      if ((unsigned)w < wave) pre += wtot[w];
      n_tile += wtot[w];
    }
    if (b_own < F) cl[b_own] = pre;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < k64Tpt; ++i)
      if (t.mask & (1u << i))
        stage[cl[br[i] >> 16] + (br[i] & 0xFFFFu)] = make_uint4((uint32_t)t.key[i], (uint32_t)(t.key[i] >> 32), t.pay[i], 0u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_tile; i += k64Block) {
      const uint4 v = stage[i];
      const uint32_t b = (part_hash64(((uint64_t)v.y << 32) | v.x) >> shift) & (F - 1);
      out[gcur[b] + (i - cl[b])] = v;
    }
    __syncthreads();
    if (b_own < F) {
      gcur[b_own] += cnt;
      cl[b_own] = 0;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------ build + probe (hash_build.c:9-35, hash_probe.c:9-46)
// counters: [0] missing keys, [1] duplicate build keys, [2] overflowing partitions, [3] max build rows
__global__ __launch_bounds__(k64Block) void k64_probe(const uint4* R, const uint64_t* roff, const uint4* L, const uint64_t* loff,
                                                       uint32_t nparts, uint64_t* out_fk, uint32_t* out_y, uint32_t* out_x,
                                                       unsigned long long* counters, uint32_t* miss_bits, ProbeSlices slices) {
  __shared__ unsigned long long s_keys[k64Slots];
  __shared__ uint32_t s_vals[k64Slots];
  __shared__ uint32_t s_sp[2];  // the one key equal to the empty marker: {present, value}
  uint32_t my_miss = 0, my_dup = 0;
  uint64_t my_max = 0;
  // (partition, slice of its probe rows): slice 0 with the partition, further slices from the list (probe_kernels.h)
  auto body = [&](const uint32_t p, const uint32_t slice) {
    const uint64_t rlo = roff[p], rhi = roff[p + 1], plo = loff[p], phi = loff[p + 1];
    if (rhi - rlo > my_max) my_max = rhi - rlo;
    if (phi == plo) return;
    if (rhi - rlo > k64TableCap) {
      if (threadIdx.x == 0 && slice == 0) atomicAdd(&counters[2], 1ull);
      return;
    }
    const uint64_t llo = plo + (uint64_t)slice * kProbeSlice, lhi = slices.count ? min(phi, llo + kProbeSlice) : phi;
    if (llo >= lhi) return;
    for (uint32_t i = threadIdx.x; i < k64Slots; i += k64Block) s_keys[i] = k64Empty;
    if (threadIdx.x == 0) s_sp[0] = 0;
    __syncthreads();
    for (uint64_t i = rlo + threadIdx.x; i < rhi; i += k64Block) {  // build: table[pk] = x
      const uint4 v = R[i];
      const unsigned long long k = ((unsigned long long)v.y << 32) | v.x;
      if (k == k64Empty) {
        s_sp[1] = v.z;
        if (atomicExch(&s_sp[0], 1u) && slice == 0) ++my_dup;
        continue;
      }
      if (table_put<k64Slots>(s_keys, s_vals, k64Empty, k, ht_hash64(k), v.z, true) && slice == 0) ++my_dup;  // first cell: hashtable.c:99-101
    }
    __syncthreads();
    for (uint64_t i = llo + threadIdx.x; i < lhi; i += k64Block) {  // probe + take of the right payload
      const uint4 v = L[i];
      const unsigned long long k = ((unsigned long long)v.y << 32) | v.x;
      uint32_t xv = 0;
      bool hit = false;
      if (k == k64Empty) {
        hit = s_sp[0] != 0;
        xv = s_sp[1];
      } else {
        hit = table_get<k64Slots>(s_keys, s_vals, k64Empty, k, ht_hash64(k), true, xv);
      }
      if (!hit) {
        ++my_miss;
        if (miss_bits) atomicOr(&miss_bits[i >> 5], 1u << (i & 31));
      }
      out_fk[i] = k;
      out_y[i] = v.z;
      out_x[i] = xv;
    }
    __syncthreads();
  };
  for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) body(p, 0u);
  const uint32_t n_extra = slices.count ? min(*slices.count, slices.capacity) : 0u;
  for (uint32_t w = blockIdx.x; w < n_extra; w += gridDim.x) body(slices.list[w].x, slices.list[w].y);
  const uint64_t wm = wave_reduce_add((uint64_t)my_miss), wd = wave_reduce_add((uint64_t)my_dup);
  if (lane_id() == 0) {
    if (wm) atomicAdd(&counters[0], (unsigned long long)wm);
    if (wd) atomicAdd(&counters[1], (unsigned long long)wd);
  }
  if (threadIdx.x == 0) atomicMax(&counters[3], (unsigned long long)my_max);
}

// order-preserving compaction of a 64-bit column by the miss bitmap (inner-join mode); 32-bit columns use k_compact_rows
__global__ __launch_bounds__(256) void k64_compact_rows(const uint64_t* src, const uint32_t* miss_bits, const uint64_t* base,
                                                          uint64_t n_rows, uint64_t* dst) {
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_run;
  const uint64_t seg = blockIdx.x;
  const uint64_t lo = seg * kCompactSeg, hi = min(n_rows, lo + kCompactSeg);
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (uint64_t r0 = lo; r0 < hi; r0 += 256) {
    const uint64_t row = r0 + threadIdx.x;
    const bool keep = row < hi && !((miss_bits[row >> 5] >> (row & 31)) & 1u);
    const unsigned long long ballot = __ballot(keep);
    const uint32_t before = __popcll(ballot & ((1ull << lane_id()) - 1ull));
    if (lane_id() == 0) s_wave[threadIdx.x / kWave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    uint32_t off = s_run;
    for (unsigned w = 0; w < threadIdx.x / kWave; ++w) off += s_wave[w];
    if (keep) dst[base[seg] + off + before] = src[row];
    __syncthreads();
    if (threadIdx.x == 0) s_run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
}

}  // namespace olap
