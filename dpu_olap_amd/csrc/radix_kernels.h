// radix_kernels.h -- device code of the radix passes (included by join.hip only).
//
// Reference behaviour (upmem/dpu_olap): dpu/shared/kernels/partition.c:20-28 (wang hash),
// :45-46 (bucket = top bits), :67-92 histogram, :94-137 prefix sum, :167-294 scatter; the
// shuffle of host/partition/partitioner.cc:280-375 never leaves HBM here.
// A pass reads the batched key/payload columns (SRC_COLS, pass 0) or the tuple array of the
// previous pass cut into segments (SRC_TUPLES), and is one of
//   hist + scan + scatter    exact layout (k_part_hist, k_part_scan, k_part_scatter[_wc])
//   capacity scatter          last pass, fixed regions, no histogram (k_part_scatter_wc<.., CAPM>)
//   claiming scatter          pass 0 of a two-pass plan, no histogram (k_part_scatter_claim)
#pragma once
#include "common.h"

namespace olap {

constexpr int kPBlock = 512;                  // 8 waves per partition workgroup
constexpr int kPTile = 8192;                  // tuples per tile
constexpr int kTPT = kPTile / kPBlock;        // 16 tuples per thread
constexpr int kMaxFBits = 10;                 // <= 1024 buckets per pass
constexpr int kMaxF = 1 << kMaxFBits;
constexpr int kMaxPasses = 4;
constexpr uint32_t kScatterLds = kPTile * 8 + kMaxF * 8 + (kMaxF + 4) * 4 + 64;

constexpr int kJBlock = 512;
constexpr int kJU = 4;                        // 16-B tuple vectors in flight per thread (build / probe)
constexpr int kSlots = 8192;                  // LDS hash table slots per partition
constexpr uint32_t kTableCap = kSlots / 4 * 3;  // max build rows (load <= 0.75, hashtable.h:55)
constexpr uint32_t kTargetBuildRows = kSlots / 2;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;

enum { SRC_COLS = 0, SRC_TUPLES = 1 };

// Input of a pass: either the batched key/payload columns (pass 0) or the tuple
// array of the previous pass cut into segments.
struct PassSrc {
  // SRC_COLS
  ColView v;
  const void* const* key_ptrs;
  const void* const* pay_ptrs;  // may be null: payload = global row index
  uint64_t tiles_per_block;
  // SRC_COLS, straight-line loader only: the batches are consecutive slices of one buffer per column, starting here (null: use the tables)
  const uint32_t* contig_key;
  const uint32_t* contig_pay;
  uint32_t interleave;  // ring scatter: workgroup w takes tiles w, w + grid, w + 2 grid, ... (0: a contiguous run of tiles_per_block)
  // SRC_TUPLES
  const uint2* tuples;
  const uint64_t* seg_off;  // [n_seg + 1]
  uint32_t n_seg;
  uint32_t blocks_per_seg;
  // Workgroups per segment in proportion to the segment SIZES (skew: one segment may hold half of the rows): workgroup k
  // belongs to the segment seg with blk_first[seg] <= k < blk_first[seg + 1] (k_seg_blocks; [n_seg + 1] on the device);
  // null: blocks_per_seg workgroups for every segment.
  const uint32_t* blk_first;
  // SRC_TUPLES, pre-split input: block k of the grid reads tuples [blk_range[2k], blk_range[2k+1])
  // (one chunk per sender of an exchanged segment) instead of an even share of its segment
  const uint64_t* blk_range;
  uint32_t chunk_split;  // workgroups per chunk (>= 1): workgroup k reads share k % chunk_split of chunk k / chunk_split
  // SRC_TUPLES, input written by the claiming scatter (k_part_scatter_claim): segment k is
  // [k * seg_cap, k * seg_cap + min(seg_cnt[k], seg_cap)) and holds filler tuples, which are
  // recognised by their hash: a real tuple of segment k has ((hash >> chk_shift) & chk_mask)
  // == (k & chk_mask)
  const uint32_t* seg_cnt;
  uint32_t seg_cap;
  uint32_t chk_shift;
  uint32_t chk_mask;  // 0: every tuple in range is real
  // ... in the WOVEN layout (weave_shift != 0, see woven_at): tuple p of segment k lies at woven_at(k, p, ...)
  uint32_t weave_shift;
  uint32_t weave_n;
  // Bucket ORDER of a single-pass split (olap_split_tuples_u32): the bucket id read from the hash is
  // [owner | group | low] (ord_owner_bits, ord_group_bits, ord_low_bits wide); it is emitted as
  // [group | owner | low], so that everything a sub-bucket group sends to all owners is one contiguous run.
  uint32_t ord_owner_bits;
  uint32_t ord_group_bits;  // 0: natural order
  uint32_t ord_low_bits;
  // SRC_TUPLES: the tuples hold wang_hash(key) in place of the key (ClaimOut::store_hash of the pass that wrote them; read
  // by k_part_scatter_ring6 only, whose packed output never needs the key itself)
  uint32_t hashed;
  // k_part_scatter_ring6 over pre-split input (blk_range, one workgroup per chunk): the output regions, counts, dense starts and
  // leading hash bits are numbered by the CHUNK (blockIdx.x), not by the segment -- every chunk becomes a set of F
  // sub-partitions of its own, and the probe builds a partition's table from the sub-partitions of all of its segment's chunks
  uint32_t out_by_block;
};

// Bucket of a tuple in a pass: the bits [shift, shift + log2 F) of its hash, in the pass's bucket order.
__device__ __forceinline__ uint32_t pass_bucket(const PassSrc& s, uint32_t h, uint32_t shift, uint32_t F) {
  uint32_t b = (h >> shift) & (F - 1);
  if (s.ord_group_bits) {
    const uint32_t low = b & ((1u << s.ord_low_bits) - 1u);
    const uint32_t grp = (b >> s.ord_low_bits) & ((1u << s.ord_group_bits) - 1u);
    const uint32_t own = b >> (s.ord_low_bits + s.ord_group_bits);
    b = (grp << (s.ord_owner_bits + s.ord_low_bits)) | (own << s.ord_low_bits) | low;
  }
  return b;
}

// Woven regions of the claiming scatter.  Laid out one after the other, the F regions of pass 0 are 64 MiB apart at
// 2^32 rows, and every workgroup writes one 128-byte line to each of them in turn: every line store misses the CU's
// address-translation cache (rocprofv3: TCP_UTCL1_TRANSLATION_MISS = 2.9e8 per launch, one per line; the same kernel
// writing inside ONE region: 2.5e3).  Woven, the regions are cut into chunks of 2^weave_shift tuples and chunk c of region
// k lies at ((c * n_regions) + k) chunks: all regions fill at the same pace (uniform hash, global cursors), so at any
// moment the whole chip writes inside a window of about n_regions chunks (32 MiB), and the reader of region k takes one
// chunk-sized stride per 64 KiB it reads.  A tile (2048 / 4096 / 8192 tuples) never straddles a chunk.
constexpr uint32_t kWeaveShift = 13;  // 8192 tuples = 64 KiB per chunk
__device__ __forceinline__ uint64_t woven_at(uint64_t region, uint64_t p, uint32_t cap, uint32_t weave_shift, uint32_t n_regions) {
  if (!weave_shift) return region * cap + p;
  return ((((p >> weave_shift) * n_regions) + region) << weave_shift) + (p & ((1ull << weave_shift) - 1ull));
}

struct TileRange {
  uint64_t t0, t1;    // tile ids
  uint64_t lo, hi;    // SRC_TUPLES: tuple range of the segment
  uint64_t a;         // SRC_TUPLES: 16-B aligned base (lo & ~1)
  uint32_t seg;
};

template <int SRC, int TILE = kPTile>
__device__ __forceinline__ TileRange block_tiles(const PassSrc& s) {
  TileRange r;
  if (SRC == SRC_COLS) {
    r.seg = 0;
    r.t0 = (uint64_t)blockIdx.x * s.tiles_per_block;
    r.t1 = min(r.t0 + s.tiles_per_block, s.v.total_units);
    if (r.t0 > r.t1) r.t0 = r.t1;
    r.lo = r.hi = r.a = 0;
  } else {
    if (s.blk_first) {  // sized segments
      uint32_t lo = 0, hi = s.n_seg;  // last seg with blk_first[seg] <= blockIdx.x
      if (blockIdx.x >= s.blk_first[hi]) {
        r.seg = 0;
        r.t0 = r.t1 = r.lo = r.hi = r.a = 0;
        return r;  // beyond the last segment's workgroups (the grid is an upper bound)
      }
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s.blk_first[mid] <= blockIdx.x) lo = mid; else hi = mid;
      }
      r.seg = lo;
      const uint32_t first = s.blk_first[lo], nblk = s.blk_first[lo + 1] - first, j = blockIdx.x - first;
      r.lo = s.seg_off[r.seg];
      r.hi = s.seg_off[r.seg + 1];
      r.a = r.lo & ~1ull;
      const uint64_t tiles = (r.hi - r.a + TILE - 1) / TILE;
      const uint64_t q = (tiles + nblk - 1) / nblk;
      r.t0 = min((uint64_t)j * q, tiles);
      r.t1 = min(r.t0 + q, tiles);
      return r;
    }
    r.seg = blockIdx.x / s.blocks_per_seg;
    if (s.blk_range) {
      const uint32_t split = s.chunk_split ? s.chunk_split : 1u;
      const uint64_t chunk = blockIdx.x / split, part = blockIdx.x - chunk * split;
      r.lo = s.blk_range[2 * chunk];
      r.hi = s.blk_range[2 * chunk + 1];
      r.a = r.lo & ~1ull;
      const uint64_t tiles = (r.hi - r.a + TILE - 1) / TILE, q = (tiles + split - 1) / split;
      r.t0 = min(part * q, tiles);
      r.t1 = min(r.t0 + q, tiles);
      return r;
    }
    const uint32_t j = blockIdx.x - r.seg * s.blocks_per_seg;
    if (s.seg_cnt) {
      r.lo = (uint64_t)r.seg * s.seg_cap;
      r.hi = r.lo + min(s.seg_cnt[r.seg], s.seg_cap);
    } else {
      r.lo = s.seg_off[r.seg];
      r.hi = s.seg_off[r.seg + 1];
    }
    r.a = r.lo & ~1ull;
    const uint64_t tiles = (r.hi - r.a + TILE - 1) / TILE;
    const uint64_t q = (tiles + s.blocks_per_seg - 1) / s.blocks_per_seg;
    r.t0 = min((uint64_t)j * q, tiles);
    r.t1 = min(r.t0 + q, tiles);
  }
  return r;
}

// Batch-table entries of the batch the workgroup is currently reading: consecutive tiles
// mostly stay in one batch (a 2 Mi-row batch is 256 tiles), so the dependent loads
// "pointer table -> data" are paid once per batch instead of once per tile.
struct ColsCursor {
  uint32_t batch = 0xFFFFFFFFu;
  const uint32_t* kp = nullptr;
  const uint32_t* pp = nullptr;
  uint64_t rows = 0, row0 = 0;
  uint64_t tile = ~0ull;       // last tile located
  uint64_t local_unit = 0;     // its index inside the batch
  uint64_t units = 0;          // tiles of the current batch
};

// Loads one tile: kPTile / BLOCK tuples per thread, valid bits in the returned mask.
#ifndef OLAP_RING_NT_LOAD
#define OLAP_RING_NT_LOAD 0   // A/B builds (scripts/build_variant.sh): streaming `nt` loads in the ring scatter's producers
#endif
#ifndef OLAP_RING_NT_STORE
#define OLAP_RING_NT_STORE 0  // ... `nt` stores for its output lines
#endif
template <int SRC, int BLOCK, bool WANT_PAYLOAD, int TILE = kPTile, bool NT = false>
__device__ __forceinline__ uint32_t load_tile(const PassSrc& s, const TileRange& r, uint64_t tile, ColsCursor& cur,
                                              uint32_t (&key)[TILE / BLOCK], uint32_t (&pay)[TILE / BLOCK]) {
  constexpr int TPT = TILE / BLOCK;
  uint32_t mask = 0;
  if (SRC == SRC_COLS) {
    // tiles are visited in order: step the (batch, local tile) pair instead of locating
    // every tile afresh (a 64-bit division or a binary search per tile and thread)
    Loc l;
    if (tile == cur.tile + 1 && cur.local_unit + 1 < cur.units) {
      l.batch = cur.batch;
      l.local_unit = cur.local_unit + 1;
    } else {
      l = locate(s.v, tile);
    }
    cur.tile = tile;
    cur.local_unit = l.local_unit;
    if (l.batch != cur.batch) {  // workgroup-uniform
      cur.batch = l.batch;
      cur.kp = (const uint32_t*)uniform_u64((uint64_t)s.key_ptrs[l.batch]);
      cur.pp = s.pay_ptrs ? (const uint32_t*)uniform_u64((uint64_t)s.pay_ptrs[l.batch]) : nullptr;
      cur.rows = uniform_u64(batch_rows(s.v, l.batch));
      cur.row0 = uniform_u64(batch_row0(s.v, l.batch));
      cur.units = (cur.rows + TILE - 1) / TILE;
    }
    const uint64_t rows = cur.rows;
    const uint64_t r0 = l.local_unit * TILE;
    const uint32_t n = (uint32_t)min((uint64_t)TILE, rows - r0);
    const uint32_t* kp = cur.kp + r0;
    const uint32_t* pp = cur.pp ? cur.pp + r0 : nullptr;
    const uint32_t rowid0 = (uint32_t)(cur.row0 + r0);
    const bool fast = n == TILE && ((((uintptr_t)kp) | ((uintptr_t)pp)) & 15) == 0;
    if (fast) {
      uint4 kv[TPT / 4], pv[TPT / 4];
#pragma unroll
      for (int j = 0; j < TPT / 4; ++j) kv[j] = NT ? gload4_nt((const uint4*)kp + j * BLOCK + threadIdx.x) : gload4((const uint4*)kp + j * BLOCK + threadIdx.x);
      if (WANT_PAYLOAD && pp) {
#pragma unroll
        for (int j = 0; j < TPT / 4; ++j) pv[j] = NT ? gload4_nt((const uint4*)pp + j * BLOCK + threadIdx.x) : gload4((const uint4*)pp + j * BLOCK + threadIdx.x);
      }
#pragma unroll
      for (int j = 0; j < TPT / 4; ++j) {
        key[4 * j + 0] = kv[j].x;
        key[4 * j + 1] = kv[j].y;
        key[4 * j + 2] = kv[j].z;
        key[4 * j + 3] = kv[j].w;
        if (WANT_PAYLOAD) {
          if (pp) {
            pay[4 * j + 0] = pv[j].x;
            pay[4 * j + 1] = pv[j].y;
            pay[4 * j + 2] = pv[j].z;
            pay[4 * j + 3] = pv[j].w;
          } else {
            const uint32_t i0 = rowid0 + (j * BLOCK + threadIdx.x) * 4;
            pay[4 * j + 0] = i0;
            pay[4 * j + 1] = i0 + 1;
            pay[4 * j + 2] = i0 + 2;
            pay[4 * j + 3] = i0 + 3;
          }
        }
      }
      mask = (1u << TPT) - 1u;
    } else {
#pragma unroll
      for (int j = 0; j < TPT / 4; ++j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t idx = (j * BLOCK + threadIdx.x) * 4 + k;
          const bool ok = idx < n;
          key[4 * j + k] = ok ? gload1(kp + idx) : 0u;
          if (WANT_PAYLOAD) pay[4 * j + k] = ok ? (pp ? gload1(pp + idx) : rowid0 + idx) : 0u;
          mask |= (ok ? 1u : 0u) << (4 * j + k);
        }
      }
    }
  } else {
    const uint64_t base = r.a + tile * (uint64_t)TILE;
    // woven input (claimed regions: r.a == r.lo == seg * seg_cap): where the tile's first tuple really lies
    const uint2* tuples = s.tuples;
    if (s.weave_shift) tuples += (int64_t)(woven_at(r.seg, tile * (uint64_t)TILE, s.seg_cap, s.weave_shift, s.weave_n) - base);
    if (base >= r.lo && base + TILE <= r.hi) {  // whole tile inside the segment (uniform)
      const uint4* tv = (const uint4*)(tuples + base);
#pragma unroll
      for (int j = 0; j < TPT / 2; ++j) {
        const uint4 t = NT ? gload4_nt(tv + j * BLOCK + threadIdx.x) : tv[j * BLOCK + threadIdx.x];
        key[2 * j] = t.x;
        pay[2 * j] = t.y;
        key[2 * j + 1] = t.z;
        pay[2 * j + 1] = t.w;
      }
      return (1u << TPT) - 1u;
    }
#pragma unroll
    for (int j = 0; j < TPT / 2; ++j) {
      const uint64_t i0 = base + (uint64_t)(j * BLOCK + threadIdx.x) * 2;
      uint4 t = make_uint4(0, 0, 0, 0);
      if (i0 + 1 < r.hi) {
        t = *(const uint4*)(tuples + i0);
      } else if (i0 < r.hi) {  // last, odd tuple: never read past the end of a caller's array
        const uint2 t1 = tuples[i0];
        t.x = t1.x;
        t.y = t1.y;
      }
      key[2 * j] = t.x;
      pay[2 * j] = t.y;
      key[2 * j + 1] = t.z;
      pay[2 * j + 1] = t.w;
      mask |= ((i0 >= r.lo && i0 < r.hi) ? 1u : 0u) << (2 * j);
      mask |= ((i0 + 1 >= r.lo && i0 + 1 < r.hi) ? 1u : 0u) << (2 * j + 1);
    }
  }
  return mask;
}

// Column tile of WHOLE, 16-byte aligned units (the host has checked every batch: ColumnSet::full_units) straight into
// 16-byte registers: TPT / 4 key vectors and TPT / 4 payload vectors per thread (HASPAY false: payload = row index).
// One straight-line sequence of loads whose destinations ARE the tile's storage.  With load_tile's ragged / unaligned
// fallback beside it, or with scalar arrays as storage, the compiler places register copies right behind the loads --
// a wait at the spot where they are issued, which turns a prefetch two tiles ahead into a synchronous load (seen in the
// ISA of the ring scatter's column variant: 16.2 ms per launch against 13.6 ms from zipped tuples).
template <int BLOCK, int TILE, bool HASPAY, bool NT>
__device__ __forceinline__ void load_tile_full(const PassSrc& s, uint64_t tile, ColsCursor& cur, u32x4_t& k0, u32x4_t& k1, u32x4_t& p0,
                                               u32x4_t& p1) {
  static_assert(TILE / BLOCK == 8, "two key and two payload vectors per thread");
  const uint32_t* kbase;
  const uint32_t* pbase = nullptr;
  uint32_t row_first;
  if (s.contig_key) {
    // The batches are consecutive slices of ONE buffer per column (ColumnSet::contiguous; record batches cut from a
    // table): tile t starts t * TILE elements into it -- no batch lookup, no pointer table (whose dependent loads cost a
    // round trip to the L2 in front of the data loads at every change of batch).
    kbase = s.contig_key + tile * (uint64_t)TILE;
    if (HASPAY) pbase = s.contig_pay + tile * (uint64_t)TILE;
    row_first = (uint32_t)(tile * (uint64_t)TILE);
  } else {
    Loc l;
    if (tile == cur.tile + 1 && cur.local_unit + 1 < cur.units) {
      l.batch = cur.batch;
      l.local_unit = cur.local_unit + 1;
    } else {
      l = locate(s.v, tile);
    }
    cur.tile = tile;
    cur.local_unit = l.local_unit;
    if (l.batch != cur.batch) {  // workgroup-uniform
      cur.batch = l.batch;
      cur.kp = (const uint32_t*)uniform_u64((uint64_t)s.key_ptrs[l.batch]);
      if (HASPAY) cur.pp = (const uint32_t*)uniform_u64((uint64_t)s.pay_ptrs[l.batch]);
      cur.row0 = uniform_u64(batch_row0(s.v, l.batch));
      cur.units = uniform_u64(batch_rows(s.v, l.batch)) / TILE;
    }
    const uint64_t r0 = l.local_unit * TILE;
    kbase = cur.kp + r0;
    if (HASPAY) pbase = cur.pp + r0;
    row_first = (uint32_t)(cur.row0 + r0);
  }
  const OLAP_GLOBAL u32x4_t* kp = (const OLAP_GLOBAL u32x4_t*)kbase + threadIdx.x;
  k0 = NT ? __builtin_nontemporal_load(kp) : kp[0];
  k1 = NT ? __builtin_nontemporal_load(kp + BLOCK) : kp[BLOCK];
  if (HASPAY) {
    const OLAP_GLOBAL u32x4_t* pp = (const OLAP_GLOBAL u32x4_t*)pbase + threadIdx.x;
    p0 = NT ? __builtin_nontemporal_load(pp) : pp[0];
    p1 = NT ? __builtin_nontemporal_load(pp + BLOCK) : pp[BLOCK];
  } else {
    const uint32_t i0 = row_first + threadIdx.x * 4;
    p0.x = i0;
    p0.y = i0 + 1;
    p0.z = i0 + 2;
    p0.w = i0 + 3;
    p1 = p0 + (uint32_t)(BLOCK * 4);
  }
}

// Rank of a tuple inside its bucket for the current tile: cl[b]++ in LDS.  When many lanes of a wave hold the SAME bucket
// (a hot key) the hardware serialises their atomics (2^27 rows, half of them one key: histogram 0.44 -> 3.8 ms, scatter
// 1.1 -> 9.1 ms).  Such a bucket is counted once per wave instead: the lanes compare their bucket with a candidate -- the
// bucket that was hot at the previous call, else the first valid lane's -- and if >= 8 agree, one lane adds the count
// and every one of them takes base + its position among them.  A wave looks for a hot bucket once per tile (look = true on
// its first tuple slot) and keeps testing every slot while one is hot; without one the cost is a readfirstlane per slot.
struct HotGuess {
  uint32_t b = 0xFFFFFFFFu;
};
__device__ __forceinline__ uint32_t lds_rank(uint32_t* cl, uint32_t b, bool valid, HotGuess& g, bool look) {
  // no bucket known to be hot and not the call that looks for one (once per tile): the plain atomic
  const uint32_t gb = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.b);  // (one lane's guess: after a divergent edge tile they may differ)
  if (gb == 0xFFFFFFFFu && !look) return valid ? atomicAdd(&cl[b], 1u) : 0u;
  const unsigned long long act = __ballot(valid);
  if (!act) return 0;
  const int lead = __ffsll((long long)act) - 1;
  const uint32_t cand = gb != 0xFFFFFFFFu ? gb : (uint32_t)__shfl((int)b, lead, kWave);
  const bool same = valid && b == cand;
  const unsigned long long m = __ballot(same);
  uint32_t r = 0;
  if (__popcll(m) >= 8) {
    const int first = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane_id() == first) base = atomicAdd(&cl[cand], (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, first, kWave);
    g.b = cand;
    if (same) r = base + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
    else if (valid) r = atomicAdd(&cl[b], 1u);
  } else {
    g.b = 0xFFFFFFFFu;  // not hot (any more)
    if (valid) r = atomicAdd(&cl[b], 1u);
  }
  return r;
}

// ------------------------------------------------------------------ hist
template <int SRC, int BLOCK, int TILE = kPTile>
__global__ __launch_bounds__(BLOCK) void k_part_hist(PassSrc s, uint32_t shift, uint32_t fbits,
                                                     uint32_t* hist) {
  constexpr int TPT = TILE / BLOCK;
  __shared__ uint32_t cl[kMaxF];
  const uint32_t F = 1u << fbits;
  const TileRange r = block_tiles<SRC, TILE>(s);
  if (r.t0 >= r.t1) {  // nothing to read: a row of zeros
    for (uint32_t b = threadIdx.x; b < F; b += BLOCK) hist[(uint64_t)blockIdx.x * F + b] = 0;
    return;
  }
  for (uint32_t b = threadIdx.x; b < F; b += BLOCK) cl[b] = 0;
  __syncthreads();
  ColsCursor cur;
  HotGuess hot;
  for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
    uint32_t key[TPT], pay[TPT];
    const uint32_t mask = load_tile<SRC, BLOCK, false, TILE>(s, r, tile, cur, key, pay);
#pragma unroll
    for (int i = 0; i < TPT; ++i) (void)lds_rank(cl, pass_bucket(s, wang_hash(key[i]), shift, F), (mask >> i) & 1u, hot, i == 0);
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < F; b += BLOCK) hist[(uint64_t)blockIdx.x * F + b] = cl[b];
}

// ------------------------------------------------------------------ scan
// One workgroup per input segment.  hist[(seg*Bseg + blk)*F + b] (counts) is
// replaced by the exclusive prefix over blk; seg_off_out[seg*F + b] receives the
// absolute start of output bucket (seg, b) -- the segment table of the next pass.
__global__ __launch_bounds__(1024) void k_part_scan(uint32_t* hist, const uint64_t* seg_off_in,
                                                    uint64_t total_in, uint32_t n_seg, uint32_t Bseg_all,
                                                    uint32_t fbits, uint64_t* seg_off_out, const uint32_t* blk_first = nullptr) {
  __shared__ uint64_t s_wave[16];
  const uint32_t F = 1u << fbits;
  const uint32_t seg = blockIdx.x;
  const uint32_t b = threadIdx.x;
  // the segment's workgroups: Bseg_all each, or the sized ranges of k_seg_blocks
  const uint32_t first_blk = blk_first ? blk_first[seg] : seg * Bseg_all;
  const uint32_t Bseg = blk_first ? blk_first[seg + 1] - first_blk : Bseg_all;
  uint64_t tot = 0;
  if (b < F) {
    uint32_t run = 0;
    uint32_t* col = hist + (uint64_t)first_blk * F + b;
    uint32_t blk = 0;
    for (; blk + 4 <= Bseg; blk += 4) {  // 4 independent loads in flight
      const uint32_t c0 = col[(uint64_t)(blk + 0) * F], c1 = col[(uint64_t)(blk + 1) * F];
      const uint32_t c2 = col[(uint64_t)(blk + 2) * F], c3 = col[(uint64_t)(blk + 3) * F];
      col[(uint64_t)(blk + 0) * F] = run;
      col[(uint64_t)(blk + 1) * F] = run + c0;
      col[(uint64_t)(blk + 2) * F] = run + c0 + c1;
      col[(uint64_t)(blk + 3) * F] = run + c0 + c1 + c2;
      tot += (uint64_t)c0 + c1 + c2 + c3;
      run += c0 + c1 + c2 + c3;
    }
    for (; blk < Bseg; ++blk) {
      const uint32_t c = col[(uint64_t)blk * F];
      col[(uint64_t)blk * F] = run;
      run += c;
      tot += c;
    }
  }
  uint64_t incl = tot;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint64_t y = __shfl_up(incl, o, kWave);
    if ((int)lane_id() >= o) incl += y;
  }
  if (lane_id() == kWave - 1) s_wave[threadIdx.x / kWave] = incl;
  __syncthreads();
  uint64_t off = seg_off_in ? seg_off_in[seg] : 0;
  for (unsigned w = 0; w < threadIdx.x / kWave; ++w) off += s_wave[w];
  if (b < F) seg_off_out[(uint64_t)seg * F + b] = off + incl - tot;
  if (seg == n_seg - 1 && threadIdx.x == 0)
    seg_off_out[(uint64_t)n_seg * F] = seg_off_in ? seg_off_in[n_seg] : total_in;
}

// Workgroups per segment by size: segment seg gets max(1, ceil(tiles(seg) / share)) of them, share = the tiles a workgroup
// would take if all segments were equal; blk_first = exclusive prefix ([n_seg + 1]).  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_seg_blocks(const uint64_t* seg_off, uint32_t n_seg, uint32_t tile, uint32_t share,
                                                     uint32_t* blk_first) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_seg; base += 1024) {
    const uint32_t seg = base + threadIdx.x;
    uint32_t n = 0;
    if (seg < n_seg) {
      const uint64_t lo = seg_off[seg], hi = seg_off[seg + 1], a = lo & ~1ull;
      const uint64_t tiles = (hi - a + tile - 1) / tile;
      n = (uint32_t)max((uint64_t)1, (tiles + share - 1) / share);
    }
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o, kWave);
      if ((int)lane_id() >= o) incl += y;
    }
    if (lane_id() == kWave - 1) s_wave[threadIdx.x / kWave] = incl;
    __syncthreads();
    uint32_t off = s_carry;
    for (unsigned w = 0; w < threadIdx.x / kWave; ++w) off += s_wave[w];
    if (seg < n_seg) blk_first[seg] = off + incl - n;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) blk_first[n_seg] = s_carry;
}

// --------------------------------------------------------------- scatter
template <int SRC>
__global__ __launch_bounds__(kPBlock, 4) void k_part_scatter(PassSrc s, uint32_t shift, uint32_t fbits,
                                                             const uint32_t* hist,
                                                             const uint64_t* seg_off_out, uint2* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint2* stage = (uint2*)smem;                                          // kPTile tuples
  uint64_t* gcur = (uint64_t*)(smem + kPTile * 8);                       // F cursors
  uint32_t* cl = (uint32_t*)(smem + kPTile * 8 + kMaxF * 8);             // F + 1 counts / local starts
  uint32_t* wtot = cl + (kMaxF + 4);                                     // 8 wave totals
  const uint32_t F = 1u << fbits;
  const uint32_t BPT = (F + kPBlock - 1) / kPBlock;  // bins per thread: 1 or 2
  const TileRange r = block_tiles<SRC>(s);
  for (uint32_t b = threadIdx.x; b < F; b += kPBlock) {
    gcur[b] = seg_off_out[(uint64_t)r.seg * F + b] + hist[(uint64_t)blockIdx.x * F + b];
    cl[b] = 0;
  }
  __syncthreads();
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x / kWave;
  ColsCursor cur;
  for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
    uint32_t key[kTPT], pay[kTPT], br[kTPT];
    const uint32_t mask = load_tile<SRC, kPBlock, true>(s, r, tile, cur, key, pay);
    // (2) bucket + rank inside the bucket
#pragma unroll
    for (int i = 0; i < kTPT; ++i) {
      br[i] = 0;
      if (mask & (1u << i)) {
        const uint32_t b = pass_bucket(s, wang_hash(key[i]), shift, F);
        br[i] = (b << 16) | atomicAdd(&cl[b], 1u);
      }
    }
    __syncthreads();
    // (3) exclusive scan of the bucket counts, in place; counts stay in registers
    uint32_t cnt[2] = {0, 0};
    uint32_t local = 0;
    const uint32_t bin0 = threadIdx.x * BPT;
    for (uint32_t q = 0; q < BPT; ++q) {
      if (bin0 + q < F) {
        cnt[q] = cl[bin0 + q];
        local += cnt[q];
      }
    }
    uint32_t incl = local;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o, kWave);
      if ((int)lane >= o) incl += y;
    }
    if (lane == kWave - 1) wtot[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - local;
    uint32_t n_tile = 0;
#pragma unroll
    for (int w = 0; w < kPBlock / kWave; ++w) {
      if ((unsigned)w < wave) pre += wtot[w];
      n_tile += wtot[w];
    }
    for (uint32_t q = 0; q < BPT; ++q) {
      if (bin0 + q < F) {
        cl[bin0 + q] = pre;
        pre += cnt[q];
      }
    }
    __syncthreads();
    // (4) tuples into the staging area, grouped by bucket
#pragma unroll
    for (int i = 0; i < kTPT; ++i) {
      if (mask & (1u << i)) stage[cl[br[i] >> 16] + (br[i] & 0xFFFFu)] = make_uint2(key[i], pay[i]);
    }
    __syncthreads();
    // (5) coalesced copy-out: consecutive lanes -> consecutive addresses inside a run
#pragma unroll 4
    for (uint32_t i = threadIdx.x; i < n_tile; i += kPBlock) {
      const uint2 t = stage[i];
      const uint32_t b = pass_bucket(s, wang_hash(t.x), shift, F);
      out[gcur[b] + (i - cl[b])] = t;
    }
    __syncthreads();
    // (6) advance the cursors, clear the counts
    for (uint32_t q = 0; q < BPT; ++q) {
      if (bin0 + q < F) {
        gcur[bin0 + q] += cnt[q];
        cl[bin0 + q] = 0;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------- scatter, write-combining variant
// Same contract as k_part_scatter, but every global store is a whole, aligned
// 64-byte line (8 tuples = one fabric request): per bucket the workgroup keeps the
// < 8 tuples that do not fill a line yet in LDS ("residue") and emits them together
// with the next tile's tuples.  Only the first and the last line of a
// (workgroup, bucket) region can be partial.  1024 threads, one workgroup per CU
// (LDS: 64 KiB staging + 56 KiB residues + tables); the next tile's loads are
// issued before the copy-out of the current one.
//
// Per bucket (bm[b]): x = start of the bucket's lines in the virtual flush index
// space of this tile, y = start of the bucket's NEW tuples in `stage`,
// (z,w) = 64-bit word: absolute, 8-aligned index of residue slot 0 | residue
// count << 56 | leading slots that belong to the previous workgroup << 60.
constexpr int kWBlock = 1024;

// LINE = tuples per emitted line: 8 (64 B) for up to 1024 buckets, 16 (128 B, a whole L2
// line; scattered 128-B lines reach ~4.8 TB/s against ~3.6 TB/s for 64-B ones,
// profiles/r1_ubench_rand_chunk_write.txt) for up to 512 buckets -- the LDS residue buffer
// F * (LINE - 1) * 8 B is what bounds the fan-out.
// Geometry: BLOCK threads work on tiles of TILE tuples; BLOCK >= max buckets (one bucket per
// owner thread).  {1024, 8192}: one workgroup per CU, up to 1024 buckets.  {512, 4096}: two
// workgroups per CU (their phases overlap), up to 512 buckets.
template <int LINE, int BLOCK = kWBlock, int TILE = kPTile>
struct WcLds {
  static constexpr int kRes = LINE - 1;
  static constexpr int kFmax = (LINE == 8 ? 1024 : 512) < BLOCK ? (LINE == 8 ? 1024 : 512) : BLOCK;
  static constexpr uint32_t kStage = 0;
  static constexpr uint32_t kResOff = kStage + TILE * 8;
  static constexpr uint32_t kBm = kResOff + kRes * kFmax * 8;
  static constexpr uint32_t kCl = kBm + kFmax * 16;
  static constexpr uint32_t kL2b = kCl + (kFmax + 4) * 4;
  static constexpr uint32_t kWtot = kL2b + ((TILE + kRes * kFmax) / LINE + 2) * 2;
  static constexpr uint32_t kBytes = ((kWtot + 15) / 16) * 16 + (BLOCK / kWave) * 8 + 64;
};
constexpr int kLineAlign = 16;  // capacity regions are aligned for either line size

// Capacity mode (CAPM, last pass only, one workgroup per input segment): no histogram
// pre-pass.  Output bucket (seg, b) owns the fixed region [p * cap, (p + 1) * cap) with
// p = seg * F + b; the workgroup reports how many tuples landed there (cnt[p]) and where
// the partition starts in the dense numbering (dense[p]).  A bucket that would outgrow
// its region raises *overflow and drops the excess; the host then reruns the join with
// exact (histogram) partitioning.
struct CapOut {
  uint32_t cap;       // tuples per region, multiple of kLineAlign
  uint32_t* cnt;      // [n_seg * F]
  uint64_t* dense;    // [n_seg * F]
  unsigned long long* overflow;
  uint64_t* seg_total;  // non-null: the input had fillers, so the segment's dense start is unknown here --
                        // dense[] receives the start inside the segment, seg_total[seg] the segment's
                        // tuple count, and k_dense_fix adds the segment bases afterwards
  // k_part_scatter_ring<.., EXACT>: the exact layout instead (histogram + scan ran before): workgroup blk writes its tuples of
  // bucket b from tuple seg_off[seg * F + b] + hist[blk * F + b] on, as k_part_scatter_wc does
  const uint32_t* hist;
  const uint64_t* seg_off;
};

template <int SRC, bool CAPM, int LINE, int BLOCK = kWBlock, int TILE = kPTile>
__global__ __launch_bounds__(BLOCK) void k_part_scatter_wc(PassSrc s, uint32_t shift, uint32_t fbits,
                                                             const uint32_t* hist,
                                                             const uint64_t* seg_off_out, uint2* out,
                                                             CapOut co) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using LDS = WcLds<LINE, BLOCK, TILE>;
  constexpr int kWTPT = TILE / BLOCK;
  constexpr int kLine = LINE, kRes = LINE - 1;
  constexpr uint32_t kFld = 15u;  // residue / skip counts: 4-bit fields at bits 56 and 60
  uint2* stage = (uint2*)(smem + LDS::kStage);
  uint2* res = (uint2*)(smem + LDS::kResOff);        // [bucket * kRes + slot]
  uint4* bm = (uint4*)(smem + LDS::kBm);
  uint32_t* cl = (uint32_t*)(smem + LDS::kCl);
  uint16_t* l2b = (uint16_t*)(smem + LDS::kL2b);     // flush line -> bucket
  uint64_t* wtot = (uint64_t*)(smem + ((LDS::kWtot + 15) / 16) * 16);  // per-wave (count | flush << 32)
  const uint32_t F = 1u << fbits;
  const TileRange r = block_tiles<SRC, TILE>(s);
  if (!CAPM && r.t0 >= r.t1) return;  // exact layout: a workgroup without tiles owns no part of the output (uniform)
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x / kWave;
  const uint32_t b_own = threadIdx.x;  // BLOCK >= kMaxF: one bucket per thread
  const bool owner = b_own < F;
  uint64_t region = 0;  // CAPM: first tuple index of this thread's bucket region
  if (owner) {
    uint64_t off;
    if (CAPM) {
      region = ((uint64_t)r.seg * F + b_own) * co.cap;
      off = region;
    } else {
      off = seg_off_out[(uint64_t)r.seg * F + b_own] + hist[(uint64_t)blockIdx.x * F + b_own];
    }
    const uint64_t a = off & (kLine - 1);
    const uint64_t w = (off - a) | (a << 56) | (a << 60);
    bm[b_own] = make_uint4(0u, 0u, (uint32_t)w, (uint32_t)(w >> 32));
    cl[b_own] = 0;
  }
  __syncthreads();
  uint32_t key[kWTPT], pay[kWTPT], nkey[kWTPT], npay[kWTPT];
  uint32_t mask = 0, nmask = 0;
  ColsCursor cur;
  HotGuess hot;
  if (r.t0 < r.t1) mask = load_tile<SRC, BLOCK, true, TILE>(s, r, r.t0, cur, key, pay);
  for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
    // (2) bucket + rank of every tuple
    uint32_t br[kWTPT];
    constexpr uint32_t kFull = (1u << kWTPT) - 1u;
    if (mask == kFull && !s.chk_mask) {  // whole tile valid (all but the edge tiles): no per-tuple predication
#pragma unroll
      for (int i = 0; i < kWTPT; ++i) {
        const uint32_t b = pass_bucket(s, wang_hash(key[i]), shift, F);
        br[i] = (b << 16) | lds_rank(cl, b, true, hot, i == 0);
      }
    } else {
      const uint32_t expect = r.seg & s.chk_mask;
#pragma unroll
      for (int i = 0; i < kWTPT; ++i) {
        const uint32_t h = wang_hash(key[i]);
        if (((h >> s.chk_shift) & s.chk_mask) != expect) mask &= ~(1u << i);  // a filler of the claiming scatter
        const bool valid = (mask >> i) & 1u;
        const uint32_t b = pass_bucket(s, h, shift, F);
        const uint32_t rk = lds_rank(cl, b, valid, hot, i == 0);
        br[i] = valid ? (b << 16) | rk : 0u;
      }
    }
    __syncthreads();
    // (3) per bucket: count c, residue rr, lines to flush f; scans of (c, f)
    uint32_t c = 0, c_app = 0, rr = 0, f = 0;
    uint64_t w64 = 0;
    if (owner) {
      c = cl[b_own];
      const uint4 m = bm[b_own];
      w64 = ((uint64_t)m.w << 32) | m.z;
      rr = (uint32_t)(w64 >> 56) & kFld;
      c_app = c;
      if (CAPM && (w64 & 0x00FFFFFFFFFFFFFFull) - region + rr + c > co.cap) {
        c_app = 0;  // region full: drop this tile's tuples of the bucket, report
        if (c) atomicAdd(co.overflow, 1ull);
      }
      f = (rr + c_app) & ~(uint32_t)(kLine - 1);
    }
    const uint64_t pair = (uint64_t)c | ((uint64_t)f << 32);
    uint64_t incl = pair;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint64_t yv = __shfl_up(incl, o, kWave);
      if ((int)lane >= o) incl += yv;
    }
    if (lane == kWave - 1) wtot[wave] = incl;
    __syncthreads();
    uint64_t pre = incl - pair, tot = 0;
#pragma unroll 1
    for (int w = 0; w < BLOCK / kWave; ++w) {
      const uint64_t t = wtot[w];
      if ((unsigned)w < wave) pre += t;
      tot += t;
    }
    const uint32_t lstart = (uint32_t)pre, vstart = (uint32_t)(pre >> 32);
    const uint32_t V = (uint32_t)(tot >> 32);
    if (owner) {
      bm[b_own] = make_uint4(vstart, lstart, (uint32_t)w64, (uint32_t)(w64 >> 32));
      cl[b_own] = lstart;  // consecutive buckets -> consecutive banks for the scatter below
      for (uint32_t q = 0; q < f / kLine; ++q) l2b[vstart / kLine + q] = (uint16_t)b_own;
    }
    __syncthreads();
    // (4) new tuples into the staging area, grouped by bucket, rank order
    if (mask == kFull) {
#pragma unroll
      for (int i = 0; i < kWTPT; ++i) stage[cl[br[i] >> 16] + (br[i] & 0xFFFFu)] = make_uint2(key[i], pay[i]);
    } else {
#pragma unroll
      for (int i = 0; i < kWTPT; ++i) {
        if (mask & (1u << i)) stage[cl[br[i] >> 16] + (br[i] & 0xFFFFu)] = make_uint2(key[i], pay[i]);
      }
    }
    // next tile's loads fly during the copy-out
    nmask = 0;
    if (tile + 1 < r.t1) nmask = load_tile<SRC, BLOCK, true, TILE>(s, r, tile + 1, cur, nkey, npay);
    __syncthreads();
    // (5) copy-out: 4 consecutive lanes write one aligned 64-byte line, 16 B (2 tuples) each
#pragma unroll 1
    for (uint32_t v = threadIdx.x * 2; v < V; v += BLOCK * 2) {
      const uint32_t b = l2b[v / kLine];
      const uint4 m = bm[b];
      const uint64_t w = ((uint64_t)m.w << 32) | m.z;
      const uint32_t j = v - m.x;  // even
      const uint32_t r_old = (uint32_t)(w >> 56) & kFld, skip = (uint32_t)(w >> 60) & kFld;
      const uint2 t0 = j < r_old ? res[b * kRes + j] : stage[m.y + (j - r_old)];
      const uint2 t1 = j + 1 < r_old ? res[b * kRes + j + 1] : stage[m.y + (j + 1 - r_old)];
      uint2* dst = out + (w & 0x00FFFFFFFFFFFFFFull) + j;
      if (j >= skip) {
        *(uint4*)dst = make_uint4(t0.x, t0.y, t1.x, t1.y);
      } else if (j + 1 >= skip) {
        dst[1] = t1;
      }
    }
    __syncthreads();
    // (6) what is left of each bucket becomes its residue
    if (owner) {
      uint64_t g = w64 & 0x00FFFFFFFFFFFFFFull;
      uint64_t skip = (w64 >> 60) & kFld;
      uint32_t r_new;
      if (f) {
        const uint32_t nf = f - rr;  // new tuples that were flushed
        r_new = rr + c_app - f;
        for (uint32_t q = 0; q < r_new; ++q) res[b_own * kRes + q] = stage[lstart + nf + q];
        g += f;
        skip = 0;
      } else {
        for (uint32_t q = 0; q < c_app; ++q) res[b_own * kRes + rr + q] = stage[lstart + q];
        r_new = rr + c_app;
      }
      const uint64_t w = g | ((uint64_t)r_new << 56) | (skip << 60);
      bm[b_own].z = (uint32_t)w;
      bm[b_own].w = (uint32_t)(w >> 32);
      cl[b_own] = 0;
    }
    __syncthreads();
    mask = nmask;
#pragma unroll
    for (int i = 0; i < kWTPT; ++i) {
      key[i] = nkey[i];
      pay[i] = npay[i];
    }
  }
  // tail: the last, partial line of every bucket
  uint64_t count = 0;
  if (owner) {
    const uint4 m = bm[b_own];
    const uint64_t w = ((uint64_t)m.w << 32) | m.z;
    const uint32_t r_old = (uint32_t)(w >> 56) & kFld, skip = (uint32_t)(w >> 60) & kFld;
    for (uint32_t j = skip; j < r_old; ++j) out[(w & 0x00FFFFFFFFFFFFFFull) + j] = res[b_own * kRes + j];
    count = (w & 0x00FFFFFFFFFFFFFFull) - region + r_old;
  }
  if (CAPM) {  // partition sizes and dense starts (exclusive scan over this segment's buckets)
    uint64_t incl = count;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint64_t yv = __shfl_up(incl, o, kWave);
      if ((int)lane >= o) incl += yv;
    }
    __syncthreads();
    if (lane == kWave - 1) wtot[wave] = incl;
    __syncthreads();
    uint64_t pre = incl - count;
#pragma unroll 1
    for (unsigned w = 0; w < wave; ++w) pre += wtot[w];
    if (owner) {
      co.cnt[(uint64_t)r.seg * F + b_own] = (uint32_t)count;
      co.dense[(uint64_t)r.seg * F + b_own] = (co.seg_total ? 0 : r.lo) + pre;
    }
    if (co.seg_total && threadIdx.x == 0) {
      uint64_t total = 0;
      for (int w = 0; w < BLOCK / kWave; ++w) total += wtot[w];
      co.seg_total[r.seg] = total;
    }
  }
}

// Claiming scatter: a pass WITHOUT a histogram that any number of workgroups can run on
// the same buckets.  Bucket (seg, b) owns the fixed region [p * cap, (p + 1) * cap),
// p = seg * F + b.  A workgroup writes a bucket's lines into an allotment of kClaimChunk
// tuples it claimed with ONE global atomic on the bucket's cursor (2^32 rows -> 32 M atomics,
// spread over the buckets), and claims the next when it runs out -- a tile's flush may
// straddle the old and the new allotment.  At the end the unused rest of each workgroup's
// last allotment is padded with filler tuples: a key of ANOTHER bucket, which the consumer
// pass recognises by its hash (PassSrc::chk_*), so regions need no per-chunk bookkeeping.
// A region that would outgrow cap raises *overflow (the host reruns with exact passes).
struct ClaimOut {
  uint32_t cap;      // tuples per region, multiple of kClaimChunk
  uint32_t* cursor;  // [n_seg * F], zeroed before the launch: tuples claimed per region
  unsigned long long* overflow;
  uint32_t weave_shift;  // != 0: woven regions (woven_at; cap is then a multiple of the chunk); ring scatter only
  uint32_t weave_n;      // n_seg * F
  uint32_t store_hash;   // ring scatter only: the tuples written hold wang_hash(key) in place of the key (the hash is a bijection;
                         // the consumer -- the packed last pass, pack6.h -- then skips its own hashing)
};
#ifndef OLAP_CLAIM_CHUNK
#define OLAP_CLAIM_CHUNK 128
#endif
constexpr uint32_t kClaimChunk = OLAP_CLAIM_CHUNK;  // tuples per claimed allotment (a multiple of 16)

template <int SRC, int BLOCK, int TILE>
__global__ __launch_bounds__(BLOCK) void k_part_scatter_claim(PassSrc s, uint32_t shift, uint32_t fbits, uint2* out,
                                                                ClaimOut co) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using LDS = WcLds<8, BLOCK, TILE>;
  constexpr int kWTPT = TILE / BLOCK;
  constexpr int kLine = 8, kRes = 7;
  uint2* stage = (uint2*)(smem + LDS::kStage);
  uint2* res = (uint2*)(smem + LDS::kResOff);
  // bm[b]: x = vstart | lstart << 16 (this tile), y = tuple offset of the bucket's next line in
  // its region, z = offset of the allotment claimed in this tile, w = tuples left in the
  // current allotment | residue count << 16
  uint4* bm = (uint4*)(smem + LDS::kBm);
  uint32_t* cl = (uint32_t*)(smem + LDS::kCl);
  uint16_t* l2b = (uint16_t*)(smem + LDS::kL2b);
  uint64_t* wtot = (uint64_t*)(smem + ((LDS::kWtot + 15) / 16) * 16);
  const uint32_t F = 1u << fbits;
  const TileRange r = block_tiles<SRC, TILE>(s);
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x / kWave;
  const uint32_t b_own = threadIdx.x;
  const bool owner = b_own < F;
  const uint64_t p_own = (uint64_t)r.seg * F + b_own;
  if (owner) {
    bm[b_own] = make_uint4(0u, 0u, 0u, 0u);
    cl[b_own] = 0;
  }
  __syncthreads();
  uint32_t key[kWTPT], pay[kWTPT], nkey[kWTPT], npay[kWTPT];
  uint32_t mask = 0, nmask = 0;
  ColsCursor cur;
  if (r.t0 < r.t1) mask = load_tile<SRC, BLOCK, true, TILE>(s, r, r.t0, cur, key, pay);
  for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
    uint32_t br[kWTPT];
    const uint32_t expect = r.seg & s.chk_mask;
#pragma unroll
    for (int i = 0; i < kWTPT; ++i) {
      br[i] = 0;
      const uint32_t h = wang_hash(key[i]);
      if (((h >> s.chk_shift) & s.chk_mask) != expect) mask &= ~(1u << i);
      if (mask & (1u << i)) {
        const uint32_t b = pass_bucket(s, h, shift, F);
        br[i] = (b << 16) | atomicAdd(&cl[b], 1u);
      }
    }
    __syncthreads();
    uint32_t c = 0, c_app = 0, rr = 0, f = 0, pos = 0, rem = 0, pos_new = 0, claimed = 0;
    if (owner) {
      c = cl[b_own];
      const uint4 m = bm[b_own];
      pos = m.y;
      rem = m.w & 0xFFFFu;
      rr = (m.w >> 16) & 15u;
      c_app = c;
      f = (rr + c) & ~(uint32_t)(kLine - 1);
      if (f > rem) {
        const uint32_t need = (f - rem + kClaimChunk - 1) / kClaimChunk * kClaimChunk;
        const uint32_t got = atomicAdd(&co.cursor[p_own], need);
        if ((uint64_t)got + need > co.cap) {  // region full: drop this tile's tuples of the bucket, report
          atomicAdd(co.overflow, 1ull);
          c_app = 0;
          f = 0;
        } else {
          pos_new = got;
          claimed = need;
        }
      }
    }
    const uint64_t pair = (uint64_t)c | ((uint64_t)f << 32);
    uint64_t incl = pair;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint64_t yv = __shfl_up(incl, o, kWave);
      if ((int)lane >= o) incl += yv;
    }
    if (lane == kWave - 1) wtot[wave] = incl;
    __syncthreads();
    uint64_t pre = incl - pair, tot = 0;
#pragma unroll 1
    for (int w = 0; w < BLOCK / kWave; ++w) {
      const uint64_t t = wtot[w];
      if ((unsigned)w < wave) pre += t;
      tot += t;
    }
    const uint32_t lstart = (uint32_t)pre, vstart = (uint32_t)(pre >> 32);
    const uint32_t V = (uint32_t)(tot >> 32);
    if (owner) {
      bm[b_own] = make_uint4(vstart | (lstart << 16), pos, pos_new, rem | (rr << 16));
      cl[b_own] = lstart;
      for (uint32_t q = 0; q < f / kLine; ++q) l2b[vstart / kLine + q] = (uint16_t)b_own;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kWTPT; ++i) {
      if (mask & (1u << i)) stage[cl[br[i] >> 16] + (br[i] & 0xFFFFu)] = make_uint2(key[i], pay[i]);
    }
    nmask = 0;
    if (tile + 1 < r.t1) nmask = load_tile<SRC, BLOCK, true, TILE>(s, r, tile + 1, cur, nkey, npay);
    __syncthreads();
    // copy-out: 4 consecutive lanes write one aligned 64-byte line; a bucket's lines go to the
    // rest of its old allotment first, then to the one claimed in this tile
#pragma unroll 1
    for (uint32_t v = threadIdx.x * 2; v < V; v += BLOCK * 2) {
      const uint32_t b = l2b[v / kLine];
      const uint4 m = bm[b];
      const uint32_t j = v - (m.x & 0xFFFFu);  // even
      const uint32_t ls = m.x >> 16, r_old = (m.w >> 16) & 15u, rem_old = m.w & 0xFFFFu;
      const uint2 t0 = j < r_old ? res[b * kRes + j] : stage[ls + (j - r_old)];
      const uint2 t1 = j + 1 < r_old ? res[b * kRes + j + 1] : stage[ls + (j + 1 - r_old)];
      const uint32_t off = j < rem_old ? m.y + j : m.z + (j - rem_old);
      *(uint4*)(out + ((uint64_t)r.seg * F + b) * co.cap + off) = make_uint4(t0.x, t0.y, t1.x, t1.y);
    }
    __syncthreads();
    if (owner) {
      uint32_t r_new;
      if (f) {
        const uint32_t nf = f - rr;
        r_new = rr + c_app - f;
        for (uint32_t q = 0; q < r_new; ++q) res[b_own * kRes + q] = stage[lstart + nf + q];
        if (f > rem) {
          pos = pos_new + (f - rem);
          rem = claimed - (f - rem);
        } else {
          pos += f;
          rem -= f;
        }
      } else {
        for (uint32_t q = 0; q < c_app; ++q) res[b_own * kRes + rr + q] = stage[lstart + q];
        r_new = rr + c_app;
      }
      bm[b_own].y = pos;
      bm[b_own].w = rem | (r_new << 16);
      cl[b_own] = 0;
    }
    __syncthreads();
    mask = nmask;
#pragma unroll
    for (int i = 0; i < kWTPT; ++i) {
      key[i] = nkey[i];
      pay[i] = npay[i];
    }
  }
  // tail: the residue goes out as one padded line, the rest of the allotment as fillers
  if (owner) {
    const uint4 m = bm[b_own];
    uint32_t pos = m.y, rem = m.w & 0xFFFFu;
    const uint32_t rr = (m.w >> 16) & 15u;
    uint32_t filler = 0;
    while (pass_bucket(s, wang_hash(filler), shift, F) == b_own) ++filler;  // a key of another bucket
    uint2* region = out + p_own * co.cap;
    bool ok = true;
    if (rr && rem == 0) {
      const uint32_t got = atomicAdd(&co.cursor[p_own], kClaimChunk);
      if ((uint64_t)got + kClaimChunk > co.cap) {
        atomicAdd(co.overflow, 1ull);
        ok = false;
      } else {
        pos = got;
        rem = kClaimChunk;
      }
    }
    if (ok) {
      if (rr) {
        for (uint32_t j = 0; j < (uint32_t)kLine; ++j) region[pos + j] = j < rr ? res[b_own * kRes + j] : make_uint2(filler, 0u);
        pos += kLine;
        rem -= kLine;
      }
      for (uint32_t q = 0; q < rem; q += 2) *(uint4*)(region + pos + q) = make_uint4(filler, 0u, filler, 0u);
    }
  }
}

// Second half of the dense numbering when the input had fillers: adds to dense[seg * F + b]
// (start inside the segment) the tuples of all earlier segments.  One workgroup per segment.
__global__ __launch_bounds__(1024) void k_dense_fix(const uint64_t* seg_total, uint32_t F, uint64_t* dense) {
  __shared__ uint64_t s_wave[16];
  uint64_t part = 0;
  for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 1024) part += seg_total[i];
  part = wave_reduce_add(part);
  if (lane_id() == 0) s_wave[threadIdx.x / kWave] = part;
  __syncthreads();
  uint64_t base = 0;
  for (int w = 0; w < 1024 / kWave; ++w) base += s_wave[w];
  for (uint32_t b = threadIdx.x; b < F; b += 1024) dense[(uint64_t)blockIdx.x * F + b] += base;
}

__global__ __launch_bounds__(256) void k_unzip(const uint2* t, uint64_t n, uint32_t* keys, uint32_t* pay) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint2 v = t[i];
    keys[i] = v.x;
    pay[i] = v.y;
  }
}

constexpr int kW2Block = 512;   // geometry 2 of the write-combining scatter: two workgroups per CU
constexpr int kW2Tile = 4096;
constexpr uint32_t kWc2Lds = WcLds<8, kW2Block, kW2Tile>::kBytes;

}  // namespace olap
