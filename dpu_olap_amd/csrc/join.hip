// join.hip -- radix partition and radix-partitioned PK-FK hash join for gfx950.
//
// Reference behaviour (upmem/dpu_olap):
//   partition  dpu/shared/kernels/partition.c:20-28 (wang hash), :45-46 (bucket =
//              top bits), :67-92 histogram, :94-137 prefix sum, :233-294 scatter
//   shuffle    host/partition/partitioner.cc:280-375 (through host DRAM)
//   build      dpu/shared/kernels/hash_build.c:9-35 + hashtable.c:89-165
//   probe      dpu/shared/kernels/hash_probe.c:9-46 + hashtable.c:167-192
//   compose    host/join/join_dpu.cc:168-400
//
// MI355X design: tuples (key, payload) are carried through 1..4 MSD radix passes
// over wang_hash(key) until a partition's build side fits an LDS hash table
// (8192 slots, 64 KiB, two workgroups per CU).  Each pass is
//   hist    (per-workgroup LDS histogram -> matrix [workgroup][bucket])
//   scan    (bucket-major exclusive scan -> cursors, next segment table)
//   scatter (tile of 8192 tuples: LDS ranks -> LDS staging grouped by bucket ->
//            coalesced runs written at the workgroup's cursors)
// so the layout is exact (no capacity slack, no global atomics) and the shuffle
// of the reference never leaves HBM.  The payload travels with the key, which
// removes the row-id round trip of the reference (kernel_take after partition).
// The hash table uses the LOW bits of the same hash, like the reference
// (hashtable.c:99-101), linear probing, LDS atomicCAS insert.
// Bound: HBM bandwidth (integer work, no MFMA).
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
  // SRC_TUPLES
  const uint2* tuples;
  const uint64_t* seg_off;  // [n_seg + 1]
  uint32_t n_seg;
  uint32_t blocks_per_seg;
  // SRC_TUPLES, pre-split input: block k of the grid reads tuples [blk_range[2k], blk_range[2k+1])
  // (one chunk per sender of an exchanged segment) instead of an even share of its segment
  const uint64_t* blk_range;
  // SRC_TUPLES, input written by the claiming scatter (k_part_scatter_claim): segment k is
  // [k * seg_cap, k * seg_cap + min(seg_cnt[k], seg_cap)) and holds filler tuples, which are
  // recognised by their hash: a real tuple of segment k has ((hash >> chk_shift) & chk_mask)
  // == (k & chk_mask)
  const uint32_t* seg_cnt;
  uint32_t seg_cap;
  uint32_t chk_shift;
  uint32_t chk_mask;  // 0: every tuple in range is real
};

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
    r.seg = blockIdx.x / s.blocks_per_seg;
    if (s.blk_range) {
      r.lo = s.blk_range[2 * (uint64_t)blockIdx.x];
      r.hi = s.blk_range[2 * (uint64_t)blockIdx.x + 1];
      r.a = r.lo & ~1ull;
      r.t0 = 0;
      r.t1 = (r.hi - r.a + TILE - 1) / TILE;
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
template <int SRC, int BLOCK, bool WANT_PAYLOAD, int TILE = kPTile>
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
      cur.kp = (const uint32_t*)s.key_ptrs[l.batch];
      cur.pp = s.pay_ptrs ? (const uint32_t*)s.pay_ptrs[l.batch] : nullptr;
      cur.rows = batch_rows(s.v, l.batch);
      cur.row0 = batch_row0(s.v, l.batch);
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
      for (int j = 0; j < TPT / 4; ++j) kv[j] = ((const uint4*)kp)[j * BLOCK + threadIdx.x];
      if (WANT_PAYLOAD && pp) {
#pragma unroll
        for (int j = 0; j < TPT / 4; ++j) pv[j] = ((const uint4*)pp)[j * BLOCK + threadIdx.x];
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
          key[4 * j + k] = ok ? kp[idx] : 0u;
          if (WANT_PAYLOAD) pay[4 * j + k] = ok ? (pp ? pp[idx] : rowid0 + idx) : 0u;
          mask |= (ok ? 1u : 0u) << (4 * j + k);
        }
      }
    }
  } else {
    const uint64_t base = r.a + tile * (uint64_t)TILE;
    if (base >= r.lo && base + TILE <= r.hi) {  // whole tile inside the segment (uniform)
      const uint4* tv = (const uint4*)(s.tuples + base);
#pragma unroll
      for (int j = 0; j < TPT / 2; ++j) {
        const uint4 t = tv[j * BLOCK + threadIdx.x];
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
        t = *(const uint4*)(s.tuples + i0);
      } else if (i0 < r.hi) {  // last, odd tuple: never read past the end of a caller's array
        const uint2 t1 = s.tuples[i0];
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

// ------------------------------------------------------------------ hist
template <int SRC, int BLOCK, int TILE = kPTile>
__global__ __launch_bounds__(BLOCK) void k_part_hist(PassSrc s, uint32_t shift, uint32_t fbits,
                                                     uint32_t* hist) {
  constexpr int TPT = TILE / BLOCK;
  __shared__ uint32_t cl[kMaxF];
  const uint32_t F = 1u << fbits;
  for (uint32_t b = threadIdx.x; b < F; b += BLOCK) cl[b] = 0;
  __syncthreads();
  const TileRange r = block_tiles<SRC, TILE>(s);
  ColsCursor cur;
  for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
    uint32_t key[TPT], pay[TPT];
    const uint32_t mask = load_tile<SRC, BLOCK, false, TILE>(s, r, tile, cur, key, pay);
#pragma unroll
    for (int i = 0; i < TPT; ++i) {
      if (mask & (1u << i)) atomicAdd(&cl[(wang_hash(key[i]) >> shift) & (F - 1)], 1u);
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < F; b += BLOCK) hist[(uint64_t)blockIdx.x * F + b] = cl[b];
}

// ------------------------------------------------------------------ scan
// One workgroup per input segment.  hist[(seg*Bseg + blk)*F + b] (counts) is
// replaced by the exclusive prefix over blk; seg_off_out[seg*F + b] receives the
// absolute start of output bucket (seg, b) -- the segment table of the next pass.
__global__ __launch_bounds__(1024) void k_part_scan(uint32_t* hist, const uint64_t* seg_off_in,
                                                    uint64_t total_in, uint32_t n_seg, uint32_t Bseg,
                                                    uint32_t fbits, uint64_t* seg_off_out) {
  __shared__ uint64_t s_wave[16];
  const uint32_t F = 1u << fbits;
  const uint32_t seg = blockIdx.x;
  const uint32_t b = threadIdx.x;
  uint64_t tot = 0;
  if (b < F) {
    uint32_t run = 0;
    uint32_t* col = hist + (uint64_t)seg * Bseg * F + b;
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
        const uint32_t b = (wang_hash(key[i]) >> shift) & (F - 1);
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
      const uint32_t b = (wang_hash(t.x) >> shift) & (F - 1);
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
  if (r.t0 < r.t1) mask = load_tile<SRC, BLOCK, true, TILE>(s, r, r.t0, cur, key, pay);
  for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
    // (2) bucket + rank of every tuple
    uint32_t br[kWTPT];
    constexpr uint32_t kFull = (1u << kWTPT) - 1u;
    if (mask == kFull && !s.chk_mask) {  // whole tile valid (all but the edge tiles): no per-tuple predication
#pragma unroll
      for (int i = 0; i < kWTPT; ++i) {
        const uint32_t b = (wang_hash(key[i]) >> shift) & (F - 1);
        br[i] = (b << 16) | atomicAdd(&cl[b], 1u);
      }
    } else {
      const uint32_t expect = r.seg & s.chk_mask;
#pragma unroll
      for (int i = 0; i < kWTPT; ++i) {
        br[i] = 0;
        const uint32_t h = wang_hash(key[i]);
        if (((h >> s.chk_shift) & s.chk_mask) != expect) mask &= ~(1u << i);  // a filler of the claiming scatter
        if (mask & (1u << i)) {
          const uint32_t b = (h >> shift) & (F - 1);
          br[i] = (b << 16) | atomicAdd(&cl[b], 1u);
        }
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
};
constexpr uint32_t kClaimChunk = 128;

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
        const uint32_t b = (h >> shift) & (F - 1);
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
    while (((wang_hash(filler) >> shift) & (F - 1)) == b_own) ++filler;  // a key of another bucket
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

// ------------------------------------------------------------ build + probe
// counters: [0] missing keys, [1] duplicate build keys, [2] overflowing
// partitions, [3] max build rows
// A partitioned side: exact layout (cap == 0: partition p = [off[p], off[p+1])) or
// capacity layout (partition p = [p * cap, p * cap + cnt[p]), dense start dense[p]).
struct PartView {
  const uint2* t;
  const uint64_t* off;
  const uint32_t* cnt;
  const uint64_t* dense;
  uint32_t cap;
};

__global__ __launch_bounds__(kJBlock, 4) void k_join_probe(PartView RV, PartView LV, uint32_t nparts,
                                                           uint32_t* out_fk, uint32_t* out_y,
                                                           uint32_t* out_x, uint32_t pair_store_ok,
                                                           unsigned long long* counters, uint32_t* miss_bits) {
  const uint2* R = RV.t;
  const uint2* L = LV.t;
  __shared__ uint32_t s_keys[kSlots];
  __shared__ uint32_t s_vals[kSlots];
  __shared__ uint32_t s_sp[2];  // the one key equal to the empty marker: {present, value}
  uint32_t my_miss = 0, my_dup = 0;
  uint64_t my_max = 0;
  for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) {
    const uint64_t rlo = RV.cap ? (uint64_t)p * RV.cap : RV.off[p];
    const uint64_t rhi = RV.cap ? rlo + RV.cnt[p] : RV.off[p + 1];
    const uint64_t llo = LV.cap ? (uint64_t)p * LV.cap : LV.off[p];
    const uint64_t lhi = LV.cap ? llo + LV.cnt[p] : LV.off[p + 1];
    const uint64_t n_r = rhi - rlo, n_l = lhi - llo;
    // output rows are dense: tuple at sparse index i goes to row i + delta
    const uint64_t delta = (LV.cap ? LV.dense[p] : llo) - llo;
    const uint32_t pair_store = pair_store_ok && (delta & 1) == 0;
    if (n_r > my_max) my_max = n_r;
    if (n_l == 0) continue;
    if (n_r > kTableCap) {
      if (threadIdx.x == 0) atomicAdd(&counters[2], 1ull);
      continue;
    }
    for (uint32_t i = threadIdx.x; i < kSlots; i += kJBlock) s_keys[i] = kEmpty;
    if (threadIdx.x == 0) s_sp[0] = 0;
    __syncthreads();
    // build: table[pk] = payload (hash_build.c:24-31; value = right payload instead of row
    // index).  Tuples are read 16 B per lane (2 tuples) from the 16-B aligned base below the
    // partition, kJU vectors per thread in flight.
    {
      const uint64_t a = rlo & ~1ull;
      const uint32_t span = (uint32_t)(rhi - a);
      for (uint32_t base = 0; base < span; base += kJBlock * 2 * kJU) {
        uint4 v[kJU];
#pragma unroll
        for (int u = 0; u < kJU; ++u) {
          const uint32_t i0 = base + (u * kJBlock + threadIdx.x) * 2;
          v[u] = make_uint4(0, 0, 0, 0);
          if (i0 < span) v[u] = *(const uint4*)(R + a + i0);  // arrays are padded by 2 tuples
        }
#pragma unroll
        for (int u = 0; u < kJU; ++u) {
          const uint64_t i0 = a + base + (uint64_t)(u * kJBlock + threadIdx.x) * 2;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t k = h ? v[u].z : v[u].x, val = h ? v[u].w : v[u].y;
            if (i0 + h < rlo || i0 + h >= rhi) continue;
            if (k == kEmpty) {
              s_sp[1] = val;
              if (atomicExch(&s_sp[0], 1u)) ++my_dup;
            } else {
              uint32_t slot = wang_hash(k) & (kSlots - 1);
              for (uint32_t step = 0; step < kSlots; ++step) {
                const uint32_t old = atomicCAS(&s_keys[slot], kEmpty, k);
                if (old == kEmpty || old == k) {  // free cell, or equal key overwrites (hashtable.c:133)
                  s_vals[slot] = val;
                  if (old == k) ++my_dup;
                  break;
                }
                slot = (slot + 1) & (kSlots - 1);
              }
            }
          }
        }
      }
    }
    __syncthreads();
    // probe (hash_probe.c:26-36) fused with the take of the right payload
    {
      const uint64_t a = llo & ~1ull;
      const uint64_t span = lhi - a;
      for (uint64_t base = 0; base < span; base += kJBlock * 2 * kJU) {
        uint4 v[kJU];
#pragma unroll
        for (int u = 0; u < kJU; ++u) {
          const uint64_t i0 = base + (uint64_t)(u * kJBlock + threadIdx.x) * 2;
          v[u] = make_uint4(0, 0, 0, 0);
          if (i0 < span) v[u] = *(const uint4*)(L + a + i0);
        }
#pragma unroll
        for (int u = 0; u < kJU; ++u) {
          const uint64_t i0 = a + base + (uint64_t)(u * kJBlock + threadIdx.x) * 2;
          uint32_t xs[2] = {0, 0};
          bool ok[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t k = h ? v[u].z : v[u].x;
            ok[h] = i0 + h >= llo && i0 + h < lhi;
            if (!ok[h]) continue;
            bool hit = false;
            if (k == kEmpty) {
              hit = s_sp[0] != 0;
              xs[h] = s_sp[1];
            } else {
              uint32_t slot = wang_hash(k) & (kSlots - 1);
              for (uint32_t step = 0; step < kSlots; ++step) {
                const uint32_t kk = s_keys[slot];
                if (kk == k) {
                  xs[h] = s_vals[slot];
                  hit = true;
                  break;
                }
                if (kk == kEmpty) break;
                slot = (slot + 1) & (kSlots - 1);
              }
            }
            if (!hit) {
              ++my_miss;
              if (miss_bits) atomicOr(&miss_bits[(i0 + h + delta) >> 5], 1u << ((i0 + h + delta) & 31));  // drop mode: mark the row
            }
          }
          if (ok[0] && ok[1] && pair_store) {  // i0 is even: 8-byte aligned pair stores
            const uint64_t o = i0 + delta;
            *(uint2*)(out_fk + o) = make_uint2(v[u].x, v[u].z);
            *(uint2*)(out_y + o) = make_uint2(v[u].y, v[u].w);
            *(uint2*)(out_x + o) = make_uint2(xs[0], xs[1]);
          } else {
            const uint64_t o = i0 + delta;
            if (ok[0]) {
              out_fk[o] = v[u].x;
              out_y[o] = v[u].y;
              out_x[o] = xs[0];
            }
            if (ok[1]) {
              out_fk[o + 1] = v[u].z;
              out_y[o + 1] = v[u].w;
              out_x[o + 1] = xs[1];
            }
          }
        }
      }
    }
    __syncthreads();
  }
  const uint64_t wm = wave_reduce_add((uint64_t)my_miss);
  const uint64_t wd = wave_reduce_add((uint64_t)my_dup);
  if (lane_id() == 0) {
    if (wm) atomicAdd(&counters[0], (unsigned long long)wm);
    if (wd) atomicAdd(&counters[1], (unsigned long long)wd);
  }
  if (threadIdx.x == 0) atomicMax(&counters[3], (unsigned long long)my_max);
}

// ------------------------------------------- build + probe, direct-address table
// Every step of the Wang hash (partition.c:20-28) is invertible modulo 2^32, so DISTINCT
// keys have DISTINCT hashes.  All keys of one final partition share the top
// (skip + radix) hash bits; when at most kDirectBits bits remain below them, the remaining
// bits index an LDS table without any collision -- for ANY key set, not only dense ones:
// no probing, no compare-and-swap; an equal key simply lands on its own slot again (the
// reference's overwrite, hashtable.c:133).  16384 slots x (key, value) = 128 KiB, one
// workgroup of 1024 per CU; partitions hold up to 16 Ki build rows, which lets the radix
// passes stop at 512 buckets each (two scatter workgroups per CU).
constexpr int kDirectBits = 14;
constexpr int kDSlots = 1 << kDirectBits;
constexpr int kDBlock = 512;

// The table does not even store keys: inside a partition the slot index IS the key (same
// top bits + same low bits = same hash = same key), so a slot holds the payload and one
// occupancy bit.  64 KiB values + 2 KiB bitmap: two workgroups of 512 per CU, and only the
// bitmap needs clearing between partitions.
__global__ __launch_bounds__(kDBlock, 4) void k_join_probe_direct(PartView RV, PartView LV, uint32_t nparts,
                                                               uint32_t table_bits, uint32_t* out_fk,
                                                               uint32_t* out_y, uint32_t* out_x,
                                                               uint32_t pair_store_ok, unsigned long long* counters,
                                                               uint32_t* miss_bits) {
  __shared__ uint32_t s_vals[kDSlots];
  __shared__ uint32_t s_used[kDSlots / 32];
  const uint2* R = RV.t;
  const uint2* L = LV.t;
  const uint32_t slots = 1u << table_bits, smask = slots - 1;
  uint32_t my_miss = 0, my_dup = 0;
  uint64_t my_max = 0;
  for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) {
    const uint64_t rlo = RV.cap ? (uint64_t)p * RV.cap : RV.off[p];
    const uint64_t rhi = RV.cap ? rlo + RV.cnt[p] : RV.off[p + 1];
    const uint64_t llo = LV.cap ? (uint64_t)p * LV.cap : LV.off[p];
    const uint64_t lhi = LV.cap ? llo + LV.cnt[p] : LV.off[p + 1];
    const uint64_t n_r = rhi - rlo, n_l = lhi - llo;
    const uint64_t delta = (LV.cap ? LV.dense[p] : llo) - llo;
    const uint32_t pair_store = pair_store_ok && (delta & 1) == 0;
    if (n_r > my_max) my_max = n_r;
    if (n_l == 0) continue;
    for (uint32_t i = threadIdx.x; i < (slots + 31) / 32; i += kDBlock) s_used[i] = 0;
    __syncthreads();
    {  // build
      const uint64_t a = rlo & ~1ull;
      const uint64_t span = rhi - a;
      for (uint64_t base = 0; base < span; base += kDBlock * 2 * kJU) {
        uint4 v[kJU];
#pragma unroll
        for (int u = 0; u < kJU; ++u) {
          const uint64_t i0 = base + (uint64_t)(u * kDBlock + threadIdx.x) * 2;
          v[u] = make_uint4(0, 0, 0, 0);
          if (i0 + 1 < span) {
            v[u] = *(const uint4*)(R + a + i0);
          } else if (i0 < span) {
            const uint2 t1 = R[a + i0];
            v[u].x = t1.x;
            v[u].y = t1.y;
          }
        }
#pragma unroll
        for (int u = 0; u < kJU; ++u) {
          const uint64_t i0 = a + base + (uint64_t)(u * kDBlock + threadIdx.x) * 2;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t k = h ? v[u].z : v[u].x, val = h ? v[u].w : v[u].y;
            if (i0 + h < rlo || i0 + h >= rhi) continue;
            const uint32_t slot = wang_hash(k) & smask;
            const uint32_t bit = 1u << (slot & 31);
            if (atomicOr(&s_used[slot >> 5], bit) & bit) ++my_dup;  // equal key overwrites (hashtable.c:133)
            s_vals[slot] = val;
          }
        }
      }
    }
    __syncthreads();
    {  // probe + take of the right payload
      const uint64_t a = llo & ~1ull;
      const uint64_t span = lhi - a;
      for (uint64_t base = 0; base < span; base += kDBlock * 2 * kJU) {
        uint4 v[kJU];
#pragma unroll
        for (int u = 0; u < kJU; ++u) {
          const uint64_t i0 = base + (uint64_t)(u * kDBlock + threadIdx.x) * 2;
          v[u] = make_uint4(0, 0, 0, 0);
          if (i0 + 1 < span) {
            v[u] = *(const uint4*)(L + a + i0);
          } else if (i0 < span) {
            const uint2 t1 = L[a + i0];
            v[u].x = t1.x;
            v[u].y = t1.y;
          }
        }
#pragma unroll
        for (int u = 0; u < kJU; ++u) {
          const uint64_t i0 = a + base + (uint64_t)(u * kDBlock + threadIdx.x) * 2;
          uint32_t xs[2] = {0, 0};
          bool ok[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t k = h ? v[u].z : v[u].x;
            ok[h] = i0 + h >= llo && i0 + h < lhi;
            if (!ok[h]) continue;
            const uint32_t slot = wang_hash(k) & smask;
            const bool hit = (s_used[slot >> 5] >> (slot & 31)) & 1u;
            xs[h] = s_vals[slot];
            if (!hit) {
              ++my_miss;
              if (miss_bits) atomicOr(&miss_bits[(i0 + h + delta) >> 5], 1u << ((i0 + h + delta) & 31));  // drop mode: mark the row
            }
          }
          const uint64_t o = i0 + delta;
          if (ok[0] && ok[1] && pair_store) {
            *(uint2*)(out_fk + o) = make_uint2(v[u].x, v[u].z);
            *(uint2*)(out_y + o) = make_uint2(v[u].y, v[u].w);
            *(uint2*)(out_x + o) = make_uint2(xs[0], xs[1]);
          } else {
            if (ok[0]) {
              out_fk[o] = v[u].x;
              out_y[o] = v[u].y;
              out_x[o] = xs[0];
            }
            if (ok[1]) {
              out_fk[o + 1] = v[u].z;
              out_y[o + 1] = v[u].w;
              out_x[o + 1] = xs[1];
            }
          }
        }
      }
    }
    __syncthreads();
  }
  const uint64_t wm = wave_reduce_add((uint64_t)my_miss);
  const uint64_t wd = wave_reduce_add((uint64_t)my_dup);
  if (lane_id() == 0) {
    if (wm) atomicAdd(&counters[0], (unsigned long long)wm);
    if (wd) atomicAdd(&counters[1], (unsigned long long)wd);
  }
  if (threadIdx.x == 0) atomicMax(&counters[3], (unsigned long long)my_max);
}

__global__ __launch_bounds__(256) void k_unzip(const uint2* t, uint64_t n, uint32_t* keys, uint32_t* pay) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint2 v = t[i];
    keys[i] = v.x;
    pay[i] = v.y;
  }
}

// ------------------------------------------------------------------ host
struct PassPlan {
  int passes = 0;
  uint32_t bits[kMaxPasses] = {0, 0, 0, 0};
  uint32_t total_bits = 0;
};

// ---- inner-join mode: rows whose foreign key found no primary key are dropped
constexpr uint32_t kCompactSeg = 1u << 16;  // rows per compaction segment (2048 bitmap words)

// kept[s] = rows of segment s that are not marked in the miss bitmap
__global__ __launch_bounds__(256) void k_compact_count(const uint32_t* miss_bits, uint64_t n_rows, uint32_t* kept) {
  __shared__ uint32_t s_wave[4];
  const uint64_t seg = blockIdx.x;
  const uint64_t lo = seg * kCompactSeg, hi = min(n_rows, lo + kCompactSeg);
  uint32_t marked = 0;
  for (uint64_t w = lo / 32 + threadIdx.x; w * 32 < hi; w += 256) {
    uint32_t bits = miss_bits[w];
    if (hi - w * 32 < 32) bits &= (1u << (hi - w * 32)) - 1u;
    marked += __popc(bits);
  }
  marked = (uint32_t)wave_reduce_add((uint64_t)marked);
  if (lane_id() == 0) s_wave[threadIdx.x / kWave] = marked;
  __syncthreads();
  if (threadIdx.x == 0) kept[seg] = (uint32_t)(hi - lo) - (s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3]);
}

// dst[base[s] + rank] = src[row] for the unmarked rows of segment s, order preserved
__global__ __launch_bounds__(256) void k_compact_rows(const uint32_t* src, const uint32_t* miss_bits, const uint64_t* base,
                                                      uint64_t n_rows, uint32_t* dst) {
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

static int ceil_log2_u64(uint64_t x) {
  int l = 0;
  while ((1ull << l) < x) ++l;
  return l;
}

static int make_plan(olap_ctx* ctx, uint32_t total_bits, PassPlan* plan) {
  int maxb = ctx->max_bits_per_pass > 0 ? ctx->max_bits_per_pass : kMaxFBits;
  if (maxb > kMaxFBits) maxb = kMaxFBits;
  if (total_bits < 1) total_bits = 1;
  const int passes = ((int)total_bits + maxb - 1) / maxb;
  if (passes > kMaxPasses)
    return ctx->fail(OLAP_ERR_INVALID, "%u radix bits need %d passes (max %d)", total_bits, passes, kMaxPasses);
  plan->passes = passes;
  plan->total_bits = total_bits;
  const uint32_t base = total_bits / passes, extra = total_bits % passes;
  for (int i = 0; i < passes; ++i) plan->bits[i] = base + ((uint32_t)i < extra ? 1u : 0u);
  return OLAP_OK;
}

struct SideBuffers {
  uint2* final_tuples = nullptr;    // rows + 2
  uint64_t* final_off = nullptr;    // [P + 1]
};

static uint64_t seg_blocks(const olap_ctx* ctx, uint64_t n_seg) {
  uint64_t want = (uint64_t)ctx->compute_units * 4;
  uint64_t b = (want + n_seg - 1) / n_seg;
  if (b < 1) b = 1;
  if (b > 512) b = 512;  // two workgroups per CU when a single segment feeds the whole chip
  return b;
}

// Tuples that arrive already grouped by their leading hash bits (below the skipped ones):
// segment j fixes `bits` = log2(n_seg) bits and is the union of `chunks` chunks
// [lo[j*chunks + c], hi[j*chunks + c]) of the tuple array -- one chunk per sender after
// the fused owner + sub-bucket exchange.  Host arrays.
struct Presplit {
  uint32_t n_seg = 0;
  uint32_t chunks = 0;
  uint32_t bits = 0;
  const uint64_t* lo = nullptr;
  const uint64_t* hi = nullptr;
};

// workspace bytes one side needs besides its final tuple buffer and the shared temps
static uint64_t side_scratch_bytes(const olap_ctx* ctx, uint64_t rows, uint64_t n_batches, const PassPlan& plan,
                                   const Presplit* pre = nullptr) {
  uint64_t bytes = 0;
  uint64_t n_seg = pre ? pre->n_seg : 1;
  // upper bound of pass-0 tiles: the smaller tile geometry, every batch rounded up on its own
  const uint64_t tiles = rows / 4096 + n_batches + 1;
  if (pre) bytes += align_up((uint64_t)pre->n_seg * pre->chunks * 16, 256) + align_up((n_seg + 1) * 8, 256);
  for (int i = 0; i < plan.passes; ++i) {
    const uint64_t F = 1ull << plan.bits[i];
    uint64_t blocks;
    if (i == 0 && pre) {
      blocks = n_seg * pre->chunks;
    } else if (i == 0) {
      blocks = (uint64_t)ctx->compute_units * 2;
      if (blocks > tiles) blocks = tiles;
    } else {
      blocks = n_seg * seg_blocks(ctx, n_seg);
    }
    bytes += align_up(blocks * F * 4, 256);          // hist matrix
    bytes += align_up((n_seg * F + 1) * 8, 256);     // segment table
    n_seg *= F;
  }
  bytes += align_up(n_seg * 4, 256) + align_up(n_seg * 8, 256);  // capacity-layout tables
  bytes += align_up((1ull << plan.bits[0]) * 4, 256) + align_up((1ull << plan.bits[0]) * 8, 256);  // claim cursors, segment totals
  return bytes + 4096;
}

// Runs all passes of one side.  `cs` views the key/payload columns with unit_rows = kPTile.
struct SideOut {
  uint64_t* off = nullptr;   // exact layout: [P + 1]
  uint32_t cap = 0;          // capacity layout: tuples per region (0 = exact)
  uint32_t* cnt = nullptr;   // capacity layout: [P]
  uint64_t* dense = nullptr; // capacity layout: [P]
};

static uint32_t region_cap(uint64_t rows, uint64_t P) {
  const uint64_t avg = (rows + P - 1) / P;
  return (uint32_t)align_up(avg + avg / 8 + 64, kLineAlign);
}

// The last pass may skip its histogram (capacity layout) when every workgroup owns whole
// input segments and there are enough of them to fill the chip.
static bool capacity_eligible(const olap_ctx* ctx, uint64_t rows, const PassPlan& plan, uint32_t pre_bits = 0) {
  if (!ctx->capacity_mode || ctx->scatter_variant != 0 || plan.passes < 2) return false;
  const uint64_t P = 1ull << (plan.total_bits + pre_bits);
  const uint64_t n_seg_last = P >> plan.bits[plan.passes - 1];
  return n_seg_last >= (uint64_t)ctx->compute_units && rows / P >= 512;
}

// Region size of the claiming scatter: the average bucket, 8 sigma of slack (uniform hash),
// and what the workgroups may leave unused in their last allotments.  0: does not fit 32 bits.
static uint32_t claim_cap(const olap_ctx* ctx, uint64_t rows, uint64_t F) {
  const uint64_t kb = (uint64_t)ctx->compute_units * 2 > 256 ? (uint64_t)ctx->compute_units * 2 : 256;  // workgroups of pass 0
  const uint64_t avg = (rows + F - 1) / F;
  uint64_t sd = 1;
  while (sd * sd < avg) ++sd;
  const uint64_t cap = align_up(avg + avg / 64 + 8 * sd + (kb + 2) * kClaimChunk, kClaimChunk);
  return cap >= (1ull << 32) ? 0u : (uint32_t)cap;
}

// Pass 0 of a two-pass plan may run as a claiming scatter (no histogram) when the last pass
// uses the capacity layout; automatic when the padding it costs stays below 1/16 of the rows.
static bool claim_eligible(const olap_ctx* ctx, uint64_t rows, const PassPlan& plan, bool capm) {
  if (!ctx->claim_mode || !capm || plan.passes != 2 || ctx->scatter_variant != 0 || ctx->line16) return false;
  const uint64_t F = 1ull << plan.bits[0];
  if (!claim_cap(ctx, rows, F)) return false;
  if (ctx->claim_mode >= 2) return true;
  const uint64_t kb = (uint64_t)ctx->compute_units * 2 > 256 ? (uint64_t)ctx->compute_units * 2 : 256;
  return kb * kClaimChunk * F <= rows / 16;
}

constexpr int kW2Block = 512;   // geometry 2 of the write-combining scatter: two workgroups per CU
constexpr int kW2Tile = 4096;
constexpr uint32_t kWc2Lds = WcLds<8, kW2Block, kW2Tile>::kBytes;

static bool pass_uses_geometry2(const olap_ctx* ctx, uint32_t fbits) {
  return ctx->scatter_variant == 0 && ctx->wc_geometry == 2 && !ctx->line16 && fbits <= 9;
}

// Tile size (rows per unit of the column view) pass 0 will use for `plan`.
static uint32_t pass0_tile(const olap_ctx* ctx, const PassPlan& plan) {
  return pass_uses_geometry2(ctx, plan.bits[0]) ? kW2Tile : kPTile;
}

// Input of pass 0: the batched columns `cs`, or (tuple_in != nullptr) an array of tuple_n
// (key, payload) tuples already in HBM -- what a rank receives from the all-to-all.
static int partition_side(olap_ctx* ctx, const ColumnSet& cs, bool has_payload, const uint2* tuple_in,
                          uint64_t tuple_n, const PassPlan& plan, uint32_t skip_bits, uint2* tmp0, uint2* tmp1,
                          uint2* final_tuples, bool capm, unsigned long long* d_overflow, SideOut* so,
                          const Presplit* pre = nullptr, bool claim = false) {
  const uint64_t rows = tuple_in ? tuple_n : cs.view.total_rows;
  PassSrc src{};
  uint64_t n_seg = 1;
  const uint64_t* seg_in = nullptr;
  const uint2* in_tuples = nullptr;
  const uint64_t* pre_ranges = nullptr;
  if (tuple_in && pre) {
    // pre-split input: upload the chunk table and the dense base of every segment
    const uint64_t nchunk = (uint64_t)pre->n_seg * pre->chunks;
    uint64_t* d_rng = (uint64_t*)ws_alloc(ctx, nchunk * 16);
    uint64_t* d_base = (uint64_t*)ws_alloc(ctx, ((uint64_t)pre->n_seg + 1) * 8);
    uint64_t* h_rng = (uint64_t*)pinned_alloc(ctx, nchunk * 16);
    uint64_t* h_base = (uint64_t*)pinned_alloc(ctx, ((uint64_t)pre->n_seg + 1) * 8);
    if (!d_rng || !d_base || !h_rng || !h_base) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (chunk table)");
    uint64_t run = 0;
    for (uint32_t j = 0; j < pre->n_seg; ++j) {
      h_base[j] = run;
      for (uint32_t c = 0; c < pre->chunks; ++c) {
        const uint64_t k = (uint64_t)j * pre->chunks + c;
        h_rng[2 * k] = pre->lo[k];
        h_rng[2 * k + 1] = pre->hi[k];
        run += pre->hi[k] - pre->lo[k];
      }
    }
    h_base[pre->n_seg] = run;
    OLAP_HIP(ctx, hipMemcpyAsync(d_rng, h_rng, nchunk * 16, hipMemcpyHostToDevice, ctx->stream));
    OLAP_HIP(ctx, hipMemcpyAsync(d_base, h_base, ((uint64_t)pre->n_seg + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    pre_ranges = d_rng;
    seg_in = d_base;
    n_seg = pre->n_seg;
    in_tuples = tuple_in;
  } else if (tuple_in) {
    uint64_t* seg0 = (uint64_t*)ws_alloc(ctx, 16);
    uint64_t* h_seg0 = (uint64_t*)pinned_alloc(ctx, 16);
    if (!seg0 || !h_seg0) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (segment table)");
    h_seg0[0] = 0;
    h_seg0[1] = tuple_n;
    OLAP_HIP(ctx, hipMemcpyAsync(seg0, h_seg0, 16, hipMemcpyHostToDevice, ctx->stream));
    seg_in = seg0;
    in_tuples = tuple_in;
  } else {
    src.v = cs.view;
    src.key_ptrs = cs.dev_ptr_tables[0];
    src.pay_ptrs = has_payload ? cs.dev_ptr_tables[1] : nullptr;
  }
  uint32_t consumed = skip_bits + (pre ? pre->bits : 0u);
  OLAP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_scatter<SRC_COLS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kScatterLds));
  OLAP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_scatter<SRC_TUPLES>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kScatterLds));
#define OLAP_WC_ATTR(SRCV, CAPV, LINEV)                                                                      \
  OLAP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_scatter_wc<SRCV, CAPV, LINEV>),     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, WcLds<LINEV>::kBytes))
  OLAP_WC_ATTR(SRC_COLS, false, 8);
  OLAP_WC_ATTR(SRC_COLS, false, 16);
  OLAP_WC_ATTR(SRC_TUPLES, false, 8);
  OLAP_WC_ATTR(SRC_TUPLES, false, 16);
  OLAP_WC_ATTR(SRC_TUPLES, true, 8);
  OLAP_WC_ATTR(SRC_TUPLES, true, 16);
#undef OLAP_WC_ATTR
#define OLAP_WC2_ATTR(SRCV, CAPV)                                                                                      \
  OLAP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_scatter_wc<SRCV, CAPV, 8, kW2Block, kW2Tile>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kWc2Lds))
  OLAP_WC2_ATTR(SRC_COLS, false);
  OLAP_WC2_ATTR(SRC_TUPLES, false);
  OLAP_WC2_ATTR(SRC_TUPLES, true);
#undef OLAP_WC2_ATTR
#define OLAP_CLAIM_ATTR(SRCV, BLOCKV, TILEV, LDSV)                                                                 \
  OLAP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_scatter_claim<SRCV, BLOCKV, TILEV>),     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDSV))
  OLAP_CLAIM_ATTR(SRC_COLS, kW2Block, kW2Tile, kWc2Lds);
  OLAP_CLAIM_ATTR(SRC_TUPLES, kW2Block, kW2Tile, kWc2Lds);
  OLAP_CLAIM_ATTR(SRC_COLS, kWBlock, kPTile, WcLds<8>::kBytes);
  OLAP_CLAIM_ATTR(SRC_TUPLES, kWBlock, kPTile, WcLds<8>::kBytes);
#undef OLAP_CLAIM_ATTR
  bool claimed_input = false;  // the current input was written by the claiming scatter
  const bool wc = ctx->scatter_variant == 0;  // write-combining scatter
  for (int i = 0; i < plan.passes; ++i) {
    const uint32_t fbits = plan.bits[i];
    const uint64_t F = 1ull << fbits;
    consumed += fbits;
    const uint32_t shift = 32 - consumed;
    uint2* out = (i == plan.passes - 1) ? final_tuples : ((i & 1) ? tmp1 : tmp0);
    uint64_t blocks, bseg;
    const bool cols = i == 0 && !tuple_in;
    const bool g2 = wc && pass_uses_geometry2(ctx, fbits);  // 512 threads, 4096-tuple tiles, 2 workgroups per CU
    if (cols) {
      const uint64_t tiles = cs.view.total_units;
      uint64_t nb = (uint64_t)ctx->compute_units * ((wc && !g2) ? 1 : 2);
      if (nb > tiles) nb = tiles;
      if (nb == 0) nb = 1;
      src.tiles_per_block = (tiles + nb - 1) / nb;
      if (src.tiles_per_block == 0) src.tiles_per_block = 1;
      blocks = (tiles + src.tiles_per_block - 1) / src.tiles_per_block;
      if (blocks == 0) blocks = 1;
      bseg = blocks;
    } else {
      const bool ranged = pre_ranges && i == 0;  // one workgroup per chunk of the pre-split input
      bseg = ranged ? pre->chunks : (capm && i == plan.passes - 1) ? 1 : seg_blocks(ctx, n_seg);
      blocks = n_seg * bseg;
      src.tuples = in_tuples;
      src.seg_off = seg_in;
      src.n_seg = (uint32_t)n_seg;
      src.blocks_per_seg = (uint32_t)bseg;
      src.blk_range = ranged ? pre_ranges : nullptr;
    }
    if (blocks > 0x7FFFFFFFull) return ctx->fail(OLAP_ERR_INVALID, "partition grid too large");
    if (claim && i == 0 && plan.passes == 2 && wc && !pre) {  // claiming scatter: no histogram
      const uint32_t ccap = claim_cap(ctx, rows, F);
      uint32_t* cursor = (uint32_t*)ws_alloc(ctx, F * 4);
      if (!cursor || !ccap) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (claim cursors)");
      OLAP_HIP(ctx, hipMemsetAsync(cursor, 0, F * 4, ctx->stream));
      {
        PhaseScope ps(ctx, PH_PART_SCATTER);
        const ClaimOut co{ccap, cursor, d_overflow};
        if (g2 && cols)
          hipLaunchKernelGGL((k_part_scatter_claim<SRC_COLS, kW2Block, kW2Tile>), dim3((unsigned)blocks), dim3(kW2Block), kWc2Lds,
                             ctx->stream, src, shift, fbits, out, co);
        else if (g2)
          hipLaunchKernelGGL((k_part_scatter_claim<SRC_TUPLES, kW2Block, kW2Tile>), dim3((unsigned)blocks), dim3(kW2Block), kWc2Lds,
                             ctx->stream, src, shift, fbits, out, co);
        else if (cols)
          hipLaunchKernelGGL((k_part_scatter_claim<SRC_COLS, kWBlock, kPTile>), dim3((unsigned)blocks), dim3(kWBlock),
                             WcLds<8>::kBytes, ctx->stream, src, shift, fbits, out, co);
        else
          hipLaunchKernelGGL((k_part_scatter_claim<SRC_TUPLES, kWBlock, kPTile>), dim3((unsigned)blocks), dim3(kWBlock),
                             WcLds<8>::kBytes, ctx->stream, src, shift, fbits, out, co);
        OLAP_HIP(ctx, hipGetLastError());
      }
      src.seg_cnt = cursor;  // the next pass reads regions of `ccap` tuples and skips the fillers
      src.seg_cap = ccap;
      src.chk_shift = shift;
      src.chk_mask = (uint32_t)F - 1;
      claimed_input = true;
      in_tuples = out;
      seg_in = nullptr;
      n_seg *= F;
      continue;
    }
    if (capm && i == plan.passes - 1 && !src.blk_range) {  // capacity layout: scatter only
      so->cap = region_cap(rows, n_seg * F);
      so->cnt = (uint32_t*)ws_alloc(ctx, n_seg * F * 4);
      so->dense = (uint64_t*)ws_alloc(ctx, n_seg * F * 8);
      if (!so->cnt || !so->dense) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (capacity tables)");
      PhaseScope ps(ctx, PH_PART_SCATTER);
      uint64_t* seg_total = nullptr;
      if (claimed_input) {
        seg_total = (uint64_t*)ws_alloc(ctx, n_seg * 8);
        if (!seg_total) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (segment totals)");
      }
      CapOut co{so->cap, so->cnt, so->dense, d_overflow, seg_total};
      if (g2)
        hipLaunchKernelGGL((k_part_scatter_wc<SRC_TUPLES, true, 8, kW2Block, kW2Tile>), dim3((unsigned)blocks), dim3(kW2Block),
                           kWc2Lds, ctx->stream, src, shift, fbits, (const uint32_t*)nullptr,
                           (const uint64_t*)nullptr, out, co);
      else if (fbits <= 9 && ctx->line16)
        hipLaunchKernelGGL((k_part_scatter_wc<SRC_TUPLES, true, 16>), dim3((unsigned)blocks), dim3(kWBlock),
                           WcLds<16>::kBytes, ctx->stream, src, shift, fbits, (const uint32_t*)nullptr,
                           (const uint64_t*)nullptr, out, co);
      else
        hipLaunchKernelGGL((k_part_scatter_wc<SRC_TUPLES, true, 8>), dim3((unsigned)blocks), dim3(kWBlock),
                           WcLds<8>::kBytes, ctx->stream, src, shift, fbits, (const uint32_t*)nullptr,
                           (const uint64_t*)nullptr, out, co);
      OLAP_HIP(ctx, hipGetLastError());
      if (claimed_input) {  // the segments' dense starts were unknown to the scatter: number the partitions now
        hipLaunchKernelGGL(k_dense_fix, dim3((unsigned)n_seg), dim3(1024), 0, ctx->stream, (const uint64_t*)seg_total, (uint32_t)F,
                           so->dense);
        OLAP_HIP(ctx, hipGetLastError());
      }
      return OLAP_OK;
    }
    if (claimed_input) return ctx->fail(OLAP_ERR_INVALID, "internal: claimed input needs a capacity-layout consumer");
    uint32_t* hist = (uint32_t*)ws_alloc(ctx, blocks * F * 4);
    uint64_t* seg_out = (uint64_t*)ws_alloc(ctx, (n_seg * F + 1) * 8);
    if (!hist || !seg_out) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (partition pass %d)", i);
    {
      PhaseScope ps(ctx, PH_PART_HIST);
      if (g2 && cols)
        hipLaunchKernelGGL((k_part_hist<SRC_COLS, kW2Block, kW2Tile>), dim3((unsigned)blocks), dim3(kW2Block), 0, ctx->stream, src, shift, fbits, hist);
      else if (g2)
        hipLaunchKernelGGL((k_part_hist<SRC_TUPLES, kW2Block, kW2Tile>), dim3((unsigned)blocks), dim3(kW2Block), 0, ctx->stream, src, shift, fbits, hist);
      else if (cols)
        hipLaunchKernelGGL((k_part_hist<SRC_COLS, kWBlock>), dim3((unsigned)blocks), dim3(kWBlock), 0, ctx->stream, src, shift, fbits, hist);
      else
        hipLaunchKernelGGL((k_part_hist<SRC_TUPLES, kWBlock>), dim3((unsigned)blocks), dim3(kWBlock), 0, ctx->stream, src, shift, fbits, hist);
      OLAP_HIP(ctx, hipGetLastError());
    }
    {
      PhaseScope ps(ctx, PH_PART_SCAN);
      hipLaunchKernelGGL(k_part_scan, dim3((unsigned)n_seg), dim3(1024), 0, ctx->stream, hist, seg_in, rows,
                         (uint32_t)n_seg, (uint32_t)bseg, fbits, seg_out);
      OLAP_HIP(ctx, hipGetLastError());
    }
    {
      PhaseScope ps(ctx, PH_PART_SCATTER);
      const bool l16 = fbits <= 9 && ctx->line16;
      if (g2 && cols)
        hipLaunchKernelGGL((k_part_scatter_wc<SRC_COLS, false, 8, kW2Block, kW2Tile>), dim3((unsigned)blocks), dim3(kW2Block),
                           kWc2Lds, ctx->stream, src, shift, fbits, (const uint32_t*)hist,
                           (const uint64_t*)seg_out, out, CapOut{});
      else if (g2)
        hipLaunchKernelGGL((k_part_scatter_wc<SRC_TUPLES, false, 8, kW2Block, kW2Tile>), dim3((unsigned)blocks), dim3(kW2Block),
                           kWc2Lds, ctx->stream, src, shift, fbits, (const uint32_t*)hist,
                           (const uint64_t*)seg_out, out, CapOut{});
      else if (wc && cols && l16)
        hipLaunchKernelGGL((k_part_scatter_wc<SRC_COLS, false, 16>), dim3((unsigned)blocks), dim3(kWBlock), WcLds<16>::kBytes,
                           ctx->stream, src, shift, fbits, (const uint32_t*)hist, (const uint64_t*)seg_out, out, CapOut{});
      else if (wc && cols)
        hipLaunchKernelGGL((k_part_scatter_wc<SRC_COLS, false, 8>), dim3((unsigned)blocks), dim3(kWBlock), WcLds<8>::kBytes,
                           ctx->stream, src, shift, fbits, (const uint32_t*)hist, (const uint64_t*)seg_out, out, CapOut{});
      else if (wc && l16)
        hipLaunchKernelGGL((k_part_scatter_wc<SRC_TUPLES, false, 16>), dim3((unsigned)blocks), dim3(kWBlock), WcLds<16>::kBytes,
                           ctx->stream, src, shift, fbits, (const uint32_t*)hist, (const uint64_t*)seg_out, out, CapOut{});
      else if (wc)
        hipLaunchKernelGGL((k_part_scatter_wc<SRC_TUPLES, false, 8>), dim3((unsigned)blocks), dim3(kWBlock), WcLds<8>::kBytes,
                           ctx->stream, src, shift, fbits, (const uint32_t*)hist, (const uint64_t*)seg_out, out, CapOut{});
      else if (cols)
        hipLaunchKernelGGL((k_part_scatter<SRC_COLS>), dim3((unsigned)blocks), dim3(kPBlock), kScatterLds, ctx->stream,
                           src, shift, fbits, (const uint32_t*)hist, (const uint64_t*)seg_out, out);
      else
        hipLaunchKernelGGL((k_part_scatter<SRC_TUPLES>), dim3((unsigned)blocks), dim3(kPBlock), kScatterLds, ctx->stream,
                           src, shift, fbits, (const uint32_t*)hist, (const uint64_t*)seg_out, out);
      OLAP_HIP(ctx, hipGetLastError());
    }
    in_tuples = out;
    seg_in = seg_out;
    n_seg *= F;
  }
  so->off = const_cast<uint64_t*>(seg_in);
  return OLAP_OK;
}

static int check_side(olap_ctx* ctx, const void* keys, const uint64_t* rows, uint64_t nb, uint64_t* total) {
  if (nb && (!keys || !rows)) return ctx->fail(OLAP_ERR_INVALID, "null batch table");
  uint64_t t = 0;
  for (uint64_t b = 0; b < nb; ++b) t += rows[b];
  if (t > (1ull << 32)) return ctx->fail(OLAP_ERR_INVALID, "more than 2^32 rows on one side");
  *total = t;
  return OLAP_OK;
}

// Join inputs: batched columns per side, or (device only) pre-zipped tuple arrays.
struct JoinIn {
  const uint32_t* const* lfk = nullptr;
  const uint32_t* const* ly = nullptr;
  const uint64_t* lrows = nullptr;
  uint64_t lnb = 0;
  const uint32_t* const* rpk = nullptr;
  const uint32_t* const* rx = nullptr;
  const uint64_t* rrows = nullptr;
  uint64_t rnb = 0;
  int in_mem = OLAP_MEM_DEVICE;
  const uint2* l_tuples = nullptr;
  const uint2* r_tuples = nullptr;
  const Presplit* l_pre = nullptr;  // both or neither; same n_seg
  const Presplit* r_pre = nullptr;
};

static int join_once(olap_ctx* ctx, const JoinIn& in, uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x, int out_mem,
                     uint64_t NL, uint64_t NR, uint32_t total_bits, bool allow_capacity, olap_join_stats* st,
                     unsigned long long counters[5], uint64_t* rows_out) {
  const uint32_t* const* lfk = in.lfk;
  const uint32_t* const* ly = in.ly;
  const uint32_t* const* rpk = in.rpk;
  const uint32_t* const* rx = in.rx;
  const uint64_t* lrows = in.lrows;
  const uint64_t* rrows = in.rrows;
  const uint64_t lnb = in.l_tuples ? 0 : in.lnb, rnb = in.r_tuples ? 0 : in.rnb;
  const int in_mem = in.in_mem;
  PassPlan plan;
  const uint32_t pre_bits = in.l_pre ? in.l_pre->bits : 0u;  // already fixed by the pre-split
  if (total_bits <= pre_bits) total_bits = pre_bits + 1;
  OLAP_TRY(make_plan(ctx, total_bits - pre_bits, &plan));
  if (pre_bits + plan.total_bits + (uint32_t)ctx->hash_skip_bits > 32)
    return ctx->fail(OLAP_ERR_INVALID, "radix bits (%u) + skipped hash bits (%d) exceed 32", pre_bits + plan.total_bits,
                     ctx->hash_skip_bits);
  const uint32_t P = 1u << (pre_bits + plan.total_bits);
  // direct-address LDS table when few enough hash bits remain below the partition bits
  const uint32_t table_bits = 32u - (uint32_t)ctx->hash_skip_bits - pre_bits - plan.total_bits;
  const bool direct = ctx->direct_table != 0 && table_bits <= (uint32_t)kDirectBits;
  const uint32_t eb[2] = {4, 4};
  const void* const* lcols[2] = {(const void* const*)lfk, (const void* const*)ly};
  const void* const* rcols[2] = {(const void* const*)rpk, (const void* const*)rx};
  const uint64_t maxN = NL > NR ? NL : NR;
  const bool lcap = allow_capacity && capacity_eligible(ctx, NL, plan, pre_bits);
  const bool rcap = allow_capacity && capacity_eligible(ctx, NR, plan, pre_bits);
  const uint64_t lt_tuples = lcap ? (uint64_t)P * region_cap(NL, P) : NL;
  const uint64_t rt_tuples = rcap ? (uint64_t)P * region_cap(NR, P) : NR;
  const bool lclaim = !in.l_pre && claim_eligible(ctx, NL, plan, lcap);
  const bool rclaim = !in.r_pre && claim_eligible(ctx, NR, plan, rcap);
  uint64_t tmp_tuples = maxN;  // pass-0 output: dense, or the claiming scatter's padded regions
  if (lclaim) tmp_tuples = std::max<uint64_t>(tmp_tuples, ((uint64_t)claim_cap(ctx, NL, 1ull << plan.bits[0])) << plan.bits[0]);
  if (rclaim) tmp_tuples = std::max<uint64_t>(tmp_tuples, ((uint64_t)claim_cap(ctx, NR, 1ull << plan.bits[0])) << plan.bits[0]);
  const bool l_pay = ly != nullptr, r_pay = rx != nullptr;  // null payload: the tuple carries its global row index
  uint64_t need = column_set_bytes(lrows, lnb, l_pay ? 2 : 1, eb, in_mem) + column_set_bytes(rrows, rnb, r_pay ? 2 : 1, eb, in_mem) +
                  side_scratch_bytes(ctx, NL, lnb, plan, in.l_pre) + side_scratch_bytes(ctx, NR, rnb, plan, in.r_pre) +
                  align_up((lt_tuples + 2) * 8, 256) + align_up((rt_tuples + 2) * 8, 256) + 8192;
  if (plan.passes >= 2) need += align_up((tmp_tuples + 2) * 8, 256);
  if (plan.passes >= 3) need += align_up((maxN + 2) * 8, 256);
  if (out_mem == OLAP_MEM_HOST) need += 3 * align_up(NL * 4 + 4, 256);
  const bool drop = ctx->join_drop_unmatched != 0;
  const uint64_t miss_words = (NL + 31) / 32 + 2;
  if (drop) need += align_up(miss_words * 4, 256);
  uint64_t pin = (lnb + rnb + 2) * 8 * 8 + 8192;
  if (in.l_pre) pin += 2 * ((uint64_t)in.l_pre->n_seg * in.l_pre->chunks * 16 + ((uint64_t)in.l_pre->n_seg + 1) * 8 + 512);
  OLAP_TRY(pinned_begin(ctx, pin));
  OLAP_TRY(ws_begin(ctx, need));
  ColumnSet lcs, rcs;
  if (!in.l_tuples) OLAP_TRY(build_column_set(ctx, lcols, l_pay ? 2 : 1, eb, lrows, lnb, in_mem, pass0_tile(ctx, plan), &lcs));
  if (!in.r_tuples) OLAP_TRY(build_column_set(ctx, rcols, r_pay ? 2 : 1, eb, rrows, rnb, in_mem, pass0_tile(ctx, plan), &rcs));
  uint2* tmp0 = plan.passes >= 2 ? (uint2*)ws_alloc(ctx, (tmp_tuples + 2) * 8) : nullptr;
  uint2* tmp1 = plan.passes >= 3 ? (uint2*)ws_alloc(ctx, (maxN + 2) * 8) : nullptr;
  uint2* Lt = (uint2*)ws_alloc(ctx, (lt_tuples + 2) * 8);
  uint2* Rt = (uint2*)ws_alloc(ctx, (rt_tuples + 2) * 8);
  unsigned long long* d_cnt = (unsigned long long*)ws_alloc(ctx, 40);
  uint32_t *d_fk = out_fk, *d_y = out_y, *d_x = out_x;
  if (out_mem == OLAP_MEM_HOST) {
    d_fk = (uint32_t*)ws_alloc(ctx, NL * 4 + 4);
    d_y = (uint32_t*)ws_alloc(ctx, NL * 4 + 4);
    d_x = (uint32_t*)ws_alloc(ctx, NL * 4 + 4);
  }
  if ((plan.passes >= 2 && !tmp0) || (plan.passes >= 3 && !tmp1) || !Lt || !Rt || !d_cnt || !d_fk || !d_y || !d_x)
    return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (join buffers)");
  OLAP_HIP(ctx, hipMemsetAsync(d_cnt, 0, 40, ctx->stream));
  uint32_t* miss_bits = nullptr;  // inner-join mode: one bit per output row, set where the probe missed
  if (drop) {
    miss_bits = (uint32_t*)ws_alloc(ctx, miss_words * 4);
    if (!miss_bits) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (miss bitmap)");
    OLAP_HIP(ctx, hipMemsetAsync(miss_bits, 0, miss_words * 4, ctx->stream));
  }
  SideOut lso, rso;
  OLAP_TRY(partition_side(ctx, lcs, l_pay, in.l_tuples, NL, plan, (uint32_t)ctx->hash_skip_bits, tmp0, tmp1, Lt, lcap,
                          d_cnt + 4, &lso, in.l_pre, lclaim));
  if (ctx->right_ready) {  // the right table may still be arriving (olap_join_set_right_ready_event)
    OLAP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->right_ready, 0));
    ctx->right_ready = nullptr;
  }
  OLAP_TRY(partition_side(ctx, rcs, r_pay, in.r_tuples, NR, plan, (uint32_t)ctx->hash_skip_bits, tmp0, tmp1, Rt, rcap,
                          d_cnt + 4, &rso, in.r_pre, rclaim));
  {
    PhaseScope ps(ctx, PH_JOIN_PROBE);
    uint64_t grid = (uint64_t)ctx->compute_units * 8;
    if (grid > P) grid = P;
    const PartView rv{Rt, rso.off, rso.cnt, rso.dense, rso.cap};
    const PartView lv{Lt, lso.off, lso.cnt, lso.dense, lso.cap};
    const uint32_t pair_ok = (uint32_t)(((((uintptr_t)d_fk) | ((uintptr_t)d_y) | ((uintptr_t)d_x)) & 7) == 0);
    if (direct) {
      uint64_t dgrid = (uint64_t)ctx->compute_units * 8;
      if (dgrid > P) dgrid = P;
      hipLaunchKernelGGL(k_join_probe_direct, dim3((unsigned)dgrid), dim3(kDBlock), 0, ctx->stream, rv, lv, P, table_bits,
                         d_fk, d_y, d_x, pair_ok, d_cnt, miss_bits);
    } else {
      hipLaunchKernelGGL(k_join_probe, dim3((unsigned)grid), dim3(kJBlock), 0, ctx->stream, rv, lv, P, d_fk, d_y, d_x,
                         pair_ok, d_cnt, miss_bits);
    }
    OLAP_HIP(ctx, hipGetLastError());
  }
  OLAP_HIP(ctx, hipMemcpyAsync(counters, d_cnt, 40, hipMemcpyDeviceToHost, ctx->stream));
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const bool sound = counters[2] == 0 && counters[4] == 0;  // no table / region overflow: the rows written are final
  uint64_t kept_rows = NL;
  if (drop && sound && counters[0]) {
    // inner join: squeeze the marked rows out of the three columns, order preserved
    const uint64_t n_seg = (NL + kCompactSeg - 1) / kCompactSeg;
    uint32_t* d_kept = nullptr;
    uint64_t* d_base = nullptr;
    uint32_t* d_tmp = nullptr;
    std::vector<uint32_t> kept(n_seg);
    std::vector<uint64_t> base(n_seg + 1, 0);
    int rc = OLAP_OK;
    do {
      if (hipMalloc(&d_kept, n_seg * 4) != hipSuccess || hipMalloc(&d_base, n_seg * 8) != hipSuccess) {
        rc = ctx->fail(OLAP_ERR_NOMEM, "device allocation for the row compaction failed");
        break;
      }
      hipLaunchKernelGGL(k_compact_count, dim3((unsigned)n_seg), dim3(256), 0, ctx->stream, (const uint32_t*)miss_bits, NL, d_kept);
      if (hipMemcpyAsync(kept.data(), d_kept, n_seg * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess) {
        rc = ctx->fail(OLAP_ERR_DEVICE, "row compaction (count) failed");
        break;
      }
      for (uint64_t k = 0; k < n_seg; ++k) base[k + 1] = base[k] + kept[k];
      kept_rows = base[n_seg];
      if (kept_rows + counters[0] != NL) {
        rc = ctx->fail(OLAP_ERR_DEVICE, "internal: miss bitmap (%llu rows kept) disagrees with the miss counter (%llu of %llu)",
                       (unsigned long long)kept_rows, counters[0], (unsigned long long)NL);
        break;
      }
      if (hipMalloc(&d_tmp, kept_rows * 4 + 4) != hipSuccess) {
        rc = ctx->fail(OLAP_ERR_NOMEM, "device allocation of %llu bytes for the row compaction failed", (unsigned long long)(kept_rows * 4));
        break;
      }
      if (hipMemcpyAsync(d_base, base.data(), n_seg * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        rc = ctx->fail(OLAP_ERR_DEVICE, "row compaction (bases) failed");
        break;
      }
      for (uint32_t* col : {d_fk, d_y, d_x}) {
        hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)n_seg), dim3(256), 0, ctx->stream, (const uint32_t*)col,
                           (const uint32_t*)miss_bits, (const uint64_t*)d_base, NL, d_tmp);
        if (hipMemcpyAsync(col, d_tmp, kept_rows * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
          rc = ctx->fail(OLAP_ERR_DEVICE, "row compaction (copy back) failed");
          break;
        }
      }
      if (rc == OLAP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ctx->fail(OLAP_ERR_DEVICE, "row compaction failed");
    } while (false);
    (void)hipFree(d_kept);
    (void)hipFree(d_base);
    (void)hipFree(d_tmp);
    if (rc != OLAP_OK) return rc;
  }
  if (rows_out) *rows_out = kept_rows;
  if (out_mem == OLAP_MEM_HOST && sound && (counters[0] == 0 || drop) && kept_rows) {
    PhaseScope ps(ctx, PH_COPY_FROM);
    OLAP_TRY(stage_d2h(ctx, {StageItem{d_fk, out_fk, kept_rows * 4}, StageItem{d_y, out_y, kept_rows * 4},
                             StageItem{d_x, out_x, kept_rows * 4}}));
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (st) {
    st->passes = plan.passes;
    for (int i = 0; i < 4; ++i) st->radix_bits[i] = i < plan.passes ? plan.bits[i] : 0;
    st->nr_partitions = P;
    st->table_slots = direct ? (1u << table_bits) : (uint32_t)kSlots;
    st->left_rows = NL;
    st->right_rows = NR;
    st->max_build_rows = counters[3];
    st->duplicate_keys = counters[1];
    st->missing_keys = counters[0];
  }
  return resolve_timers(ctx);
}

static int join_run(olap_ctx* ctx, const JoinIn& jin, uint64_t NL, uint64_t NR, uint32_t* out_fk, uint32_t* out_y,
                    uint32_t* out_x, uint64_t out_capacity, int out_mem, uint64_t* out_rows, olap_join_stats* stats) {
  if (stats) std::memset(stats, 0, sizeof *stats);
  *out_rows = 0;
  if (NL == 0) return OLAP_OK;
  if (NL > out_capacity) return ctx->fail(OLAP_ERR_CAPACITY, "output capacity %llu < %llu left rows",
                                          (unsigned long long)out_capacity, (unsigned long long)NL);
  if (!out_fk || !out_y || !out_x) return ctx->fail(OLAP_ERR_INVALID, "null output column");
  if (NR == 0) {
    if (stats) { stats->left_rows = NL; stats->missing_keys = NL; }
    if (ctx->join_drop_unmatched) return OLAP_OK;  // inner join with an empty build side: no rows
    return ctx->fail(OLAP_ERR_NO_MATCH, "right table is empty: %llu foreign keys without a match", (unsigned long long)NL);
  }
  uint32_t bits;
  if (ctx->radix_bits_total >= 0) {
    bits = ctx->radix_bits_total < 1 ? 1u : (uint32_t)ctx->radix_bits_total;
  } else {
    bits = (uint32_t)ceil_log2_u64((NR + kTargetBuildRows - 1) / kTargetBuildRows);
    if (bits < 1) bits = 1;
    // Large build sides: stop at the direct-address table's 16 Ki-row partitions (two passes
    // of <= 512 buckets at 2^32 rows) instead of 4 Ki-row partitions for the probing table.
    const int dbits = 32 - kDirectBits - ctx->hash_skip_bits;
    if (ctx->direct_table && dbits >= 1 && dbits < (int)bits && (NR >> dbits) >= 2048) bits = (uint32_t)dbits;
  }
  if (jin.l_pre && bits <= jin.l_pre->bits) bits = jin.l_pre->bits + 1;  // at least one pass gathers the chunks
  unsigned long long counters[5] = {0, 0, 0, 0, 0};
  uint64_t kept_rows = NL;
  bool allow_capacity = true;
  for (int attempt = 0; attempt < 4; ++attempt) {
    OLAP_TRY(join_once(ctx, jin, out_fk, out_y, out_x, out_mem, NL, NR, bits, allow_capacity, stats, counters, &kept_rows));
    if (counters[4]) {  // a fixed-capacity region overflowed (skew): redo with exact partitioning
      allow_capacity = false;
      continue;
    }
    if (counters[2] == 0) break;
    bits += 2;  // a build partition overflowed the LDS table: split finer and retry
    if (bits + (uint32_t)ctx->hash_skip_bits > 28) break;
  }
  if (counters[2])
    return ctx->fail(OLAP_ERR_OVERFLOW, "%llu partition(s) exceed the LDS hash table (%u rows); largest build partition %llu rows",
                     counters[2], kTableCap, counters[3]);
  if (counters[0] && !ctx->join_drop_unmatched)
    return ctx->fail(OLAP_ERR_NO_MATCH, "%llu foreign key(s) without a matching primary key", counters[0]);
  *out_rows = kept_rows;
  return OLAP_OK;
}

}  // namespace olap

using namespace olap;

extern "C" {

int olap_set_tuning(olap_ctx* ctx, const char* key, int value) {
  if (!ctx || !key) return OLAP_ERR_INVALID;
  if (!std::strcmp(key, "scatter_variant")) {
    if (value < 0 || value > 1) return ctx->fail(OLAP_ERR_INVALID, "scatter_variant must be 0 (write-combining) or 1 (tile-staged)");
    ctx->scatter_variant = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "direct_table")) {
    ctx->direct_table = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "wc_geometry")) {
    if (value != 1 && value != 2) return ctx->fail(OLAP_ERR_INVALID, "wc_geometry must be 1 or 2");
    ctx->wc_geometry = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "line16")) {
    ctx->line16 = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "capacity_mode")) {
    ctx->capacity_mode = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "take_variant")) {
    ctx->take_variant = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "claim_mode")) {
    if (value < 0 || value > 2) return ctx->fail(OLAP_ERR_INVALID, "claim_mode must be 0 (off), 1 (auto) or 2 (always)");
    ctx->claim_mode = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "host_stage")) {
    ctx->host_stage = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "host_copy_threads")) {
    if (value < 0 || value > 256) return ctx->fail(OLAP_ERR_INVALID, "host_copy_threads must be in [0, 256]");
    if (value != ctx->host_copy_threads) stager_destroy(ctx);  // rebuilt with the new pool on next use
    ctx->host_copy_threads = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "probe_variant")) {
    ctx->probe_variant = value;
    return OLAP_OK;
  }
  return ctx->fail(OLAP_ERR_INVALID, "unknown tuning key '%s'", key);
}

int olap_join_set_hash_skip_bits(olap_ctx* ctx, int bits) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (bits < 0 || bits > 16) return ctx->fail(OLAP_ERR_INVALID, "hash skip bits must be in [0, 16]");
  ctx->hash_skip_bits = bits;
  return OLAP_OK;
}

int olap_join_set_right_ready_event(olap_ctx* ctx, void* hip_event) {
  if (!ctx) return OLAP_ERR_INVALID;
  ctx->right_ready = static_cast<hipEvent_t>(hip_event);
  return OLAP_OK;
}

int olap_join_set_unmatched(olap_ctx* ctx, int mode) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (mode != OLAP_UNMATCHED_ERROR && mode != OLAP_UNMATCHED_DROP) return ctx->fail(OLAP_ERR_INVALID, "unknown unmatched-row mode %d", mode);
  ctx->join_drop_unmatched = mode == OLAP_UNMATCHED_DROP;
  return OLAP_OK;
}

int olap_join_configure(olap_ctx* ctx, int radix_bits_total, int max_bits_per_pass) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (radix_bits_total > 30 || max_bits_per_pass > kMaxFBits || max_bits_per_pass < 0)
    return ctx->fail(OLAP_ERR_INVALID, "bad join configuration");
  ctx->radix_bits_total = radix_bits_total;
  ctx->max_bits_per_pass = max_bits_per_pass;
  return OLAP_OK;
}

int olap_join_u32(olap_ctx* ctx, const uint32_t* const* left_fk_ptrs, const uint32_t* const* left_y_ptrs,
                  const uint64_t* left_rows, uint64_t n_left_batches, const uint32_t* const* right_pk_ptrs,
                  const uint32_t* const* right_x_ptrs, const uint64_t* right_rows, uint64_t n_right_batches,
                  int in_mem, uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x, uint64_t out_capacity,
                  int out_mem, uint64_t* out_rows, olap_join_stats* stats) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
  if ((in_mem != OLAP_MEM_HOST && in_mem != OLAP_MEM_DEVICE) || (out_mem != OLAP_MEM_HOST && out_mem != OLAP_MEM_DEVICE))
    return ctx->fail(OLAP_ERR_INVALID, "bad memory space");
  if (!out_rows) return ctx->fail(OLAP_ERR_INVALID, "out_rows is null");
  uint64_t NL = 0, NR = 0;
  OLAP_TRY(check_side(ctx, left_fk_ptrs, left_rows, n_left_batches, &NL));
  OLAP_TRY(check_side(ctx, right_pk_ptrs, right_rows, n_right_batches, &NR));
  JoinIn jin;
  jin.lfk = left_fk_ptrs;
  jin.ly = left_y_ptrs;
  jin.lrows = left_rows;
  jin.lnb = n_left_batches;
  jin.rpk = right_pk_ptrs;
  jin.rx = right_x_ptrs;
  jin.rrows = right_rows;
  jin.rnb = n_right_batches;
  jin.in_mem = in_mem;
  return join_run(ctx, jin, NL, NR, out_fk, out_y, out_x, out_capacity, out_mem, out_rows, stats);
}

int olap_join_tuples_u32(olap_ctx* ctx, const uint64_t* left_tuples, uint64_t n_left, const uint64_t* right_tuples,
                         uint64_t n_right, uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x, uint64_t out_capacity,
                         uint64_t* out_rows, olap_join_stats* stats) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
  if (!out_rows) return ctx->fail(OLAP_ERR_INVALID, "out_rows is null");
  if ((n_left && !left_tuples) || (n_right && !right_tuples)) return ctx->fail(OLAP_ERR_INVALID, "null tuple array");
  if ((((uintptr_t)left_tuples) | ((uintptr_t)right_tuples)) & 15)
    return ctx->fail(OLAP_ERR_INVALID, "tuple arrays must be 16-byte aligned");
  if (n_left > (1ull << 32) || n_right > (1ull << 32)) return ctx->fail(OLAP_ERR_INVALID, "more than 2^32 rows on one side");
  JoinIn jin;
  jin.l_tuples = reinterpret_cast<const uint2*>(left_tuples);
  jin.r_tuples = reinterpret_cast<const uint2*>(right_tuples);
  return join_run(ctx, jin, n_left, n_right, out_fk, out_y, out_x, out_capacity, OLAP_MEM_DEVICE, out_rows, stats);
}

static int check_presplit(olap_ctx* ctx, const uint64_t* lo, const uint64_t* hi, uint64_t n_chunks, uint64_t n_tuples,
                          uint64_t* rows) {
  uint64_t t = 0;
  for (uint64_t k = 0; k < n_chunks; ++k) {
    if (lo[k] > hi[k] || hi[k] > n_tuples) return ctx->fail(OLAP_ERR_INVALID, "chunk %llu lies outside the tuple array", (unsigned long long)k);
    t += hi[k] - lo[k];
  }
  if (t > (1ull << 32)) return ctx->fail(OLAP_ERR_INVALID, "more than 2^32 rows on one side");
  *rows = t;
  return OLAP_OK;
}

int olap_join_presplit_u32(olap_ctx* ctx, const uint64_t* left_tuples, uint64_t n_left, const uint64_t* left_chunk_lo,
                           const uint64_t* left_chunk_hi, const uint64_t* right_tuples, uint64_t n_right,
                           const uint64_t* right_chunk_lo, const uint64_t* right_chunk_hi, uint32_t n_segments,
                           uint32_t chunks_per_segment, uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x,
                           uint64_t out_capacity, uint64_t* out_rows, olap_join_stats* stats) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
  if (!out_rows) return ctx->fail(OLAP_ERR_INVALID, "out_rows is null");
  if (n_segments < 1 || (n_segments & (n_segments - 1)) || n_segments > (1u << 16))
    return ctx->fail(OLAP_ERR_INVALID, "n_segments must be a power of two in [1, 2^16]");
  if (chunks_per_segment < 1 || chunks_per_segment > 4096) return ctx->fail(OLAP_ERR_INVALID, "chunks_per_segment must be in [1, 4096]");
  if (!left_chunk_lo || !left_chunk_hi || !right_chunk_lo || !right_chunk_hi) return ctx->fail(OLAP_ERR_INVALID, "null chunk table");
  if ((n_left && !left_tuples) || (n_right && !right_tuples)) return ctx->fail(OLAP_ERR_INVALID, "null tuple array");
  if ((((uintptr_t)left_tuples) | ((uintptr_t)right_tuples)) & 15)
    return ctx->fail(OLAP_ERR_INVALID, "tuple arrays must be 16-byte aligned");
  const uint64_t nchunk = (uint64_t)n_segments * chunks_per_segment;
  uint64_t NL = 0, NR = 0;
  OLAP_TRY(check_presplit(ctx, left_chunk_lo, left_chunk_hi, nchunk, n_left, &NL));
  OLAP_TRY(check_presplit(ctx, right_chunk_lo, right_chunk_hi, nchunk, n_right, &NR));
  Presplit lp, rp;
  lp.n_seg = rp.n_seg = n_segments;
  lp.chunks = rp.chunks = chunks_per_segment;
  lp.bits = rp.bits = (uint32_t)ceil_log2_u64(n_segments);
  lp.lo = left_chunk_lo;
  lp.hi = left_chunk_hi;
  rp.lo = right_chunk_lo;
  rp.hi = right_chunk_hi;
  JoinIn jin;
  jin.l_tuples = reinterpret_cast<const uint2*>(left_tuples);
  jin.r_tuples = reinterpret_cast<const uint2*>(right_tuples);
  jin.l_pre = &lp;
  jin.r_pre = &rp;
  return join_run(ctx, jin, NL, NR, out_fk, out_y, out_x, out_capacity, OLAP_MEM_DEVICE, out_rows, stats);
}

// Shared body of olap_partition_u32 (keys/payload columns out) and
// olap_partition_tuples_u32 (zipped tuples straight into the caller's device buffer).
static int partition_run(olap_ctx* ctx, const uint32_t* const* key_ptrs, const uint32_t* const* payload_ptrs,
                         const uint64_t* rows, uint64_t n_batches, int in_mem, uint32_t nr_partitions,
                         uint32_t* out_keys, uint32_t* out_payload, uint2* out_tuples, uint64_t out_capacity,
                         int out_mem, uint64_t* out_offsets) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
  if ((in_mem != OLAP_MEM_HOST && in_mem != OLAP_MEM_DEVICE) || (out_mem != OLAP_MEM_HOST && out_mem != OLAP_MEM_DEVICE))
    return ctx->fail(OLAP_ERR_INVALID, "bad memory space");
  // dpu/join/main.c:87-89: power of two, at least 2
  if (nr_partitions < 2 || (nr_partitions & (nr_partitions - 1)) || nr_partitions > (1u << 20))
    return ctx->fail(OLAP_ERR_INVALID, "nr_partitions must be a power of two in [2, 2^20]");
  if (!out_offsets) return ctx->fail(OLAP_ERR_INVALID, "out_offsets is null");
  uint64_t N = 0;
  OLAP_TRY(check_side(ctx, key_ptrs, rows, n_batches, &N));
  if (N > out_capacity) return ctx->fail(OLAP_ERR_CAPACITY, "output capacity too small");
  const bool zipped = out_tuples != nullptr;
  if (N && !zipped && (!out_keys || !out_payload)) return ctx->fail(OLAP_ERR_INVALID, "null output column");
  if (zipped && (((uintptr_t)out_tuples) & 15)) return ctx->fail(OLAP_ERR_INVALID, "out_tuples must be 16-byte aligned");
  PassPlan plan;
  OLAP_TRY(make_plan(ctx, (uint32_t)ceil_log2_u64(nr_partitions), &plan));
  const uint32_t eb[2] = {4, 4};
  const int ncols = payload_ptrs ? 2 : 1;
  const void* const* cols[2] = {(const void* const*)key_ptrs, (const void* const*)payload_ptrs};
  uint64_t need = column_set_bytes(rows, n_batches, ncols, eb, in_mem) + side_scratch_bytes(ctx, N, n_batches, plan) +
                  align_up((N + 2) * 8, 256) * (uint64_t)(plan.passes >= 3 ? 3 : plan.passes) + 8192;
  if (out_mem == OLAP_MEM_HOST) need += 2 * align_up(N * 4 + 4, 256);
  OLAP_TRY(pinned_begin(ctx, (n_batches + 2) * 8 * 4 + ((uint64_t)nr_partitions + 1) * 8 + 8192));
  OLAP_TRY(ws_begin(ctx, need));
  ColumnSet cs;
  OLAP_TRY(build_column_set(ctx, cols, ncols, eb, rows, n_batches, in_mem, pass0_tile(ctx, plan), &cs));
  uint2* tmp0 = plan.passes >= 2 ? (uint2*)ws_alloc(ctx, (N + 2) * 8) : nullptr;
  uint2* tmp1 = plan.passes >= 3 ? (uint2*)ws_alloc(ctx, (N + 2) * 8) : nullptr;
  uint2* fin = zipped ? out_tuples : (uint2*)ws_alloc(ctx, (N + 2) * 8);
  uint32_t *d_k = out_keys, *d_p = out_payload;
  if (!zipped && out_mem == OLAP_MEM_HOST) {
    d_k = (uint32_t*)ws_alloc(ctx, N * 4 + 4);
    d_p = (uint32_t*)ws_alloc(ctx, N * 4 + 4);
  }
  if ((plan.passes >= 2 && !tmp0) || (plan.passes >= 3 && !tmp1) || (N && !fin) || (N && !zipped && (!d_k || !d_p)))
    return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (partition buffers)");
  SideOut pso;
  OLAP_TRY(partition_side(ctx, cs, payload_ptrs != nullptr, nullptr, 0, plan, (uint32_t)ctx->hash_skip_bits, tmp0, tmp1,
                          fin, false, nullptr, &pso));
  if (N && !zipped) {
    PhaseScope ps(ctx, PH_PART_SCATTER);
    uint64_t grid = (N + 255) / 256;
    const uint64_t cap = (uint64_t)ctx->compute_units * 16;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k_unzip, dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const uint2*)fin, N, d_k, d_p);
    OLAP_HIP(ctx, hipGetLastError());
  }
  OLAP_HIP(ctx, hipMemcpyAsync(out_offsets, pso.off, ((uint64_t)nr_partitions + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (!zipped && out_mem == OLAP_MEM_HOST && N) {
    PhaseScope ps(ctx, PH_COPY_FROM);
    OLAP_TRY(stage_d2h(ctx, {StageItem{d_k, out_keys, N * 4}, StageItem{d_p, out_payload, N * 4}}));
  }
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return resolve_timers(ctx);
}

int olap_partition_u32(olap_ctx* ctx, const uint32_t* const* key_ptrs, const uint32_t* const* payload_ptrs,
                       const uint64_t* rows, uint64_t n_batches, int in_mem, uint32_t nr_partitions,
                       uint32_t* out_keys, uint32_t* out_payload, uint64_t out_capacity, int out_mem,
                       uint64_t* out_offsets) {
  return partition_run(ctx, key_ptrs, payload_ptrs, rows, n_batches, in_mem, nr_partitions, out_keys, out_payload,
                       nullptr, out_capacity, out_mem, out_offsets);
}

int olap_partition_tuples_u32(olap_ctx* ctx, const uint32_t* const* key_ptrs, const uint32_t* const* payload_ptrs,
                              const uint64_t* rows, uint64_t n_batches, int in_mem, uint32_t nr_partitions,
                              uint64_t* out_tuples, uint64_t out_capacity, uint64_t* out_offsets) {
  if (ctx && !out_tuples) {
    uint64_t n = 0;
    for (uint64_t b = 0; rows && b < n_batches; ++b) n += rows[b];
    if (n) return ctx->fail(OLAP_ERR_INVALID, "out_tuples is null");
    static uint64_t dummy[2] __attribute__((aligned(16)));
    out_tuples = dummy;  // nothing is written for an empty input
  }
  return partition_run(ctx, key_ptrs, payload_ptrs, rows, n_batches, in_mem, nr_partitions, nullptr, nullptr,
                       reinterpret_cast<uint2*>(out_tuples), out_capacity, OLAP_MEM_DEVICE, out_offsets);
}

}  // extern "C"
