// ring_scatter.h -- the radix scatter of the large-join plan (included by join.hip only).
//
// Same contract as k_part_scatter_claim (pass 0, no histogram, claimed allotments + fillers) and
// k_part_scatter_wc<.., CAPM> (last pass, capacity layout) of radix_kernels.h -- reference:
// dpu/shared/kernels/partition.c:167-294 (partition_array / write_new_output) and the shuffle
// of host/partition/partitioner.cc:280-375 -- but built around per-bucket RINGS in LDS:
//
//   * bucket b owns a ring of 16384 / F tuples (two 128-byte lines at 512 buckets, more for smaller
//     fan-outs: the rings always fill the same 128 KiB).  A tuple is inserted with ONE
//     returning LDS atomic on cl[b] = (tuples in the ring | ring head << 16) followed by one
//     8-byte LDS store: there is no tile staging area, no rank/scan over the buckets, no
//     residue copy.  With 512 buckets and 128-byte lines the rings take 128 KiB, one workgroup
//     of 1024 threads per CU.
//   * waves are specialised.  PRODUCER waves (the first NPROD) only issue global LOADS: tile
//     i + 2 is in flight while tile i is inserted and tile i + 1 is hashed.  FLUSHER waves own
//     the buckets (one per lane) and only issue global STORES: after the insert phase every
//     owner whose ring holds a complete line hands it to its wave, which writes 8 lanes x
//     16 B = one aligned 128-byte line per bucket (LINE = 16) -- the granularity at which
//     scattered writes run at copy speed on MI355X (profiles/r1_ubench_stream_fanout.txt:
//     64-byte lines 3.6-3.9 TB/s, 128-byte lines 5.1-5.4 TB/s).  Loads and stores sit in
//     different waves' vmcnt queues, so waiting for a tile never drains the stores of the
//     previous one (stores and loads share vmcnt on gfx950).
//   * two workgroup barriers per tile: insert | flush (+ hashing of the next tile).
//   * claiming mode: an owner writes into the allotment it holds and keeps ONE more in reserve.  The
//     reserve is refilled by the PRODUCER thread of that bucket (global atomic on the bucket's cursor,
//     issued beside its loads and collected with them one tile later, handed over through LDS), so the
//     atomic's round trip is never waited for and the flushers issue nothing but stores.
//   * skew: a bucket that receives more than a ring's worth in one tile ("hot") flushes both
//     lines; the tuples that did not fit are written by their producers straight to a range
//     the owner reserved for them (one extra barrier, only in tiles where that happens).
#pragma once
#include "radix_kernels.h"

namespace olap {

constexpr uint32_t kRingNone = 0xFFFFFFFFu;

#ifdef OLAP_EXP_PHASE_CLOCKS
// Timing experiment (scripts/build_variant.sh + scripts/exp_join_timing.py --clocks): shader-clock sums of the ring scatter's
// phases, taken by one producer wave and one flusher wave of every 16th workgroup.
// [role 0 claim / 1 capacity][0] producer: insert, [1] producer: wait at barrier 1, [2] producer: hash + loads, [3] producer: wait at
// barrier 2, [4] flusher: wait at barrier 1, [5] flusher: flush, [6] flusher: wait at barrier 2, [7] tiles counted
__device__ unsigned long long g_ring_clk[2][16];  // ... [8] producer: wait for tile + 1, [9] hash, [10] claims, [11] load issue; [12] flusher: line rounds, [13] owner update
#define OLAP_CLK(var) const unsigned long long var = __builtin_readcyclecounter()
#define OLAP_CLK_ADD(role, slot, val) do { if (clk_on) atomicAdd(&g_ring_clk[role][slot], (unsigned long long)(val)); } while (0)
#else
#define OLAP_CLK(var) do {} while (0)
#define OLAP_CLK_ADD(role, slot, val) do {} while (0)
#endif

template <int FMAX, int LINE, int BLOCK, int NPROD>
struct RingCfg {
  static constexpr int kRingMin = 2 * LINE;        // tuples per bucket ring at FMAX buckets; F buckets: FMAX * kRingMin / F
  static constexpr int kNW = BLOCK / kWave;
  static constexpr int kNF = kNW - NPROD;          // flusher waves
  static constexpr int kProd = NPROD * kWave;      // producer threads
  static_assert(FMAX <= kNF * kWave, "one bucket per flusher lane");
  static_assert(FMAX <= BLOCK, "the closing scan runs one bucket per thread");
  static constexpr int kDummy = 32;                              // invalid tuples (edge tiles, fillers) go to dummy counters / slots: no branches
  static constexpr uint32_t kRingOff = 0;
  static constexpr uint32_t kCl = (FMAX * kRingMin + kDummy) * 8;  // per bucket: count | head << 16
  static constexpr uint32_t kHot = kCl + (FMAX + kDummy) * 4;   // per bucket: hot-tile hand-off {base, full | head << 16}
  static constexpr uint32_t kList = kHot + FMAX * 8;            // per flusher wave: lines to emit {line index, ring offset}
  static constexpr uint32_t kMisc = kList + kNF * kWave * 8;    // hot flag, scan scratch
  static constexpr uint32_t kRsv = kMisc + kNW * 8 + 64;        // CLAIM, per bucket: the allotment claimed ahead (or kRingNone)
  static constexpr uint32_t kWant = kRsv + FMAX * 4;            // CLAIM, per bucket: 1 = claim one more, 2 = claim in flight
  static constexpr uint32_t kBytes = kWant + FMAX * 4;
  static_assert(FMAX <= kProd, "the producers serve the claims, one bucket per thread");
};

// A tile's tuples in the producers' registers.  FULL = 0: scalar arrays filled by load_tile (any source, ragged tiles,
// unaligned buffers).  FULL = 1 / 2: column input of whole aligned tiles with / without a payload column -- the 16-byte
// vectors as loaded (load_tile_full), read component by component.
template <int TPT, int SRC, int FULL>
struct TileRegs {
  uint32_t k[TPT], p[TPT], bk[TPT];
  uint32_t mask;
  __device__ __forceinline__ uint32_t key(int i) const { return k[i]; }
  __device__ __forceinline__ uint32_t pay(int i) const { return p[i]; }
  __device__ __forceinline__ void set_key(int i, uint32_t v) { k[i] = v; }
  template <int BLOCK, int TILE, bool NT>
  __device__ __forceinline__ void load(const PassSrc& s, const TileRange& r, uint64_t tile, ColsCursor& cur) {
    mask = load_tile<SRC, BLOCK, true, TILE, NT>(s, r, tile, cur, k, p);
  }
  __device__ __forceinline__ void pin() {  // a scheduling barrier for the tile's loads (see step())
#pragma unroll
    for (int i = 0; i < TPT; ++i) asm volatile("" : "+v"(k[i]) : : "memory");
  }
};
template <int TPT, bool HASPAY>
struct TileRegsFull {
  static_assert(TPT == 8, "two key and two payload vectors per thread");
  u32x4_t k0, k1, p0, p1;  // (named members, not arrays: arrays of vectors read by element stayed in scratch memory)
  uint32_t bk[TPT];
  uint32_t mask;
  __device__ __forceinline__ uint32_t key(int i) const { return i < 4 ? k0[i & 3] : k1[i & 3]; }
  __device__ __forceinline__ void set_key(int i, uint32_t v) {
    if (i < 4) k0[i & 3] = v;
    else k1[i & 3] = v;
  }
  __device__ __forceinline__ uint32_t pay(int i) const { return i < 4 ? p0[i & 3] : p1[i & 3]; }
  template <int BLOCK, int TILE, bool NT>
  __device__ __forceinline__ void load(const PassSrc& s, const TileRange&, uint64_t tile, ColsCursor& cur) {
    load_tile_full<BLOCK, TILE, HASPAY, NT>(s, tile, cur, k0, k1, p0, p1);
    mask = (1u << TPT) - 1u;
  }
  __device__ __forceinline__ void pin() {
    asm volatile("" : "+v"(k0), "+v"(k1) : : "memory");
  }
};
template <int TPT>
struct TileRegs<TPT, SRC_COLS, 1> : TileRegsFull<TPT, true> {};
template <int TPT>
struct TileRegs<TPT, SRC_COLS, 2> : TileRegsFull<TPT, false> {};

// SRC: SRC_COLS (pass 0 from the Arrow columns) or SRC_TUPLES.  CLAIM: claimed allotments + fillers
// (any number of workgroups per output bucket; ClaimOut) instead of the capacity layout (one
// workgroup per input segment; CapOut).  Tiles of NPROD * 64 * TPT tuples.
// EXACT (with !CLAIM): the exact layout of a histogram + scan in front (CapOut::hist / seg_off; any number of workgroups per
// segment, bucket order of pass_bucket) -- what k_part_scatter_wc writes, at the ring scatter's rate.  A workgroup's run of a
// bucket starts at an arbitrary tuple S: the bucket's ring starts with S mod 16 phantom tuples, so that ring lines and memory
// lines coincide; the first line (its front belongs to the run before) and the last, partial one leave as 8-byte stores of
// the owner lane, everything between as whole lines.
template <int SRC, bool CLAIM, int FMAX, int LINE, int BLOCK, int NPROD, int TPT, int FULL = 0, bool EXACT = false>
__global__ __launch_bounds__(BLOCK) void k_part_scatter_ring(PassSrc s, uint32_t shift, uint32_t fbits, uint2* out, CapOut co,
                                                               ClaimOut cq) {
  static_assert(!(EXACT && CLAIM), "the exact layout has no allotments");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = RingCfg<FMAX, LINE, BLOCK, NPROD>;
  constexpr int kNF = C::kNF, kProd = C::kProd;
  static_assert(FMAX * C::kRingMin <= 32768, "ring positions and counts are 16-bit fields");
  constexpr int TILE = kProd * TPT;
  constexpr int LPL = LINE / 2;       // lanes per line (16 B each)
  constexpr int LPI = kWave / LPL;    // lines per store instruction
  uint2* ring = (uint2*)(smem + C::kRingOff);
  uint32_t* cl = (uint32_t*)(smem + C::kCl);
  uint2* hot = (uint2*)(smem + C::kHot);
  uint32_t* misc = (uint32_t*)(smem + C::kMisc);
  uint32_t* rsv = (uint32_t*)(smem + C::kRsv);
  uint32_t* want = (uint32_t*)(smem + C::kWant);
  const uint32_t F = 1u << fbits;
  const uint32_t kRing = (uint32_t)(FMAX * C::kRingMin) >> fbits;  // tuples per bucket ring (a power of two >= 2 * LINE)
  TileRange r = block_tiles<SRC, TILE>(s);
  // Column input, interleaved: at any moment the workgroups read neighbouring tiles -- one contiguous window of each
  // column that moves through it, spread over all HBM channels -- instead of 256 streams that start 64 MiB apart and
  // advance in lockstep (every stream on the same channel at the same time).  tile_at(i) = i-th tile of this workgroup.
  // s.interleave = N: the tiles are dealt in chunks of N consecutive tiles (a batch's pointers are looked up once per chunk).
  const bool il = SRC == SRC_COLS && s.interleave != 0;
  const uint64_t il_n = il ? s.interleave : 1, il_total = s.v.total_units;
  if (il) {
    const uint64_t chunks = (il_total + il_n - 1) / il_n;
    r.t0 = 0;
    r.t1 = chunks > blockIdx.x ? (chunks - blockIdx.x + gridDim.x - 1) / gridDim.x * il_n : 0;
  }
#ifdef OLAP_EXP_NOLOAD  // timing experiment: every workgroup re-reads its first four tiles (cache resident)
  auto tile_at = [&](uint64_t i) { return il ? (((i & 3) / il_n) * gridDim.x + blockIdx.x) * il_n + (i & 3) % il_n : r.t0 + ((i - r.t0) & 3); };
#else
  auto tile_at = [&](uint64_t i) { return il ? ((i / il_n) * gridDim.x + blockIdx.x) * il_n + i % il_n : i; };
#endif
  auto tile_ok = [&](uint64_t i) { return i < r.t1 && (!il || tile_at(i) < il_total); };
  if (CLAIM && r.t0 >= r.t1) return;  // nothing to read: claim nothing (uniform)
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x / kWave;
  const uint32_t cap = CLAIM ? cq.cap : co.cap;
  unsigned long long* const overflow = CLAIM ? cq.overflow : co.overflow;
  // where tuple p of output region `region` lies (CLAIM: possibly woven regions, radix_kernels.h woven_at)
  const uint32_t wv_shift = CLAIM ? cq.weave_shift : 0u, wv_n = CLAIM ? cq.weave_n : 0u;
  auto at = [&](uint64_t region, uint32_t p) { return woven_at(region, p, cap, wv_shift, wv_n); };
  // where this workgroup's run of bucket b starts (EXACT)
  auto run_start = [&](uint32_t b) -> uint64_t { return co.seg_off[(uint64_t)r.seg * F + b] + co.hist[(uint64_t)blockIdx.x * F + b]; };
  for (uint32_t b = threadIdx.x; b < FMAX + C::kDummy; b += BLOCK) cl[b] = (EXACT && b < F) ? (uint32_t)(run_start(b) & (LINE - 1)) : 0u;
  if (CLAIM) {
    for (uint32_t b = threadIdx.x; b < FMAX; b += BLOCK) {
      rsv[b] = kRingNone;
      want[b] = 0;
    }
  }
  if (threadIdx.x == 0) misc[0] = 0;
  __syncthreads();

  if (wave < (unsigned)NPROD) {
    // ------------------------------------------------------------ producers
    const uint32_t expect = r.seg & s.chk_mask;
    const uint32_t dummy = FMAX + (lane & (C::kDummy - 1));  // bucket id of an invalid tuple: counter cl[dummy], slot ring[FMAX * kRingMin + ...]
    using Regs = TileRegs<TPT, SRC, FULL>;
    auto hash_tile = [&](Regs& t) {
#pragma unroll
      for (int i = 0; i < TPT; ++i) {
        const uint32_t h = wang_hash(t.key(i));
        // fillers of a claiming producer pass carry another bucket's hash bits (chk_mask 0: none)
        const bool ok = (((t.mask >> i) & 1u) != 0) & (((h >> s.chk_shift) & s.chk_mask) == expect);
        t.bk[i] = ok ? (EXACT ? pass_bucket(s, h, shift, F) : ((h >> shift) & (F - 1u))) : dummy;
      }
    };
    ColsCursor cur;
    Regs A, B, Cc;
    A.mask = B.mask = Cc.mask = 0;
#pragma unroll
    for (int i = 0; i < TPT; ++i) A.bk[i] = B.bk[i] = Cc.bk[i] = dummy;
    if (tile_ok(r.t0)) A.template load<kProd, TILE, OLAP_RING_NT_LOAD != 0>(s, r, tile_at(r.t0), cur);
    if (tile_ok(r.t0 + 1)) B.template load<kProd, TILE, OLAP_RING_NT_LOAD != 0>(s, r, tile_at(r.t0 + 1), cur);
    hash_tile(A);
    // CLAIM: thread t refills the reserve of bucket t
    const uint32_t tb = threadIdx.x;
    uint32_t claimed = 0;   // the atomic's answer, in flight until the next tile's loads have been awaited
    bool pending = false;
#ifdef OLAP_EXP_PHASE_CLOCKS
    const bool clk_on = (blockIdx.x & 15) == 0 && threadIdx.x == 0;
#endif
    auto step = [&](uint64_t tile, Regs& c, Regs& nx, Regs& n2) {
      OLAP_CLK(c0);
      // insert tile `tile`: one returning LDS atomic + one LDS store per tuple
      uint32_t v[TPT];
#pragma unroll
      for (int i = 0; i < TPT; ++i) v[i] = atomicAdd(&cl[c.bk[i]], 1u);
      uint32_t dmask = 0;
#pragma unroll
      for (int i = 0; i < TPT; ++i) {
        const uint32_t sl = v[i] & 0xFFFFu, hd = v[i] >> 16;
        const bool valid = c.bk[i] < (uint32_t)FMAX;
        const bool fits = sl < kRing;
        const uint32_t at = (valid & fits) ? c.bk[i] * kRing + ((hd + sl) & (kRing - 1)) : FMAX * C::kRingMin + (lane & (C::kDummy - 1));
        ring[at] = make_uint2(c.key(i), c.pay(i));
        dmask |= (valid & !fits) ? 1u << i : 0u;  // the ring is full: this tuple is placed after the flush
      }
      if (dmask) misc[0] = (uint32_t)tile + 1u;
#ifdef OLAP_EXP_PHASE_CLOCKS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      OLAP_CLK(c1);
      __syncthreads();  // ---- barrier 1: the flushers take over the rings
      OLAP_CLK(c2);
      // Pin the schedule: tile + 1 is awaited and hashed HERE, beside the flush (the compiler otherwise moves
      // the hashing in front of barrier 1 or down to the next insert), and tile + 2 is requested only after
      // tile + 1 has arrived, so that the wait for tile + 1 (vmcnt(0)) never waits for loads issued a moment ago.
#ifdef OLAP_EXP_PHASE_CLOCKS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      nx.pin();
      OLAP_CLK(c2a);
      hash_tile(nx);
#pragma unroll
      for (int i = 0; i < TPT; ++i) asm volatile("" : "+v"(nx.bk[i]) : : "memory");
      OLAP_CLK(c2b);
      if (CLAIM) {  // (every VMEM operation of this wave has completed at this point)
        if (pending) {
          rsv[tb] = claimed;
          want[tb] = 0;
          pending = false;
        }
        if (tb < F && want[tb] == 1u) {
          claimed = atomicAdd(&cq.cursor[(uint64_t)r.seg * F + tb], kClaimChunk);
          want[tb] = 2u;
          pending = true;
        }
      }
      OLAP_CLK(c2c);
      n2.mask = 0;
      if (tile_ok(tile + 2)) n2.template load<kProd, TILE, OLAP_RING_NT_LOAD != 0>(s, r, tile_at(tile + 2), cur);
      OLAP_CLK(c3);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 8, c2a - c2);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 9, c2b - c2a);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 10, c2c - c2b);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 11, c3 - c2c);
      __syncthreads();  // ---- barrier 2: complete lines are out, the rings have room again
      OLAP_CLK(c4);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 0, c1 - c0);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 1, c2 - c1);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 2, c3 - c2);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 3, c4 - c3);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 7, 1);
      if (misc[0] == (uint32_t)tile + 1u) {  // some bucket was hot in this tile (uniform)
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          if (dmask & (1u << i)) {
            const uint32_t b = c.bk[i];
            const uint2 info = hot[b];
            const uint32_t full = info.y & 0xFFFFu, hd = info.y >> 16, sl = v[i] & 0xFFFFu;
            if (sl < full) {  // goes straight to the range the owner reserved (dropped when the region is full)
              if (EXACT) {
                const uint64_t S = run_start(b);
                gstore2(out + (S - (S & (LINE - 1))) + info.x + (sl - kRing), make_uint2(c.key(i), c.pay(i)));
              } else if (info.x != kRingNone) {
                gstore2(out + at((uint64_t)r.seg * F + b, info.x + (sl - kRing)), make_uint2(c.key(i), c.pay(i)));
              }
            } else {          // tail of the tile: the bucket's new residue
              ring[b * kRing + ((hd + (sl - full)) & (kRing - 1))] = make_uint2(c.key(i), c.pay(i));
            }
          }
        }
        __syncthreads();  // ---- barrier 3 (hot tiles only)
      }
    };
    for (uint64_t tile = r.t0; tile < r.t1; tile += 3) {
      step(tile, A, B, Cc);
      if (tile + 1 >= r.t1) break;
      step(tile + 1, B, Cc, A);
      if (tile + 2 >= r.t1) break;
      step(tile + 2, Cc, A, B);
    }
    if (CLAIM) {
      if (pending) rsv[tb] = claimed;  // delivered before the owners close their regions
      __syncthreads();
    }
  } else {
    // ------------------------------------------------------------- flushers
    const unsigned fw = wave - NPROD;
    const uint32_t bpw = F >= (uint32_t)kNF ? F / kNF : 1u;  // buckets per flusher wave: consecutive buckets, consecutive lanes
    const uint32_t b_own = fw * bpw + lane;
    const bool owner = lane < bpw && b_own < F;
    const uint64_t p_own = (uint64_t)r.seg * F + b_own;
    uint2* list = (uint2*)(smem + C::kList) + fw * kWave;
    uint32_t pos = 0, rem = 0;                   // next line's tuple offset in the region; CLAIM: tuples left in the allotment
    uint32_t fill_lo = 0, fill_hi = 0;           // CLAIM: unused tail of a hot tile's block
    bool dead = !owner;                          // region exhausted (skew): tuples are dropped, *overflow raised
    uint32_t filler = 0;
    uint64_t xbase = 0;                          // EXACT: the memory line this workgroup's run of the bucket starts in, as a tuple index
    uint32_t xc0 = 0;                            // ... and the tuples of that line in front of the run (phantoms in the ring)
    bool first = false;                          // ... the first line has not left yet
    if (EXACT && owner) {
      const uint64_t S = run_start(b_own);
      xc0 = (uint32_t)(S & (LINE - 1));
      xbase = S - xc0;
      first = xc0 != 0;
    }
    if (CLAIM && owner) {
      while (pass_bucket(s, wang_hash(filler), shift, F) == b_own) ++filler;  // a key of another bucket
      const uint32_t got = atomicAdd(&cq.cursor[p_own], 2 * kClaimChunk);
      if ((uint64_t)got + 2 * kClaimChunk > cap) {
        dead = true;
        atomicAdd(overflow, 1ull);
      } else {
        pos = got;
        rem = kClaimChunk;
        rsv[b_own] = got + kClaimChunk;
      }
    }
    // CLAIM: the next allotment -- the reserve (a producer refills it), or claimed on the spot if that has not arrived
    auto next_allotment = [&]() {
      uint32_t a = rsv[b_own];
      if (a != kRingNone) {
        rsv[b_own] = kRingNone;
        want[b_own] = 1u;
      } else {
        a = atomicAdd(&cq.cursor[p_own], kClaimChunk);
      }
      if ((uint64_t)a + kClaimChunk > cap) {
        dead = true;
        atomicAdd(overflow, 1ull);
      } else {
        pos = a;
        rem = kClaimChunk;
      }
    };
    auto flush_round = [&](bool has, uint32_t d, uint32_t ring_line) -> bool {
      const unsigned long long m = __ballot(has);
      if (!m) return false;
      const uint32_t n = (uint32_t)__popcll(m);
      if (has) list[__popcll(m & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)((EXACT ? xbase + d : at(p_own, d)) / LINE), b_own * kRing + ring_line);  // line index: < 2^32 (cap, chunks: multiples of LINE)
      __builtin_amdgcn_wave_barrier();
      for (uint32_t k = 0; k < n; k += LPI) {
        const uint32_t j = k + lane / LPL;
        if (j < n) {
          const uint2 e = list[j];
          const uint4 t = *(const uint4*)(ring + e.y + 2 * (lane % LPL));
#ifdef OLAP_EXP_NOSTORE  // timing experiment (scripts/build_variant.sh, scripts/exp_join_timing.py): everything but the line stores of the claiming (1) / capacity (2) pass
          if ((CLAIM ? 1 : 2) != OLAP_EXP_NOSTORE || (t.x == 0x12345u && t.y == 0x6789u && t.z == 0xABCDu)) gstore4((uint4*)out + (uint64_t)e.x * LPL + (lane % LPL), t);
#else
          if (OLAP_RING_NT_STORE) gstore4_nt((uint4*)out + (uint64_t)e.x * LPL + (lane % LPL), t);
          else gstore4((uint4*)out + (uint64_t)e.x * LPL + (lane % LPL), t);
#endif
        }
      }
      __builtin_amdgcn_wave_barrier();
      return true;
    };
#ifdef OLAP_EXP_PHASE_CLOCKS
    const bool clk_on = (blockIdx.x & 15) == 0 && threadIdx.x == NPROD * kWave;
#endif
    for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
      OLAP_CLK(f0);
      __syncthreads();  // ---- barrier 1
      OLAP_CLK(f1);
      const uint32_t vv = owner ? cl[b_own] : 0u;
      const uint32_t total = vv & 0xFFFFu, hd = vv >> 16;
      const bool hotb = total > kRing;
      const uint32_t placed = hotb ? kRing : total;
      const uint32_t nl = placed / LINE;  // complete lines in the ring
      for (uint32_t q = 0;; ++q) {       // one line per bucket and round (two lines per ring at 512 buckets)
        uint32_t dq = kRingNone;
        if (q < nl && !dead) {
          if (CLAIM) {
            if (rem == 0) next_allotment();
            if (!dead) {
              dq = pos;
              pos += LINE;
              rem -= LINE;
            }
          } else if (EXACT) {
            if (first) {  // (q == 0, pos == 0) the line's front belongs to the run before: the run's part as 8-byte stores
              for (uint32_t k = xc0; k < (uint32_t)LINE; ++k) gstore2(out + xbase + k, ring[b_own * kRing + ((hd + k) & (kRing - 1))]);
              first = false;
            } else {
              dq = pos;
            }
            pos += LINE;
          } else if ((uint64_t)pos + LINE > cap) {
            dead = true;
            atomicAdd(overflow, 1ull);
          } else {
            dq = pos;
            pos += LINE;
          }
        }
        if (!flush_round(dq != kRingNone, dq, (hd + q * LINE) & (kRing - 1)) && !__ballot(q + 1 < nl)) break;
      }
      OLAP_CLK(f1a);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 12, f1a - f1);
      if (owner) {
        if (hotb) {  // the tuples beyond the ring: whole lines' worth goes out directly, the rest is the new residue
          const uint32_t full = total & ~(uint32_t)(LINE - 1);
          const uint32_t nd = full - kRing;
          uint32_t dbase = kRingNone;
          if (nd && !dead) {
            if (CLAIM) {
              // nd is a multiple of LINE.  Normally it comes out of the allotment in use, or -- what is left of that one
              // padded with fillers -- out of the reserve: no atomic to wait for (some bucket is hot in most tiles at 512
              // buckets: a block claimed on the spot cost a round trip to the cursor per tile).
              if (nd <= kClaimChunk) {
                if (rem < nd) {
                  fill_lo = pos;
                  fill_hi = pos + rem;
                  next_allotment();
                }
                if (!dead) {
                  dbase = pos;
                  pos += nd;
                  rem -= nd;
                }
              } else {  // a very hot bucket: its own block
                const uint32_t blk = (nd + kClaimChunk - 1) / kClaimChunk * kClaimChunk;
                const uint32_t got = atomicAdd(&cq.cursor[p_own], blk);
                if ((uint64_t)got + blk > cap) {
                  dead = true;
                  atomicAdd(overflow, 1ull);
                } else {
                  dbase = got;
                  fill_lo = got + nd;
                  fill_hi = got + blk;
                }
              }
            } else if (!EXACT && (uint64_t)pos + nd > cap) {
              dead = true;
              atomicAdd(overflow, 1ull);
            } else {
              dbase = pos;
              pos += nd;
            }
          }
          hot[b_own] = make_uint2(dbase, full | (hd << 16));
          cl[b_own] = (total - full) | (hd << 16);
        } else {
          cl[b_own] = (placed - nl * LINE) | (((hd + nl * LINE) & (kRing - 1)) << 16);
        }
      }
      OLAP_CLK(f2);
      __syncthreads();  // ---- barrier 2
      OLAP_CLK(f3);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 4, f1 - f0);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 5, f2 - f1);
      OLAP_CLK_ADD(CLAIM ? 0 : 1, 6, f3 - f2);
      if (misc[0] == (uint32_t)tile + 1u) {
        if (CLAIM) {
          for (uint32_t q = fill_lo; q < fill_hi; q += 2) *(uint4*)(out + at(p_own, q)) = make_uint4(filler, 0u, filler, 0u);
          fill_lo = fill_hi = 0;
        }
        __syncthreads();  // ---- barrier 3 (hot tiles only)
      }
    }
    if (CLAIM) __syncthreads();  // the producers have delivered the claims that were in flight
    // tail: what is left in the rings (< LINE tuples per bucket)
    if (owner) {
      const uint32_t vv = cl[b_own];
      const uint32_t rr = vv & 0xFFFFu, hd = vv >> 16;
      uint2* region = out + p_own * cap;  // (capacity layout; a claimed allotment is addressed through at())
      if (CLAIM) {
        if (!dead && rr) {
          if (rem == 0) next_allotment();
          if (!dead) {  // the residue as one line padded with fillers
            uint2* line = out + at(p_own, pos);
            for (uint32_t j = 0; j < (uint32_t)LINE; ++j)
              line[j] = j < rr ? ring[b_own * kRing + ((hd + j) & (kRing - 1))] : make_uint2(filler, 0u);
            pos += LINE;
            rem -= LINE;
          }
        }
        if (!dead) {  // the rest of the allotment, and the one claimed ahead, hold only fillers
          for (uint32_t q = 0; q < rem; q += 2) *(uint4*)(out + at(p_own, pos + q)) = make_uint4(filler, 0u, filler, 0u);
        }
        const uint32_t ahead = rsv[b_own];
        if (ahead != kRingNone && (uint64_t)ahead + kClaimChunk <= cap)
          for (uint32_t q = 0; q < kClaimChunk; q += 2) *(uint4*)(out + at(p_own, ahead + q)) = make_uint4(filler, 0u, filler, 0u);
      } else if (EXACT) {  // the residue (a first line that never filled still has its phantoms in front)
        for (uint32_t j = first ? xc0 : 0u; j < rr; ++j) gstore2(out + xbase + pos + j, ring[b_own * kRing + ((hd + j) & (kRing - 1))]);
      } else {
        uint32_t count = pos;
        if (!dead) {
          if ((uint64_t)pos + rr > cap) {
            atomicAdd(overflow, 1ull);
          } else {
            for (uint32_t j = 0; j < rr; ++j) region[pos + j] = ring[b_own * kRing + ((hd + j) & (kRing - 1))];
            count = pos + rr;
          }
        }
        ((uint32_t*)hot)[b_own] = count;  // for the closing scan
      }
    }
  }
  if (!CLAIM && !EXACT) {  // partition sizes and dense starts (exclusive scan over this segment's buckets, one per thread)
    __syncthreads();
    const uint32_t t = threadIdx.x;
    const uint64_t count = t < F ? ((const uint32_t*)hot)[t] : 0u;
    uint64_t* wtot = (uint64_t*)(smem + C::kMisc + 16);
    uint64_t incl = count;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint64_t yv = __shfl_up(incl, o, kWave);
      if ((int)lane >= o) incl += yv;
    }
    if (lane == kWave - 1) wtot[wave] = incl;
    __syncthreads();
    uint64_t pre = incl - count;
#pragma unroll 1
    for (unsigned w = 0; w < wave; ++w) pre += wtot[w];
    if (t < F) {
      co.cnt[(uint64_t)r.seg * F + t] = (uint32_t)count;
      co.dense[(uint64_t)r.seg * F + t] = (co.seg_total ? 0 : r.lo) + pre;
    }
    if (co.seg_total && threadIdx.x == 0) {
      uint64_t total = 0;
      for (int w = 0; w < BLOCK / kWave; ++w) total += wtot[w];
      co.seg_total[r.seg] = total;
    }
  }
}

// Geometry of the ring scatter: 1024 threads (8 producer + 8 flusher waves), one workgroup per CU,
// <= 512 buckets, 128-byte lines.  Producers hold kRingTpt tuples per thread and tile.
constexpr int kRingBlock = 1024;
constexpr int kRingProd = 8;
constexpr int kRingFmax = 512;
constexpr int kRingLine = 16;

}  // namespace olap
