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
//   * strided pieces (pass 0 of the large-join plan): the claimed allotments above put a workgroup's consecutive line stores
//     into 512 bucket regions 68 MB apart -- 512 pages, one L1-TLB miss per line (2.9e8 UTCL1 misses per launch against 2.5e3
//     for the capacity pass, profiles/r2_pmc_ring_claim_layout.txt).  In the SP layout every workgroup owns ONE contiguous
//     area and writes bucket b's r-th piece (256 tuples) at (r * F + b) * 256 inside it: all buckets advance at about the
//     same rate, so the lines a workgroup writes at any time lie within a few MiB.  No atomics, no fillers: the
//     workgroup records how many tuples it wrote per bucket (sp_used), and the next pass (SRC_SP) walks the pieces of
//     its bucket through all the areas.
//   * skew: a bucket that receives more than a ring's worth in one tile ("hot") flushes both
//     lines; the tuples that did not fit are written by their producers straight to a range
//     the owner reserved for them (one extra barrier, only in tiles where that happens).
#pragma once
#include "radix_kernels.h"

namespace olap {

constexpr uint32_t kRingNone = 0xFFFFFFFFu;

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
  static constexpr int kSpMaxWgs = 1024;                        // SRC_SP: producer workgroups of the previous pass
  static constexpr uint32_t kSpPfx = kWant + FMAX * 4;          // SRC_SP: pieces before producer w; its tuple count
  static constexpr uint32_t kSpCnt = kSpPfx + (kSpMaxWgs + 1) * 4;
  static constexpr uint32_t kSpTab = ((kSpCnt + kSpMaxWgs * 4 + 15) / 16) * 16;  // SRC_SP: the pieces of the tile being requested {base lo, base hi, valid, -}
  static constexpr uint32_t kBytes = kSpTab + 96 * 16;  // [0,32) tile 0, [32,64) tile 1 (prologue), [64,96) the tile being requested
  static_assert(FMAX <= kProd, "the producers serve the claims, one bucket per thread");
};

enum { RING_CAP = 0, RING_CLAIM = 1, RING_SP = 2 };  // output layout: capacity regions | claimed allotments | strided pieces

// Output of the strided-piece layout (MODE RING_SP).
struct SpOut {
  uint32_t* used;     // [grid * F]: tuples workgroup w wrote for bucket b
  uint64_t area;      // tuples per workgroup area = rounds * F * pt
  uint32_t pt;        // tuples per piece (a multiple of the line, >= 128)
  uint32_t rounds;    // pieces per bucket an area has room for
  unsigned long long* overflow;
};

template <int TPT>
struct TileRegs {
  uint32_t key[TPT], pay[TPT], bk[TPT];
  uint32_t mask;
};

// SRC: SRC_COLS (pass 0 from the Arrow columns), SRC_TUPLES, or SRC_SP (the strided pieces a RING_SP pass wrote).
// MODE: RING_CAP capacity layout (one workgroup per input segment; CapOut), RING_CLAIM claimed allotments + fillers
// (any number of workgroups per output bucket; ClaimOut), RING_SP strided pieces (SpOut).  Tiles of NPROD * 64 * TPT tuples.
template <int SRC, int MODE, int FMAX, int LINE, int BLOCK, int NPROD, int TPT>
__global__ __launch_bounds__(BLOCK) void k_part_scatter_ring(PassSrc s, uint32_t shift, uint32_t fbits, uint2* out, CapOut co,
                                                               ClaimOut cq, SpOut sp) {
  constexpr bool CLAIM = MODE == RING_CLAIM, SP = MODE == RING_SP;
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
  TileRange r;
  if constexpr (SRC == SRC_SP) {
    r.seg = blockIdx.x;
    r.t0 = r.t1 = r.lo = r.hi = r.a = 0;  // t1: set below, from the producers' counts
  } else {
    r = block_tiles<SRC, TILE>(s);
  }
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
  auto tile_at = [&](uint64_t i) { return il ? ((i / il_n) * gridDim.x + blockIdx.x) * il_n + i % il_n : i; };
  auto tile_ok = [&](uint64_t i) { return i < r.t1 && (!il || tile_at(i) < il_total); };
  if (CLAIM && r.t0 >= r.t1) return;  // nothing to read: claim nothing (uniform)
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x / kWave;
  const uint32_t cap = CLAIM ? cq.cap : co.cap;
  unsigned long long* const overflow = CLAIM ? cq.overflow : SP ? sp.overflow : co.overflow;
  // Tuple index of logical position `p` of bucket `b`'s output stream of this workgroup.
  const uint64_t sp_base = SP ? (uint64_t)blockIdx.x * sp.area : 0;
  auto out_index = [&](uint32_t b, uint32_t p) -> uint64_t {
    if (SP) return sp_base + ((uint64_t)(p / sp.pt) * F + b) * sp.pt + p % sp.pt;
    return ((uint64_t)r.seg * F + b) * cap + p;
  };
  const uint32_t room = SP ? sp.rounds * sp.pt : cap;  // tuples a bucket's stream may hold
  // SRC_SP: the pieces of this segment, producer by producer
  uint32_t* sp_pfx = (uint32_t*)(smem + C::kSpPfx);
  uint32_t* sp_cnt = (uint32_t*)(smem + C::kSpCnt);
  uint4* sp_tab = (uint4*)(smem + C::kSpTab);
  const uint32_t sp_ppt = SRC == SRC_SP ? (uint32_t)TILE / s.sp_pt : 0;  // pieces per tile
  if constexpr (SRC == SRC_SP) {
    for (uint32_t w = threadIdx.x; w < s.sp_wgs; w += BLOCK) sp_cnt[w] = s.sp_used[(uint64_t)w * s.sp_f + r.seg];
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t run = 0;
      for (uint32_t w = 0; w < s.sp_wgs; ++w) {
        sp_pfx[w] = run;
        run += (sp_cnt[w] + s.sp_pt - 1) / s.sp_pt;
      }
      sp_pfx[s.sp_wgs] = run;
    }
    __syncthreads();
    r.t1 = (sp_pfx[s.sp_wgs] + sp_ppt - 1) / sp_ppt;
  }
  // piece number q of the segment -> {first tuple index, valid tuples}
  auto sp_piece = [&](uint32_t q) -> uint4 {
    if (q >= sp_pfx[s.sp_wgs]) return make_uint4(0u, 0u, 0u, 0u);
    uint32_t lo = 0, hi = s.sp_wgs;  // last w with sp_pfx[w] <= q
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (sp_pfx[mid] <= q) lo = mid; else hi = mid;
    }
    const uint32_t rnd = q - sp_pfx[lo];
    const uint64_t base = (uint64_t)lo * s.sp_area + ((uint64_t)rnd * s.sp_f + r.seg) * s.sp_pt;
    const uint32_t left = sp_cnt[lo] - rnd * s.sp_pt;
    return make_uint4((uint32_t)base, (uint32_t)(base >> 32), left < s.sp_pt ? left : s.sp_pt, 0u);
  };
  if constexpr (SRC == SRC_SP) {  // the pieces of the first two tiles (sp_ppt <= 32: pieces hold >= 128 tuples)
    if (threadIdx.x < sp_ppt) {
      sp_tab[threadIdx.x] = sp_piece(threadIdx.x);
      sp_tab[32 + threadIdx.x] = sp_piece(sp_ppt + threadIdx.x);
    }
    __syncthreads();
  }
  for (uint32_t b = threadIdx.x; b < FMAX + C::kDummy; b += BLOCK) cl[b] = 0;
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
    auto hash_tile = [&](TileRegs<TPT>& t) {
#pragma unroll
      for (int i = 0; i < TPT; ++i) {
        const uint32_t h = wang_hash(t.key[i]);
        // fillers of a claiming producer pass carry another bucket's hash bits (chk_mask 0: none)
        const bool ok = (((t.mask >> i) & 1u) != 0) & (((h >> s.chk_shift) & s.chk_mask) == expect);
        t.bk[i] = ok ? ((h >> shift) & (F - 1u)) : dummy;
      }
    };
    ColsCursor cur;
    TileRegs<TPT> A, B, Cc;
    A.mask = B.mask = Cc.mask = 0;
#pragma unroll
    for (int i = 0; i < TPT; ++i) A.bk[i] = B.bk[i] = Cc.bk[i] = dummy;
    // one tile into registers; SRC_SP: its pieces are listed at sp_tab[tab ...] (LDS, written a barrier ago)
    auto load_any = [&](uint64_t i, TileRegs<TPT>& t, uint32_t tab) {
      t.mask = 0;
      if (!tile_ok(i)) return;
      if constexpr (SRC == SRC_SP) {
#pragma unroll
        for (int j = 0; j < TPT / 2; ++j) {
          const uint32_t e = (j * kProd + threadIdx.x) * 2;       // tuple offset inside the tile (a wave stays inside one piece)
          const uint32_t pi = e / s.sp_pt, ei = e % s.sp_pt;
          const uint4 pc = sp_tab[tab + pi];
          uint4 v = make_uint4(0, 0, 0, 0);
          if (ei < pc.z) v = gload4(s.tuples + (((uint64_t)pc.y << 32) | pc.x) + ei);
          t.key[2 * j] = v.x;
          t.pay[2 * j] = v.y;
          t.key[2 * j + 1] = v.z;
          t.pay[2 * j + 1] = v.w;
          t.mask |= (ei < pc.z ? 1u : 0u) << (2 * j);
          t.mask |= (ei + 1 < pc.z ? 1u : 0u) << (2 * j + 1);
        }
      } else {
        t.mask = load_tile<SRC, kProd, true, TILE, OLAP_RING_NT_LOAD != 0>(s, r, tile_at(i), cur, t.key, t.pay);
      }
    };
    load_any(r.t0, A, 0);
    load_any(r.t0 + 1, B, 32);
    hash_tile(A);
    // CLAIM: thread t refills the reserve of bucket t
    const uint32_t tb = threadIdx.x;
    uint32_t claimed = 0;   // the atomic's answer, in flight until the next tile's loads have been awaited
    bool pending = false;
    auto step = [&](uint64_t tile, TileRegs<TPT>& c, TileRegs<TPT>& nx, TileRegs<TPT>& n2) {
      if constexpr (SRC == SRC_SP) {  // the pieces of tile + 2, for the loads behind barrier 1
        if (threadIdx.x < sp_ppt) sp_tab[64 + threadIdx.x] = sp_piece((uint32_t)(tile + 2) * sp_ppt + threadIdx.x);
      }
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
        ring[at] = make_uint2(c.key[i], c.pay[i]);
        dmask |= (valid & !fits) ? 1u << i : 0u;  // the ring is full: this tuple is placed after the flush
      }
      if (dmask) misc[0] = (uint32_t)tile + 1u;
      __syncthreads();  // ---- barrier 1: the flushers take over the rings
      // Pin the schedule: tile + 1 is awaited and hashed HERE, beside the flush (the compiler otherwise moves
      // the hashing in front of barrier 1 or down to the next insert), and tile + 2 is requested only after
      // tile + 1 has arrived, so that the wait for tile + 1 (vmcnt(0)) never waits for loads issued a moment ago.
#pragma unroll
      for (int i = 0; i < TPT; ++i) asm volatile("" : "+v"(nx.key[i]) : : "memory");
      hash_tile(nx);
#pragma unroll
      for (int i = 0; i < TPT; ++i) asm volatile("" : "+v"(nx.bk[i]) : : "memory");
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
      load_any(tile + 2, n2, 64);
      __syncthreads();  // ---- barrier 2: complete lines are out, the rings have room again
      if (misc[0] == (uint32_t)tile + 1u) {  // some bucket was hot in this tile (uniform)
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          if (dmask & (1u << i)) {
            const uint32_t b = c.bk[i];
            const uint2 info = hot[b];
            const uint32_t full = info.y & 0xFFFFu, hd = info.y >> 16, sl = v[i] & 0xFFFFu;
            if (sl < full) {  // goes straight to the range the owner reserved (dropped when the region is full)
              if (info.x != kRingNone) gstore2(out + out_index(b, info.x + (sl - kRing)), make_uint2(c.key[i], c.pay[i]));
            } else {          // tail of the tile: the bucket's new residue
              ring[b * kRing + ((hd + (sl - full)) & (kRing - 1))] = make_uint2(c.key[i], c.pay[i]);
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
    // line number of logical position d (a multiple of LINE) of this bucket's stream
    auto line_of = [&](uint32_t d) { return (uint32_t)(out_index(b_own, d) / LINE); };
    uint2* list = (uint2*)(smem + C::kList) + fw * kWave;
    uint32_t pos = 0, rem = 0;                   // next line's tuple offset in the region; CLAIM: tuples left in the allotment
    uint32_t fill_lo = 0, fill_hi = 0;           // CLAIM: unused tail of a hot tile's block
    bool dead = !owner;                          // region exhausted (skew): tuples are dropped, *overflow raised
    uint32_t filler = 0;
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
      if (has) list[__popcll(m & ((1ull << lane) - 1ull))] = make_uint2(line_of(d), b_own * kRing + ring_line);
      __builtin_amdgcn_wave_barrier();
      for (uint32_t k = 0; k < n; k += LPI) {
        const uint32_t j = k + lane / LPL;
        if (j < n) {
          const uint2 e = list[j];
          const uint4 t = *(const uint4*)(ring + e.y + 2 * (lane % LPL));
          if (OLAP_RING_NT_STORE) gstore4_nt((uint4*)out + (uint64_t)e.x * LPL + (lane % LPL), t);
          else gstore4((uint4*)out + (uint64_t)e.x * LPL + (lane % LPL), t);
        }
      }
      __builtin_amdgcn_wave_barrier();
      return true;
    };
    for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
      __syncthreads();  // ---- barrier 1
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
          } else if ((uint64_t)pos + LINE > room) {
            dead = true;
            atomicAdd(overflow, 1ull);
          } else {
            dq = pos;
            pos += LINE;
          }
        }
        if (!flush_round(dq != kRingNone, dq, (hd + q * LINE) & (kRing - 1)) && !__ballot(q + 1 < nl)) break;
      }
      if (owner) {
        if (hotb) {  // the tuples beyond the ring: whole lines' worth goes out directly, the rest is the new residue
          const uint32_t full = total & ~(uint32_t)(LINE - 1);
          const uint32_t nd = full - kRing;
          uint32_t dbase = kRingNone;
          if (nd && !dead) {
            if (CLAIM) {
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
            } else if ((uint64_t)pos + nd > room) {
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
      __syncthreads();  // ---- barrier 2
      if (misc[0] == (uint32_t)tile + 1u) {
        if (CLAIM) {
          for (uint32_t q = fill_lo; q < fill_hi; q += 2) *(uint4*)(out + p_own * cap + q) = make_uint4(filler, 0u, filler, 0u);
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
      uint2* region = out + p_own * cap;
      if (CLAIM) {
        if (!dead && rr) {
          if (rem == 0) next_allotment();
          if (!dead) {  // the residue as one line padded with fillers
            for (uint32_t j = 0; j < (uint32_t)LINE; ++j)
              region[pos + j] = j < rr ? ring[b_own * kRing + ((hd + j) & (kRing - 1))] : make_uint2(filler, 0u);
            pos += LINE;
            rem -= LINE;
          }
        }
        if (!dead) {  // the rest of the allotment, and the one claimed ahead, hold only fillers
          for (uint32_t q = 0; q < rem; q += 2) *(uint4*)(region + pos + q) = make_uint4(filler, 0u, filler, 0u);
        }
        const uint32_t ahead = rsv[b_own];
        if (ahead != kRingNone && (uint64_t)ahead + kClaimChunk <= cap)
          for (uint32_t q = 0; q < kClaimChunk; q += 2) *(uint4*)(region + ahead + q) = make_uint4(filler, 0u, filler, 0u);
      } else {
        uint32_t count = pos;
        if (!dead) {
          if ((uint64_t)pos + rr > room) {
            atomicAdd(overflow, 1ull);
          } else {
            for (uint32_t j = 0; j < rr; ++j) out[out_index(b_own, pos + j)] = ring[b_own * kRing + ((hd + j) & (kRing - 1))];
            count = pos + rr;
          }
        }
        if (SP) sp.used[(uint64_t)blockIdx.x * F + b_own] = count;  // what the next pass reads of this bucket here
        else ((uint32_t*)hot)[b_own] = count;                       // for the closing scan
      }
    }
  }
  if (MODE == RING_CAP) {  // partition sizes and dense starts (exclusive scan over this segment's buckets, one per thread)
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
