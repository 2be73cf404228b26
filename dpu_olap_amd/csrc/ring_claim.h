// ring_claim.h -- pass 0 of the large-join plan from the Arrow columns: the claiming ring scatter of ring_scatter.h
// (k_part_scatter_ring<SRC_COLS, CLAIM = true, ...>; same contract, same output: claimed allotments in woven regions,
// filler tuples, ClaimOut) rebuilt around the instruction count.  Included by join.hip only.
//
// Reference behaviour: dpu/shared/kernels/partition.c:167-294 (partition_array / write_new_output) and the shuffle of
// host/partition/partitioner.cc:280-375, without a histogram pass.
//
// The ring scatter is bound by instruction issue, not by HBM (profiles/r3_ring_phase_clocks.txt: without its line stores
// the pass takes as long as with them; per 4096-tuple tile 2200 cycles of insert and 3500 of flush, waves of either kind
// idle while the other works).  What this kernel does differently:
//   * a tuple's bucket is carried as ONE register, the byte offset of its ring (bucket << log2 ring bytes): the
//     bucket's count word sits at 4 x bucket in front of the rings, so the LDS atomic's address is a shift away and the
//     slot is (offset | position & mask) -- no multiply, no select;
//   * invalid tuples (ragged tails) go to dummy RINGS, so the insert has no per-tuple validity test; a ring that
//     overflows is found with one max + ballot per tile, and only then does the wave take the careful path;
//   * the flusher lists all complete lines of its 64 buckets first (a ballot per round) and emits them in batches:
//     list reads, ring reads and stores each back to back -- two LDS round trips per batch instead of two per store;
//   * STOREH: the tuple leaves as {wang_hash(key), payload}; the hash is a bijection (pack6.h) and the packed last pass
//     only needs hashes, so one of the two passes' hashing disappears.
#pragma once
#include "ring_scatter.h"

#ifndef OLAP_CLAIM_NT_LOAD
#define OLAP_CLAIM_NT_LOAD 0  // A/B builds (scripts/build_variant.sh): non-temporal loads of the column tiles
#endif

namespace olap {

template <int FMAX, int BLOCK, int NPROD>
struct Claim8Cfg {
  static constexpr int kRingLog2 = 17;             // all rings together: 128 KiB (FMAX buckets x two 128-byte lines of 16 tuples)
  static_assert(FMAX * 256 == 1 << kRingLog2, "two lines per bucket at FMAX buckets");
  static constexpr int kNW = BLOCK / kWave;
  static constexpr int kNF = kNW - NPROD;
  static constexpr int kProd = NPROD * kWave;
  static_assert(FMAX <= kNF * kWave, "one bucket per flusher lane");
  static_assert(FMAX <= kProd, "the producers serve the claims, one bucket per thread");
  static constexpr int kSlots = FMAX + FMAX / 16;  // rings: the buckets' and the dummies'
  static constexpr uint32_t kCl = 0;                                   // per ring: bytes in the ring | ring head << 16
  static constexpr uint32_t kMisc = 2304;                              // [0] hot flag
  static_assert(kSlots * 4 <= kMisc, "bucket words");
  static constexpr uint32_t kHot = kMisc + 64;                         // per bucket: hot-tile hand-off {base, full | head << 16}
  static constexpr uint32_t kRsv = kHot + FMAX * 8;                    // per bucket: the allotment claimed ahead (or kRingNone)
  static constexpr uint32_t kWant = kRsv + FMAX * 4;                   // per bucket: 1 = claim one more, 2 = claim in flight
  static constexpr int kListPerWave = (FMAX / kNF) * 2;                // a wave's buckets x lines per ring: the same for every fan-out
  static constexpr uint32_t kList = kWant + FMAX * 4;                  // per flusher wave: lines to emit {line index, ring address}
  static constexpr uint32_t kRing = (kList + kNF * kListPerWave * 8 + 4095) / 4096 * 4096;
  static constexpr uint32_t kBytes = kRing + (kSlots << 8);
  static_assert(kRing < 65536, "the ring area's offset is an immediate of the LDS stores");
  static_assert(kBytes <= 160 * 1024, "LDS of one CU");
};

// Column input only (pass 0).  FULL as in ring_scatter.h (0: generic loader, 1 / 2: whole aligned tiles with / without a
// payload column).  Tiles of NPROD * 64 * TPT tuples, one workgroup per CU.
template <int FMAX, int BLOCK, int NPROD, int TPT, bool STOREH, int FULL>
__global__ __launch_bounds__(BLOCK) void k_part_claim_ring(PassSrc s, uint32_t shift, uint32_t fbits, uint2* out, ClaimOut cq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = Claim8Cfg<FMAX, BLOCK, NPROD>;
  constexpr int kNF = C::kNF, kProd = C::kProd;
  constexpr int TILE = kProd * TPT;
  constexpr uint32_t LB = 128, LT = 16;  // bytes / tuples per line
  constexpr int LPL = 8;                  // lanes per line (16 B each)
  constexpr int LPI = kWave / LPL;        // lines per store instruction
  uint32_t* cl = (uint32_t*)(smem + C::kCl);
  uint2* hot = (uint2*)(smem + C::kHot);
  uint32_t* misc = (uint32_t*)(smem + C::kMisc);
  uint32_t* rsv = (uint32_t*)(smem + C::kRsv);
  uint32_t* want = (uint32_t*)(smem + C::kWant);
  char* rings = smem + C::kRing;
  const uint32_t F = 1u << fbits;
  const uint32_t rsh = (uint32_t)C::kRingLog2 - fbits;  // log2 of a ring's bytes (>= 8)
  const uint32_t RB = 1u << rsh, M = RB - 1u;
  const uint32_t nd = F >= 16u ? F >> 4 : 1u;           // dummy rings
  TileRange r = block_tiles<SRC_COLS, TILE>(s);
  // tiles dealt round-robin in chunks of s.interleave (ring_scatter.h); tile_at(i) = i-th tile of this workgroup
  const bool il = s.interleave != 0;
  const uint64_t il_n = il ? s.interleave : 1, il_total = s.v.total_units;
  if (il) {
    const uint64_t chunks = (il_total + il_n - 1) / il_n;
    r.t0 = 0;
    r.t1 = chunks > blockIdx.x ? (chunks - blockIdx.x + gridDim.x - 1) / gridDim.x * il_n : 0;
  }
  if (r.t0 >= r.t1) return;  // nothing to read: claim nothing (uniform)
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x / kWave;
  const uint32_t cap = cq.cap;
  const uint32_t wv_shift = cq.weave_shift, wv_n = cq.weave_n;
  auto at = [&](uint64_t region, uint32_t p) { return woven_at(region, p, cap, wv_shift, wv_n); };
  // ... the same place as a line index (16 tuples; p a multiple of 16): 32-bit arithmetic, the host keeps the buffer below 2^32 lines
  auto at_line = [&](uint32_t region, uint32_t p) -> uint32_t {
    if (!wv_shift) return (uint32_t)(((uint64_t)region * cap + p) >> 4);
    return ((((p >> wv_shift) * wv_n) + region) << (wv_shift - 4u)) + ((p & ((1u << wv_shift) - 1u)) >> 4);
  };
  for (uint32_t b = threadIdx.x; b < F + nd; b += BLOCK) cl[b] = 0;
  for (uint32_t b = threadIdx.x; b < FMAX; b += BLOCK) {
    rsv[b] = kRingNone;
    want[b] = 0;
  }
  if (threadIdx.x == 0) misc[0] = 0;
  __syncthreads();

  if (wave < (unsigned)NPROD) {
    // ------------------------------------------------------------ producers
    const uint32_t dummy_bs = (F + (lane & (nd - 1u))) << rsh;  // ring offset of this lane's dummy ring
    const uint32_t lim = RB - 8u;                               // a tuple fits while the ring holds at most this many bytes
    using Regs = TileRegs<TPT, SRC_COLS, FULL>;
    // bk[i] <- ring offset of the tuple's bucket (or dummy ring); STOREH: key <- hash
    auto hash_tile = [&](Regs& t) {
#pragma unroll
      for (int i = 0; i < TPT; ++i) {
        const uint32_t h = wang_hash(t.key(i));
        const uint32_t bs = __builtin_amdgcn_ubfe(h, shift, fbits) << rsh;
        // (whole tiles have no invalid tuple; a set that was never loaded -- beyond the last tile -- has mask 0)
        t.bk[i] = (FULL ? t.mask != 0u : ((t.mask >> i) & 1u) != 0u) ? bs : dummy_bs;
        if (STOREH) t.set_key(i, h);
      }
    };
    ColsCursor cur;
    Regs A, B, Cc;
    A.mask = B.mask = Cc.mask = 0;
#pragma unroll
    for (int i = 0; i < TPT; ++i) A.bk[i] = B.bk[i] = Cc.bk[i] = dummy_bs;
    // The tiles of this workgroup in the order they are requested: the global index of the next one is stepped, not computed
    // (its closed form divides by the chunk length -- 64-bit divisions on the scalar unit that the eight producer waves share:
    // issuing a tile's four loads took 1700 cycles).  next_tile(g): false when the workgroup has no further tile.
    uint64_t seq_i = r.t0, seq_g = il ? (uint64_t)blockIdx.x * il_n : r.t0;
    uint32_t seq_j = 0;
    const uint64_t seq_jump = il ? ((uint64_t)gridDim.x - 1) * il_n + 1 : 1;
    auto next_tile = [&](uint64_t& g) -> bool {
      g = seq_g;
      const bool ok = seq_i < r.t1 && (!il || seq_g < il_total);
      ++seq_i;
      if (il && ++seq_j == (uint32_t)il_n) {
        seq_j = 0;
        seq_g += seq_jump;
      } else {
        ++seq_g;
      }
      return ok;
    };
    {
      uint64_t g;
      if (next_tile(g)) A.template load<kProd, TILE, OLAP_CLAIM_NT_LOAD != 0>(s, r, g, cur);
      if (next_tile(g)) B.template load<kProd, TILE, OLAP_CLAIM_NT_LOAD != 0>(s, r, g, cur);
    }
    hash_tile(A);
    // thread t refills the reserve of bucket t
    const uint32_t tb = threadIdx.x;
    uint32_t claimed = 0;   // the atomic's answer, in flight until the next tile's loads have been awaited
    bool pending = false;
#ifdef OLAP_EXP_PHASE_CLOCKS
    const bool clk_on = (blockIdx.x & 15) == 0 && threadIdx.x == 0;
#endif
    auto step = [&](uint64_t tile, Regs& c, Regs& nx, Regs& n2) {
      OLAP_CLK(c0);
      // insert tile `tile`: one returning LDS atomic and one 8-byte LDS store per tuple
      uint32_t v[TPT];
#pragma unroll
      for (int i = 0; i < TPT; ++i) v[i] = atomicAdd((uint32_t*)(smem + C::kCl + (c.bk[i] >> (rsh - 2u))), 8u);
      uint32_t mx = 0;
#pragma unroll
      for (int i = 0; i < TPT; ++i) mx = max(mx, v[i] & 0xFFFFu);
      uint32_t dmask = 0;
      if (!__ballot(mx > lim)) {  // every tuple of this wave fits its ring (dummy rings are reset every tile): no selects
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          const uint32_t p = (v[i] >> 16) + (v[i] & 0xFFFFu);
          *(uint2*)(rings + (c.bk[i] | (p & M))) = make_uint2(c.key(i), c.pay(i));
        }
      } else {  // a ring of this wave overflows (or a dummy ring ran over in an edge tile)
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          const uint32_t cnt = v[i] & 0xFFFFu, hd = v[i] >> 16, bs = c.bk[i];
          const bool valid = bs < (F << rsh);
          // what lies beyond the ring is placed after the flush; a dummy ring just wraps
          const uint32_t a = (!valid | (cnt < RB)) ? (bs | ((hd + cnt) & M)) : dummy_bs;
          *(uint2*)(rings + a) = make_uint2(c.key(i), c.pay(i));
          dmask |= (valid & (cnt > lim)) ? 1u << i : 0u;
        }
        if (dmask) misc[0] = (uint32_t)tile + 1u;
      }
#ifdef OLAP_EXP_PHASE_CLOCKS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      OLAP_CLK(c1);
      __syncthreads();  // ---- barrier 1: the flushers take over the rings
      OLAP_CLK(c2);
#ifdef OLAP_EXP_PHASE_CLOCKS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      OLAP_CLK(c2a);
      // tile + 1 is awaited and hashed HERE, beside the flush, and tile + 2 is requested only after tile + 1 has arrived
      // (ring_scatter.h)
      nx.pin();
      hash_tile(nx);
#pragma unroll
      for (int i = 0; i < TPT; ++i) asm volatile("" : "+v"(nx.bk[i]) : : "memory");
      OLAP_CLK(c2b);
      {  // (every VMEM operation of this wave has completed at this point)
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
      {
        uint64_t g;
        if (next_tile(g)) n2.template load<kProd, TILE, OLAP_CLAIM_NT_LOAD != 0>(s, r, g, cur);
      }
      OLAP_CLK(c3);
      __syncthreads();  // ---- barrier 2: complete lines are out, the rings have room again
      OLAP_CLK(c4);
      OLAP_CLK_ADD(0, 0, c1 - c0);
      OLAP_CLK_ADD(0, 1, c2 - c1);
      OLAP_CLK_ADD(0, 2, c3 - c2);
      OLAP_CLK_ADD(0, 3, c4 - c3);
      OLAP_CLK_ADD(0, 8, c2a - c2);
      OLAP_CLK_ADD(0, 9, c2b - c2a);
      OLAP_CLK_ADD(0, 10, c2c - c2b);
      OLAP_CLK_ADD(0, 11, c3 - c2c);
      OLAP_CLK_ADD(0, 7, 1);
      if (misc[0] == (uint32_t)tile + 1u) {  // some bucket was hot in this tile (uniform)
        OLAP_CLK_ADD(0, 12, 1);
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          if (dmask & (1u << i)) {
            const uint32_t bs = c.bk[i], b = bs >> rsh;
            const uint2 info = hot[b];
            const uint32_t full = info.y & 0xFFFFu, hd = info.y >> 16, cnt = v[i] & 0xFFFFu;
            if (cnt < full) {  // goes straight to the range the owner reserved (dropped when the region is full)
              if (info.x != kRingNone) gstore2(out + at((uint64_t)r.seg * F + b, info.x + (cnt - RB) / 8u), make_uint2(c.key(i), c.pay(i)));
            } else {           // tail of the tile: the bucket's new residue
              *(uint2*)(rings + (bs | ((hd + (cnt - full)) & M))) = make_uint2(c.key(i), c.pay(i));
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
    if (pending) rsv[tb] = claimed;  // delivered before the owners close their regions
    __syncthreads();
  } else {
    // ------------------------------------------------------------- flushers
    const unsigned fw = wave - NPROD;
    const uint32_t bpw = F >= (uint32_t)kNF ? F / kNF : 1u;  // buckets per flusher wave: consecutive buckets, consecutive lanes
    const uint32_t b_own = fw * bpw + lane;
    const bool owner = lane < bpw && b_own < F;
    const uint64_t p_own = (uint64_t)r.seg * F + b_own;
    const uint32_t ring_own = C::kRing + (b_own << rsh);     // LDS address of the bucket's ring
    uint2* list = (uint2*)(smem + C::kList) + fw * C::kListPerWave;
    uint32_t pos = 0, rem = 0;                   // next line's tuple offset in the region; tuples left in the allotment
    uint32_t fill_lo = 0, fill_hi = 0;           // unused tail of a hot tile's block
    bool dead = !owner;                          // region exhausted (skew): tuples are dropped, *overflow raised
    uint32_t filler = 0;
    if (owner) {
      while (__builtin_amdgcn_ubfe(wang_hash(filler), shift, fbits) == b_own) ++filler;  // a key of another bucket
      if (STOREH) filler = wang_hash(filler);
      const uint32_t got = atomicAdd(&cq.cursor[p_own], 2 * kClaimChunk);
      if ((uint64_t)got + 2 * kClaimChunk > cap) {
        dead = true;
        atomicAdd(cq.overflow, 1ull);
      } else {
        pos = got;
        rem = kClaimChunk;
        rsv[b_own] = got + kClaimChunk;
      }
    }
    // the next allotment -- the reserve (a producer refills it), or claimed on the spot if that has not arrived
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
        atomicAdd(cq.overflow, 1ull);
      } else {
        pos = a;
        rem = kClaimChunk;
      }
    };
    // The wave emits the listed lines eight per store instruction, kFU instructions' worth at a time -- list entries, then
    // ring reads, then stores, each kind back to back, so that a batch costs two LDS round trips instead of two per instruction.
    constexpr int kFU = 4;
    auto emit_lines = [&](uint32_t n) {
      for (uint32_t k0 = 0; k0 < n; k0 += LPI * kFU) {
        uint2 e[kFU];
        uint4 t[kFU];
#pragma unroll
        for (int u = 0; u < kFU; ++u) {
          const uint32_t j = k0 + u * LPI + lane / LPL;
          e[u] = list[j < n ? j : 0];
        }
#pragma unroll
        for (int u = 0; u < kFU; ++u) t[u] = *(const uint4*)(smem + e[u].y + 16 * (lane % LPL));
#pragma unroll
        for (int u = 0; u < kFU; ++u) {
          const uint32_t j = k0 + u * LPI + lane / LPL;
          if (j < n) gstore4((uint4*)out + (uint64_t)e[u].x * LPL + (lane % LPL), t[u]);
        }
      }
    };
#ifdef OLAP_EXP_PHASE_CLOCKS
    const bool clk_on = (blockIdx.x & 15) == 0 && threadIdx.x == NPROD * kWave;
#endif
    for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
      OLAP_CLK(f0);
      __syncthreads();  // ---- barrier 1
      OLAP_CLK(f1);
      const uint32_t vv = owner ? cl[b_own] : 0u;
      if (fw == 0 && lane < nd) cl[F + lane] = 0;  // the dummy rings start every tile empty
      const uint32_t total = vv & 0xFFFFu, hd = vv >> 16;
      const bool hotb = total > RB;
      const uint32_t placed = hotb ? RB : total;
      const uint32_t nl = placed / LB;  // complete lines in the ring
      // list them in rounds (round q: line q of the owners that have one), every line at the next 16 tuples of its
      // bucket's allotment; then emit
      uint32_t n = 0;
      for (uint32_t q = 0;; ++q) {
        bool has = q < nl && !dead;
        if (has) {
          if (rem == 0) next_allotment();
          has = !dead;
        }
        const unsigned long long m = __ballot(has);
        if (!m) {
          if (!__ballot(q + 1 < nl)) break;
          continue;
        }
        if (has) {
          list[n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
              make_uint2(at_line((uint32_t)p_own, pos), ring_own + ((hd + q * LB) & M));
          pos += LT;
          rem -= LT;
        }
        n += (uint32_t)__popcll(m);
      }
      // an allotment that has just run out is replaced NOW from the reserve, beside the emission (one LDS round trip per tile for
      // the few lanes concerned), so that the listing above rarely has to wait for one; no reserve yet: it is asked for again when needed
      if (owner && !dead && rem == 0) {
        const uint32_t a = rsv[b_own];
        if (a != kRingNone) {
          rsv[b_own] = kRingNone;
          want[b_own] = 1u;
          if ((uint64_t)a + kClaimChunk > cap) {
            dead = true;
            atomicAdd(cq.overflow, 1ull);
          } else {
            pos = a;
            rem = kClaimChunk;
          }
        }
      }
      OLAP_CLK(fa);
      __builtin_amdgcn_wave_barrier();
      emit_lines(n);
      __builtin_amdgcn_wave_barrier();
      OLAP_CLK(fb);
      OLAP_CLK_ADD(0, 13, fa - f1);
      OLAP_CLK_ADD(0, 14, fb - fa);
      OLAP_CLK_ADD(0, 15, n);
      if (owner) {
        if (hotb) {  // the tuples beyond the ring: whole lines' worth goes out directly, the rest is the new residue
          const uint32_t full = total & ~(LB - 1);
          const uint32_t ndt = (full - RB) / 8u;  // tuples that go out directly (a multiple of 16)
          uint32_t dbase = kRingNone;
          if (ndt && !dead) {
            // Normally out of the allotment in use, or -- what is left of that one padded with fillers -- out of the reserve:
            // no atomic to wait for (some bucket is hot in most tiles at 512 buckets)
            if (ndt <= kClaimChunk) {
              if (rem < ndt) {
                fill_lo = pos;
                fill_hi = pos + rem;
                next_allotment();
              }
              if (!dead) {
                dbase = pos;
                pos += ndt;
                rem -= ndt;
              }
            } else {  // a very hot bucket: its own block
              const uint32_t blk = (ndt + kClaimChunk - 1) / kClaimChunk * kClaimChunk;
              const uint32_t got = atomicAdd(&cq.cursor[p_own], blk);
              if ((uint64_t)got + blk > cap) {
                dead = true;
                atomicAdd(cq.overflow, 1ull);
              } else {
                dbase = got;
                fill_lo = got + ndt;
                fill_hi = got + blk;
              }
            }
          }
          hot[b_own] = make_uint2(dbase, full | (hd << 16));
          cl[b_own] = (total - full) | (hd << 16);
        } else {
          cl[b_own] = (placed - nl * LB) | (((hd + nl * LB) & M) << 16);
        }
      }
      OLAP_CLK(f2);
      __syncthreads();  // ---- barrier 2
      OLAP_CLK(f3);
      OLAP_CLK_ADD(0, 4, f1 - f0);
      OLAP_CLK_ADD(0, 5, f2 - f1);
      OLAP_CLK_ADD(0, 6, f3 - f2);
      if (misc[0] == (uint32_t)tile + 1u) {
        for (uint32_t q = fill_lo; q < fill_hi; q += 2) *(uint4*)(out + at(p_own, q)) = make_uint4(filler, 0u, filler, 0u);
        fill_lo = fill_hi = 0;
        __syncthreads();  // ---- barrier 3 (hot tiles only)
      }
    }
    __syncthreads();  // the producers have delivered the claims that were in flight
    // tail: what is left in the rings (less than a line per bucket)
    if (owner) {
      const uint32_t vv = cl[b_own];
      const uint32_t rr = (vv & 0xFFFFu) / 8u, hd = vv >> 16;  // tuples
      if (!dead && rr) {
        if (rem == 0) next_allotment();
        if (!dead) {  // the residue as one line padded with fillers
          uint2* line = out + at(p_own, pos);
          for (uint32_t j = 0; j < LT; ++j)
            line[j] = j < rr ? *(const uint2*)(smem + ring_own + ((hd + 8u * j) & M)) : make_uint2(filler, 0u);
          pos += LT;
          rem -= LT;
        }
      }
      if (!dead) {  // the rest of the allotment, and the one claimed ahead, hold only fillers
        for (uint32_t q = 0; q < rem; q += 2) *(uint4*)(out + at(p_own, pos + q)) = make_uint4(filler, 0u, filler, 0u);
      }
      const uint32_t ahead = rsv[b_own];
      if (ahead != kRingNone && (uint64_t)ahead + kClaimChunk <= cap)
        for (uint32_t q = 0; q < kClaimChunk; q += 2) *(uint4*)(out + at(p_own, ahead + q)) = make_uint4(filler, 0u, filler, 0u);
    }
  }
}

}  // namespace olap
