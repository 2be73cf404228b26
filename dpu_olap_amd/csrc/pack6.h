// pack6.h -- 6-byte tuples between the last radix pass and the probe (included by join.hip only).
//
// Reference behaviour: the same partition / build / probe pipeline as ring_scatter.h and probe_kernels.h
// (dpu/shared/kernels/partition.c:167-294, hash_build.c:9-35, hash_probe.c:9-46, hashtable.c:89-192); what changes is
// the width of the tuples the last pass hands to the probe.
//
// Every step of the Wang hash is invertible (probe_kernels.h), so a key IS its hash.  All tuples of a final partition
// share the hash bits above the table bits; the direct-address table is indexed by the bits below.  A tuple of a final
// partition therefore needs only {payload, low 16 hash bits} = 6 bytes instead of {key, payload} = 8: the probe
// rebuilds fk = wang_hash_inv(top(p) << table_bits | low bits), where top(p) is recorded once per partition by the pass
// that wrote it.  At 2^32 x 2^32 rows this takes 2 x 8.6 GB off the last pass's stores and 17 GB off the probe's loads
// (92 -> 84 B per row pair).
//
// Layout: partition p owns the byte range [p * cap * 6, (p + 1) * cap * 6) (cap a multiple of 64 tuples, so every
// region starts on a 128-byte line); tuple j lies at byte 6 j: payload (4 bytes, little endian), then hash & 0xFFFF.
// Tuples straddle 16-byte vectors and 128-byte lines, so both kernels work on the region as a stream of HALFWORDS:
//   * k_part_scatter_ring6: the ring scatter of ring_scatter.h with per-bucket rings of 128 halfwords (two lines); a tuple
//     is a 4-byte and a 2-byte LDS store behind one returning LDS atomic (+3 on the bucket's halfword count); complete lines
//     leave as 8 lanes x 16 B exactly as before.  42 tuples fit a ring instead of 32, so far fewer tiles take the
//     hot-bucket detour.
//   * k_join_probe_direct6: every wave copies whole 16-byte vectors of its chunk (coalesced) into a private LDS
//     staging strip and reads them back as pairs of tuples (12 bytes = 3 dwords per lane, conflict free), so loads
//     AND result stores stay lane-contiguous.
#pragma once
#include "probe_kernels.h"
#include "ring_scatter.h"

namespace olap {

// ---- inverse of wang_hash (common.h; partition.c:20-28): the steps undone last to first
constexpr uint32_t inv_odd_u32(uint32_t a) {  // multiplicative inverse modulo 2^32 of an odd number (Newton)
  uint32_t x = a;
  for (int i = 0; i < 5; ++i) x *= 2u - a * x;
  return x;
}
constexpr uint32_t kWangInv11 = inv_odd_u32(1u - (1u << 11));  // key += ~(key << 11)  ==  key * (1 - 2^11) - 1
constexpr uint32_t kWangInv9 = inv_odd_u32(9u);                // key += key << 3      ==  key * 9
constexpr uint32_t kWangInv15 = inv_odd_u32(1u - (1u << 15));  // key += ~(key << 15)  ==  key * (1 - 2^15) - 1
__host__ __device__ __forceinline__ uint32_t wang_hash_inv(uint32_t k) {
  k ^= k >> 16;
  k = (k + 1u) * kWangInv11;
  k ^= (k >> 6) ^ (k >> 12) ^ (k >> 18) ^ (k >> 24) ^ (k >> 30);
  k *= kWangInv9;
  k ^= (k >> 10) ^ (k >> 20) ^ (k >> 30);
  k = (k + 1u) * kWangInv15;
  return k;
}

constexpr uint32_t kPackAlign = 64;  // tuples: packed regions are whole 128-byte lines (64 x 6 B = 3 lines)

// ------------------------------------------------------------------ last radix pass, 6-byte output
template <int FMAX, int BLOCK, int NPROD>
struct Ring6Cfg {
  static constexpr int kHwMin = 128;               // halfwords per bucket ring at FMAX buckets (two 128-byte lines); F buckets: FMAX * 128 / F
  static constexpr int kNW = BLOCK / kWave;
  static constexpr int kNF = kNW - NPROD;
  static constexpr int kProd = NPROD * kWave;
  static_assert(FMAX <= kNF * kWave, "one bucket per flusher lane");
  static_assert(FMAX <= BLOCK, "the closing scan runs one bucket per thread");
  static constexpr int kDummy = 32;                // invalid tuples go to dummy counters / slots: no branches
  static constexpr uint32_t kRingBytes = FMAX * kHwMin * 2;
  static constexpr uint32_t kDummyOff = kRingBytes;                    // kDummy lanes x 4 halfwords
  static constexpr uint32_t kCl = kDummyOff + kDummy * 8;              // per bucket: halfwords in the ring | ring head << 16
  static constexpr uint32_t kHot = kCl + (FMAX + kDummy) * 4;          // per bucket: hot-tile hand-off {base, full | head << 16}
  static constexpr int kListPerWave = (FMAX / kNF) * (kHwMin / 64);    // a wave's buckets x lines per ring: the same for every fan-out
  static constexpr uint32_t kList = kHot + FMAX * 8;                   // per flusher wave: lines to emit {line index, ring byte offset}
  static constexpr uint32_t kMisc = kList + kNF * kListPerWave * 8;    // [0] hot flag, [1] the segment's leading hash bits, scan scratch
  static constexpr uint32_t kBytes = kMisc + kNW * 8 + 64;
};

// Capacity layout only (one workgroup per input segment, CapOut), tuple input only.  out: the packed regions as halfwords.
// ptop[seg * F + b] = the hash bits above `shift` that every tuple of partition (seg, b) carries; *foreign counts tuples
// of a segment that disagree about the bits above the pass's field (a violated olap_join_set_hash_skip_bits promise).
template <int FMAX, int BLOCK, int NPROD, int TPT>
__global__ __launch_bounds__(BLOCK) void k_part_scatter_ring6(PassSrc s, uint32_t shift, uint32_t fbits, uint16_t* out, CapOut co,
                                                                uint32_t* ptop, unsigned long long* foreign) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = Ring6Cfg<FMAX, BLOCK, NPROD>;
  constexpr int kNF = C::kNF, kProd = C::kProd;
  constexpr int TILE = kProd * TPT;
  constexpr uint32_t LHW = 64;        // halfwords per 128-byte line
  constexpr int LPL = 8;              // lanes per line (16 B each)
  constexpr int LPI = kWave / LPL;    // lines per store instruction
  uint16_t* ring = (uint16_t*)smem;
  uint32_t* cl = (uint32_t*)(smem + C::kCl);
  uint2* hot = (uint2*)(smem + C::kHot);
  uint32_t* misc = (uint32_t*)(smem + C::kMisc);
  const uint32_t F = 1u << fbits;
  const uint32_t RB = (uint32_t)(FMAX * C::kHwMin) >> fbits;  // halfwords per bucket ring (a power of two >= 128)
  const TileRange r = block_tiles<SRC_TUPLES, TILE>(s);
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x / kWave;
  const uint32_t cap_hw = co.cap * 3u;  // halfwords per region (the host keeps it below 2^32)
  for (uint32_t b = threadIdx.x; b < FMAX + C::kDummy; b += BLOCK) cl[b] = 0;
  if (threadIdx.x == 0) {
    misc[0] = 0;
    misc[1] = kRingNone;
  }
  __syncthreads();
  uint32_t mytop = kRingNone, bad = 0;  // leading hash bits of the first tuple this thread saw; tuples that differ from it

  if (wave < (unsigned)NPROD) {
    // ------------------------------------------------------------ producers
    const uint32_t expect = r.seg & s.chk_mask;
    const uint32_t dummy = FMAX + (lane & (C::kDummy - 1));
    const uint32_t dummy_hw = C::kDummyOff / 2 + (lane & (C::kDummy - 1)) * 4;
    const uint32_t top_shift = shift + fbits;  // < 32 (host)
    using Regs = TileRegs<TPT, SRC_TUPLES, 0>;
    // bk[i] = bucket << 16 | low 16 hash bits (the key itself is not needed after this)
    auto hash_tile = [&](Regs& t) {
#pragma unroll
      for (int i = 0; i < TPT; ++i) {
        const uint32_t h = wang_hash(t.k[i]);
        const bool ok = (((t.mask >> i) & 1u) != 0) & (((h >> s.chk_shift) & s.chk_mask) == expect);
        t.bk[i] = ((ok ? ((h >> shift) & (F - 1u)) : dummy) << 16) | (h & 0xFFFFu);
        const uint32_t tp = h >> top_shift;
        mytop = (ok & (mytop == kRingNone)) ? tp : mytop;
        bad += (ok & (tp != mytop)) ? 1u : 0u;
      }
    };
    ColsCursor cur;
    Regs A, B, Cc;
    A.mask = B.mask = Cc.mask = 0;
#pragma unroll
    for (int i = 0; i < TPT; ++i) A.bk[i] = B.bk[i] = Cc.bk[i] = dummy << 16;
    if (r.t0 < r.t1) A.template load<kProd, TILE, false>(s, r, r.t0, cur);
    if (r.t0 + 1 < r.t1) B.template load<kProd, TILE, false>(s, r, r.t0 + 1, cur);
    hash_tile(A);
    auto hw_of = [](uint32_t pay, uint32_t bkh, int j) -> uint16_t { return j == 0 ? (uint16_t)pay : j == 1 ? (uint16_t)(pay >> 16) : (uint16_t)bkh; };
    auto step = [&](uint64_t tile, Regs& c, Regs& nx, Regs& n2) {
      // insert tile `tile`: one returning LDS atomic, one 4-byte and one 2-byte LDS store per tuple
      uint32_t v[TPT];
#pragma unroll
      for (int i = 0; i < TPT; ++i) v[i] = atomicAdd(&cl[c.bk[i] >> 16], 3u);
      uint32_t dmask = 0;
#pragma unroll
      for (int i = 0; i < TPT; ++i) {
        const uint32_t b = c.bk[i] >> 16, cnt = v[i] & 0xFFFFu, hd = v[i] >> 16;
        const bool valid = b < (uint32_t)FMAX;
        // The tuple's 6 bytes start on a halfword: an aligned dword and a halfword -- {payload | hash} from an even start,
        // {payload low | payload high : hash} from an odd one (the ring head is a multiple of 64, so the count's parity is the
        // start's, and the dword fits the ring as a whole or not at all).  The pieces may lie on both sides of the ring's wrap.
        const bool odd = (cnt & 1u) != 0;
        const uint32_t pay = c.p[i], h16 = c.bk[i] & 0xFFFFu;
        const uint32_t q32 = cnt + (odd ? 1u : 0u), q16 = cnt + (odd ? 0u : 2u);
        const uint32_t v32 = odd ? ((pay >> 16) | (h16 << 16)) : pay;
        const uint32_t v16 = odd ? pay : h16;
        const uint32_t a32 = (valid & (q32 < RB)) ? b * RB + ((hd + q32) & (RB - 1)) : dummy_hw;
        const uint32_t a16 = (valid & (q16 < RB)) ? b * RB + ((hd + q16) & (RB - 1)) : dummy_hw + 2;
        *(uint32_t*)(ring + a32) = v32;
        ring[a16] = (uint16_t)v16;
        dmask |= (valid & (cnt + 3u > RB)) ? 1u << i : 0u;  // (part of) this tuple is placed after the flush
      }
      if (dmask) misc[0] = (uint32_t)tile + 1u;
      __syncthreads();  // ---- barrier 1: the flushers take over the rings
      nx.pin();
      hash_tile(nx);
#pragma unroll
      for (int i = 0; i < TPT; ++i) asm volatile("" : "+v"(nx.bk[i]) : : "memory");
      n2.mask = 0;
      if (tile + 2 < r.t1) n2.template load<kProd, TILE, false>(s, r, tile + 2, cur);
      __syncthreads();  // ---- barrier 2: complete lines are out, the rings have room again
      if (misc[0] == (uint32_t)tile + 1u) {  // some bucket was hot in this tile (uniform)
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          if (dmask & (1u << i)) {
            const uint32_t b = c.bk[i] >> 16;
            const uint2 info = hot[b];
            const uint32_t full = info.y & 0xFFFFu, hd = info.y >> 16, cnt = v[i] & 0xFFFFu;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const uint32_t q = cnt + j;
              if (q < RB) continue;  // already in the ring (flushed with it)
              const uint16_t val = hw_of(c.p[i], c.bk[i], j);
              if (q < full) {  // straight to the range the owner reserved (dropped when the region is full)
                if (info.x != kRingNone) gstore_t(out + ((uint64_t)r.seg * F + b) * cap_hw + info.x + (q - RB), val);
              } else {         // tail of the tile: the bucket's new residue
                ring[b * RB + ((hd + (q - full)) & (RB - 1))] = val;
              }
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
  } else {
    // ------------------------------------------------------------- flushers
    const unsigned fw = wave - NPROD;
    const uint32_t bpw = F >= (uint32_t)kNF ? F / kNF : 1u;  // buckets per flusher wave: consecutive buckets, consecutive lanes
    const uint32_t b_own = fw * bpw + lane;
    const bool owner = lane < bpw && b_own < F;
    const uint64_t p_own = (uint64_t)r.seg * F + b_own;
    uint2* list = (uint2*)(smem + C::kList) + fw * C::kListPerWave;
    uint32_t pos = 0;     // next line's halfword offset in the region
    bool dead = !owner;   // region exhausted (skew): tuples are dropped, *overflow raised
    // The flush runs ONE TILE BEHIND the insert: the lines that are complete after tile t are listed between the two barriers
    // of tile t and leave (LDS -> HBM) while the producers insert tile t + 1; they stay in the ring until the bookkeeping of
    // tile t + 1 releases them.  `pend`: this owner's lines listed for the coming emission; n_list: the wave's.
    uint32_t pend = 0, n_list = 0;
    const uint32_t lines_per_ring = RB / LHW;
    uint32_t nplanes = 1;
    while ((1u << nplanes) <= lines_per_ring) ++nplanes;  // bit planes of a per-owner line count
    // exclusive prefix of x over the wave's lanes (+ the wave's total) from ballots of x's bit planes
    auto wave_rank = [&](uint32_t x, uint32_t& total) -> uint32_t {
      const unsigned long long below = (1ull << lane) - 1ull;
      uint32_t rank = 0;
      total = 0;
      for (uint32_t bit = 0; bit < nplanes; ++bit) {
        const unsigned long long m = __ballot((x >> bit) & 1u);
        rank += (uint32_t)__popcll(m & below) << bit;
        total += (uint32_t)__popcll(m) << bit;
      }
      return rank;
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
    // lists `count` lines of this owner: ring lines from halfword `ring0` on, output from `pos` on
    auto list_lines = [&](uint32_t rank, uint32_t count, uint32_t ring0) {
      for (uint32_t q = 0; q < count; ++q)
        list[rank + q] = make_uint2((uint32_t)((p_own * cap_hw + pos + q * LHW) / LHW), (b_own * RB + ((ring0 + q * LHW) & (RB - 1))) * 2u);
      pos += count * LHW;
    };
#ifdef OLAP_EXP_PHASE_CLOCKS
    const bool clk_on = (blockIdx.x & 15) == 0 && threadIdx.x == NPROD * kWave;
#endif
    for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
      OLAP_CLK(f0);
      emit_lines(n_list);  // the lines completed up to the previous tile, beside the insert of this one
      OLAP_CLK(f0a);
      __syncthreads();  // ---- barrier 1: the inserts of this tile are in the rings
      OLAP_CLK(f1);
      const uint32_t vv = owner ? cl[b_own] : 0u;
      const uint32_t tot0 = vv & 0xFFFFu, hd0 = vv >> 16;
      const bool hotb = tot0 > RB;  // (the window of this tile's insert: RB halfwords from hd0, the listed lines included)
      if (__ballot(hotb)) {
        // A hot bucket's ring is exactly full; its first `pend` lines have just left.  The rest leaves NOW, so that the
        // ring is empty for what did not fit; whole lines' worth of that goes out directly, the rest is the new residue.
        uint32_t imm = 0;
        if (hotb && !dead) {
          imm = lines_per_ring - pend;
          if ((uint64_t)pos + imm * LHW > cap_hw) {
            dead = true;
            atomicAdd(co.overflow, 1ull);
            imm = 0;
          }
        }
        uint32_t n_imm;
        const uint32_t rank = wave_rank(imm, n_imm);
        list_lines(rank, imm, hd0 + pend * LHW);
        __builtin_amdgcn_wave_barrier();
        emit_lines(n_imm);
        __builtin_amdgcn_wave_barrier();
        if (hotb) {
          const uint32_t full = tot0 & ~(LHW - 1);
          const uint32_t nd = full - RB;
          uint32_t dbase = kRingNone;
          if (nd && !dead) {
            if ((uint64_t)pos + nd > cap_hw) {
              dead = true;
              atomicAdd(co.overflow, 1ull);
            } else {
              dbase = pos;
              pos += nd;
            }
          }
          hot[b_own] = make_uint2(dbase, full | (hd0 << 16));
          cl[b_own] = (tot0 - full) | (hd0 << 16);
          pend = 0;
        }
      }
      uint32_t mine = 0, hd1 = 0;
      if (owner && !hotb) {
        const uint32_t cnt1 = tot0 - pend * LHW;  // release the lines that went out during the insert
        hd1 = (hd0 + pend * LHW) & (RB - 1);
        const uint32_t nl = cnt1 / LHW;           // complete lines now: they go out during the next insert
        if (nl && !dead) {
          if ((uint64_t)pos + nl * LHW > cap_hw) {
            dead = true;
            atomicAdd(co.overflow, 1ull);
          } else {
            mine = nl;
          }
        }
        if (dead) cl[b_own] = (cnt1 - nl * LHW) | (((hd1 + nl * LHW) & (RB - 1)) << 16);  // dropped at once
        else cl[b_own] = cnt1 | (hd1 << 16);
        pend = mine;
      }
      const uint32_t rank = wave_rank(mine, n_list);
      list_lines(rank, mine, hd1);
      OLAP_CLK(f2);
      __syncthreads();  // ---- barrier 2: the rings' counts are up to date
      OLAP_CLK(f3);
      OLAP_CLK_ADD(1, 4, f1 - f0a);
      OLAP_CLK_ADD(1, 5, f2 - f1);
      OLAP_CLK_ADD(1, 6, f3 - f2);
      OLAP_CLK_ADD(1, 10, f0a - f0);
      OLAP_CLK_ADD(1, 11, n_list);
      OLAP_CLK_ADD(1, 7, 1);
      if (misc[0] == (uint32_t)tile + 1u) __syncthreads();  // ---- barrier 3 (hot tiles only)
    }
    emit_lines(n_list);  // what the last tile completed
    // tail: what is left in the rings (less than a line per bucket)
    if (owner) {
      const uint32_t vv = cl[b_own];
      const uint32_t rr = (vv & 0xFFFFu) - pend * LHW, hd = ((vv >> 16) + pend * LHW) & (RB - 1);
      uint16_t* region = out + p_own * cap_hw;
      uint32_t count_hw = pos;
      if (!dead) {
        if ((uint64_t)pos + rr > cap_hw) {
          atomicAdd(co.overflow, 1ull);
        } else {
          for (uint32_t j = 0; j < rr; ++j) region[pos + j] = ring[b_own * RB + ((hd + j) & (RB - 1))];
          count_hw = pos + rr;
        }
      }
      ((uint32_t*)hot)[b_own] = count_hw / 3u;  // tuples, for the closing scan
    }
  }
  // partition sizes, dense starts and leading hash bits (exclusive scan over this segment's buckets, one per thread)
  if (mytop != kRingNone) misc[1] = mytop;  // (any of the segment's tuples: they all agree, or `foreign` says so)
  __syncthreads();
  const uint32_t segtop = misc[1];
  if (mytop != kRingNone && mytop != segtop) ++bad;
  if (bad) atomicAdd(foreign, (unsigned long long)bad);
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
    ptop[(uint64_t)r.seg * F + t] = segtop == kRingNone ? 0u : ((segtop << fbits) | t);
  }
  if (co.seg_total && threadIdx.x == 0) {
    uint64_t total = 0;
    for (int w = 0; w < BLOCK / kWave; ++w) total += wtot[w];
    co.seg_total[r.seg] = total;
  }
}

// ------------------------------------------------------------------ build + probe over packed partitions
// A wave walks the chunks (CHUNK tuples = CHUNK * 6 bytes, a whole number of 16-byte vectors) of a packed region that
// fall to it: vectors in with coalesced 16-byte loads (the next chunk in flight meanwhile), through its private staging
// strip, out as PAIRS of tuples -- lane l takes pairs l, l + 64, ... : 12 bytes at dword 3 * pair, so the 64 lanes of a
// read hit 64 different banks and the pair {2m, 2m + 1} needs no byte shuffling beyond one funnel shift.
// f(i, pay0, h0, pay1, h1): tuples i (even) and i + 1 of the region.
template <int CHUNK, class Fn>
__device__ __forceinline__ void scan_packed(const char* region, uint32_t lo, uint32_t hi, char* stage, unsigned wave, unsigned nwaves,
                                            unsigned lane, Fn&& f) {
  static_assert(CHUNK % 128 == 0, "whole pairs per lane");
  constexpr int NP = CHUNK * 6 / 16;       // 16-byte vectors per chunk
  constexpr int LPL = (NP + kWave - 1) / kWave;
  const uint32_t c1 = (hi + CHUNK - 1) / CHUNK;
  uint32_t c = lo / CHUNK + wave;
  uint4 cur[LPL], nxt[LPL];
#pragma unroll
  for (int k = 0; k < LPL; ++k) cur[k] = nxt[k] = make_uint4(0, 0, 0, 0);
  auto load = [&](uint32_t cc, uint4 (&v)[LPL]) {
#pragma unroll
    for (int k = 0; k < LPL; ++k) {
      const uint32_t piece = k * kWave + lane;
      if (NP % kWave == 0 || piece < (uint32_t)NP) v[k] = gload4(region + ((uint64_t)cc * NP + piece) * 16);
    }
  };
  if (c < c1) load(c, cur);
  for (; c < c1; c += nwaves) {
    if (c + nwaves < c1) load(c + nwaves, nxt);
#pragma unroll
    for (int k = 0; k < LPL; ++k) {
      const uint32_t piece = k * kWave + lane;
      if (NP % kWave == 0 || piece < (uint32_t)NP) *(uint4*)(stage + piece * 16) = cur[k];
    }
#pragma unroll
    for (int j = 0; j < CHUNK / 128; ++j) {
      const uint32_t m = j * kWave + lane;  // pair index inside the chunk
      const uint32_t* d = (const uint32_t*)(stage + 12 * m);
      const uint32_t d0 = d[0], d1 = d[1], d2 = d[2];
      f(c * CHUNK + 2 * m, d0, d1 & 0xFFFFu, (d1 >> 16) | (d2 << 16), d2 >> 16);
    }
#pragma unroll
    for (int k = 0; k < LPL; ++k) cur[k] = nxt[k];
  }
}

// k_join_probe_direct (probe_kernels.h) over partitions whose right (RPK) and / or left (LPK) side is packed.  A side
// that is not packed is an 8-byte-tuple side in the capacity or the exact layout, read as there (what an overflowing
// side is redone as).  V.top[p] of a packed side = the hash bits above the table bits of partition p.
template <int BLOCK, int CHUNK, bool RPK, bool LPK>
__global__ __launch_bounds__(BLOCK) void k_join_probe_direct6(PartView RV, PartView LV, uint32_t nparts, uint32_t table_bits,
                                                                uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x,
                                                                uint32_t pair_store_ok, unsigned long long* counters, uint32_t* miss_bits,
                                                                ProbeSlices slices) {
  static_assert(RPK || LPK, "use k_join_probe_direct for two unpacked sides");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* s_vals = (uint32_t*)smem;
  uint32_t* s_used = (uint32_t*)(smem + kDSlots * 4);
  const unsigned lane = lane_id(), wave = threadIdx.x / kWave;
  constexpr unsigned kNW = BLOCK / kWave;
  char* stage = smem + kDSlots * 4 + kDSlots / 8 + wave * (CHUNK * 6);
  const uint2* R = RV.t;
  const uint2* L = LV.t;
  const uint32_t slots = 1u << table_bits, smask = slots - 1;
  uint32_t my_miss = 0, my_dup = 0, my_foreign = 0;
  uint64_t my_max = 0;
  auto body = [&](const uint32_t p, const uint32_t slice) {
    const uint64_t n_r = RV.cap ? (uint64_t)RV.cnt[p] : RV.off[p + 1] - RV.off[p];
    const uint64_t n_l = LV.cap ? (uint64_t)LV.cnt[p] : LV.off[p + 1] - LV.off[p];
    if (n_r > my_max) my_max = n_r;
    if (n_l == 0) return;
    const uint64_t lb = (uint64_t)slice * kProbeSlice, le = slices.count ? min(n_l, lb + kProbeSlice) : n_l;  // partition-relative
    if (lb >= le) return;
    // the hash bits above the table bits: from the packed side(s); an unpacked side's tuples are checked against them
    const uint32_t top = (LPK ? LV.top[p] : RV.top[p]);
    if (RPK && LPK && n_r && RV.top[p] != top && threadIdx.x == 0) my_foreign += (uint32_t)min(n_l, (uint64_t)0xFFFFFFFFu);
    for (uint32_t i = threadIdx.x; i < (slots + 31) / 32; i += BLOCK) s_used[i] = 0;
    __syncthreads();
    if (RPK) {
      scan_packed<CHUNK>((const char*)R + (uint64_t)p * RV.cap * 6, 0u, (uint32_t)n_r, stage, wave, kNW, lane,
                         [&](uint32_t i, uint32_t p0, uint32_t h0, uint32_t p1, uint32_t h1) {
#pragma unroll
                           for (int h = 0; h < 2; ++h) {
                             if (i + h >= n_r) continue;
                             const uint32_t slot = (h ? h1 : h0) & smask;
                             const uint32_t bit = 1u << (slot & 31);
                             if ((atomicOr(&s_used[slot >> 5], bit) & bit) && slice == 0) ++my_dup;  // equal key overwrites (hashtable.c:133)
                             s_vals[slot] = h ? p1 : p0;
                           }
                         });
    } else {
      const uint64_t rlo = RV.cap ? (uint64_t)p * RV.cap : RV.off[p], rhi = rlo + n_r;
      const uint64_t a = rlo & ~1ull;
      const uint64_t span = rhi - a;
      for (uint64_t base = 0; base < span; base += BLOCK * 2 * kJU) {
        uint4 v[kJU];
#pragma unroll
        for (int u = 0; u < kJU; ++u) {
          const uint64_t i0 = base + (uint64_t)(u * BLOCK + threadIdx.x) * 2;
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
          const uint64_t i0 = a + base + (uint64_t)(u * BLOCK + threadIdx.x) * 2;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t k = h ? v[u].z : v[u].x, val = h ? v[u].w : v[u].y;
            if (i0 + h < rlo || i0 + h >= rhi) continue;
            const uint32_t hk = wang_hash(k);
            if ((hk >> table_bits) != top) ++my_foreign;
            const uint32_t slot = hk & smask;
            const uint32_t bit = 1u << (slot & 31);
            if ((atomicOr(&s_used[slot >> 5], bit) & bit) && slice == 0) ++my_dup;
            s_vals[slot] = val;
          }
        }
      }
    }
    __syncthreads();
    if (LPK) {
      const uint64_t dense = LV.dense[p];
      // 8-byte pair stores need an 8-byte aligned element (tuple indices here are even): as k_join_probe_direct
      const uint32_t pair_store = pair_store_ok && ((dense + (pair_store_ok >> 1)) & 1) == 0;
      const uint32_t top_bits = top << table_bits;
      scan_packed<CHUNK>((const char*)L + (uint64_t)p * LV.cap * 6, (uint32_t)lb, (uint32_t)le, stage, wave, kNW, lane,
                         [&](uint32_t i, uint32_t p0, uint32_t h0, uint32_t p1, uint32_t h1) {
                           uint32_t xs[2], fk[2];
                           bool ok[2];
#pragma unroll
                           for (int h = 0; h < 2; ++h) {
                             ok[h] = i + h < le;  // (lb is a multiple of the chunk: nothing in front of it is visited)
                             const uint32_t slot = (h ? h1 : h0) & smask;
                             fk[h] = wang_hash_inv(top_bits | slot);
                             const bool hit = (s_used[slot >> 5] >> (slot & 31)) & 1u;
                             xs[h] = s_vals[slot];
                             if (ok[h] && !hit) {
                               ++my_miss;
                               if (miss_bits) atomicOr(&miss_bits[(i + h + dense) >> 5], 1u << ((i + h + dense) & 31));  // drop mode: mark the row
                             }
                           }
                           const uint64_t o = i + dense;
                           if (ok[0] && ok[1] && pair_store) {
                             *(uint2*)(out_fk + o) = make_uint2(fk[0], fk[1]);
                             *(uint2*)(out_y + o) = make_uint2(p0, p1);
                             *(uint2*)(out_x + o) = make_uint2(xs[0], xs[1]);
                           } else {
                             if (ok[0]) {
                               out_fk[o] = fk[0];
                               out_y[o] = p0;
                               out_x[o] = xs[0];
                             }
                             if (ok[1]) {
                               out_fk[o + 1] = fk[1];
                               out_y[o + 1] = p1;
                               out_x[o + 1] = xs[1];
                             }
                           }
                         });
    } else {
      const uint64_t llo = LV.cap ? (uint64_t)p * LV.cap : LV.off[p];
      const uint64_t delta = (LV.cap ? LV.dense[p] : llo) - llo;
      const uint32_t pair_store = pair_store_ok && ((delta + (pair_store_ok >> 1)) & 1) == 0;
      const uint64_t glb = llo + lb, gle = llo + le;
      const uint64_t a = glb & ~1ull;
      const uint64_t span = gle - a;
      for (uint64_t base = 0; base < span; base += BLOCK * 2 * kJU) {
        uint4 v[kJU];
#pragma unroll
        for (int u = 0; u < kJU; ++u) {
          const uint64_t i0 = base + (uint64_t)(u * BLOCK + threadIdx.x) * 2;
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
          const uint64_t i0 = a + base + (uint64_t)(u * BLOCK + threadIdx.x) * 2;
          uint32_t xs[2] = {0, 0};
          bool ok[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t k = h ? v[u].z : v[u].x;
            ok[h] = i0 + h >= glb && i0 + h < gle;
            if (!ok[h]) continue;
            const uint32_t hk = wang_hash(k);
            if (n_r && (hk >> table_bits) != top) ++my_foreign;
            const uint32_t slot = hk & smask;
            const bool hit = (s_used[slot >> 5] >> (slot & 31)) & 1u;
            xs[h] = s_vals[slot];
            if (!hit) {
              ++my_miss;
              if (miss_bits) atomicOr(&miss_bits[(i0 + h + delta) >> 5], 1u << ((i0 + h + delta) & 31));
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
  };
  for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) body(p, 0u);
  const uint32_t n_extra = slices.count ? min(*slices.count, slices.capacity) : 0u;
  for (uint32_t w = blockIdx.x; w < n_extra; w += gridDim.x) body(slices.list[w].x, slices.list[w].y);
  const uint64_t wm = wave_reduce_add((uint64_t)my_miss);
  const uint64_t wd = wave_reduce_add((uint64_t)my_dup);
  const uint64_t wf = wave_reduce_add((uint64_t)my_foreign);
  if (lane == 0) {
    if (wm) atomicAdd(&counters[0], (unsigned long long)wm);
    if (wd) atomicAdd(&counters[1], (unsigned long long)wd);
    if (wf) atomicAdd(&counters[5], (unsigned long long)wf);
  }
  if (threadIdx.x == 0) atomicMax(&counters[3], (unsigned long long)my_max);
}

}  // namespace olap
