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
//   * k_part_scatter_ring6: the ring scatter of ring_scatter.h with per-bucket rings of 256 bytes (two lines); a tuple
//     is three 2-byte LDS stores behind one returning LDS atomic (+6 on the bucket's byte count); complete lines
//     leave as 8 lanes x 16 B exactly as before, but one tile behind the insert (the kernel is bound by instruction issue,
//     not by HBM: insert and flush overlap).  42 tuples fit a ring instead of 32.
//   * k_join_probe_direct6: every wave copies whole 16-byte vectors of its chunk (coalesced) into a private LDS
//     staging strip and reads them back as pairs of tuples (12 bytes = 3 dwords per lane, conflict free), so loads
//     AND result stores stay lane-contiguous.
#pragma once
#include <type_traits>

#include "probe_kernels.h"
#include "ring_scatter.h"

#ifndef OLAP_PACK6_NT_LOAD
#define OLAP_PACK6_NT_LOAD 0  // A/B builds (scripts/build_variant.sh): non-temporal loads of the packed last pass's input tuples
#endif
#ifndef OLAP_PROBE6_NT_LOAD
#define OLAP_PROBE6_NT_LOAD 0  // ... of the packed probe's chunks
#endif
#if OLAP_PROBE6_NT_LOAD
#define OLAP_PROBE6_LOAD gload4_nt
#else
#define OLAP_PROBE6_LOAD gload4
#endif

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
// LDS: the bucket words first (a bucket's word sits at byte 4 b, its ring at kRing + (b << ring shift): both follow from ONE
// register per tuple, the ring offset), then the rings -- F of them plus F / 16 dummy rings that swallow invalid tuples
// (fillers, edge tiles) without a branch: 17/16 x 128 KiB for every fan-out.
template <int FMAX, int BLOCK, int NPROD>
struct Ring6Cfg {
  static constexpr int kRingLog2 = 17;             // all rings together: 128 KiB (FMAX buckets x two 128-byte lines)
  static_assert(FMAX * 256 == 1 << kRingLog2, "two lines per bucket at FMAX buckets");
  static constexpr int kNW = BLOCK / kWave;
  static constexpr int kNF = kNW - NPROD;
  static constexpr int kProd = NPROD * kWave;
  static_assert(FMAX <= kNF * kWave, "one bucket per flusher lane");
  static_assert(FMAX <= BLOCK, "the closing scan runs one bucket per thread");
  static constexpr int kSlots = FMAX + FMAX / 16;
  static constexpr uint32_t kCl = 0;                                   // per ring: bytes in the ring | ring head << 16
  static constexpr uint32_t kMisc = 2304;                              // [0] hot flag, [1] the segment's leading hash bits, scan scratch
  static_assert(kSlots * 4 <= kMisc, "bucket words");
  static constexpr uint32_t kHot = kMisc + 64 + kNW * 8;               // per bucket: hot-tile hand-off {base, full | head << 16}
  static constexpr int kListPerWave = (FMAX / kNF) * 2;                // a wave's buckets x lines per ring: the same for every fan-out
  static constexpr uint32_t kList = kHot + FMAX * 8;                   // per flusher wave: lines to emit {line index, ring address}
  static constexpr uint32_t kRing = (kList + kNF * kListPerWave * 8 + 4095) / 4096 * 4096;
  static constexpr uint32_t kBytes = kRing + (kSlots << 8);
  static_assert(kRing < 65536, "the ring area's offset is an immediate of the LDS stores");
};

// Capacity layout only (one workgroup per input segment, CapOut), tuple input only.  out: the packed regions as halfwords.
// ptop[seg * F + b] = the hash bits above `shift` that every tuple of partition (seg, b) carries.  CHECK: the input's
// segments are not this plan's own pass-0 regions (skipped hash bits, pre-split input) -- the leading bits are then
// taken from the tuples, and *foreign counts threads that saw tuples of one segment disagree about them (a violated
// olap_join_set_hash_skip_bits promise); otherwise they are the segment's index.
template <int FMAX, int BLOCK, int NPROD, int TPT, bool CHECK, bool PIPE>
__global__ __launch_bounds__(BLOCK) void k_part_scatter_ring6(PassSrc s, uint32_t shift, uint32_t fbits, uint16_t* out, CapOut co,
                                                                uint32_t* ptop, unsigned long long* foreign) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = Ring6Cfg<FMAX, BLOCK, NPROD>;
  constexpr int kNF = C::kNF, kProd = C::kProd;
  constexpr int TILE = kProd * TPT;
  constexpr uint32_t LB = 128;        // bytes per line
  constexpr int LPL = 8;              // lanes per line (16 B each)
  constexpr int LPI = kWave / LPL;    // lines per store instruction
  uint32_t* cl = (uint32_t*)(smem + C::kCl);
  uint2* hot = (uint2*)(smem + C::kHot);
  uint32_t* misc = (uint32_t*)(smem + C::kMisc);
  char* rings = smem + C::kRing;
  const uint32_t F = 1u << fbits;
  const uint32_t rsh = (uint32_t)C::kRingLog2 - fbits;  // log2 of a ring's bytes (>= 8)
  const uint32_t RB = 1u << rsh, M = RB - 1u;
  const uint32_t nd = F >= 16u ? F >> 4 : 1u;           // dummy rings
  const TileRange r = block_tiles<SRC_TUPLES, TILE>(s);
  const uint32_t oseg = s.out_by_block ? blockIdx.x : r.seg;  // what the output regions are numbered by (PassSrc::out_by_block)
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x / kWave;
  const uint32_t cap_l = co.cap * 3u / 64u;  // lines per region (cap is a multiple of 64 tuples; the host keeps P * cap_l below 2^32)
  for (uint32_t b = threadIdx.x; b < F + nd; b += BLOCK) cl[b] = 0;
  if (threadIdx.x == 0) {
    misc[0] = 0;
    misc[1] = CHECK ? kRingNone : r.seg;
  }
  __syncthreads();
  uint32_t acc_or = 0, acc_and = 0xFFFFFFFFu;  // CHECK: OR / AND of the hashes of the valid tuples this thread saw

  if (wave < (unsigned)NPROD) {
    // ------------------------------------------------------------ producers
    const uint32_t expect = r.seg & s.chk_mask;
    const uint32_t dummy_bs = (F + (lane & (nd - 1u))) << rsh;  // ring offset of this lane's dummy ring
    const uint32_t lim = RB - 6u;                               // a tuple fits while the ring holds at most this many bytes
    using Regs = TileRegs<TPT, SRC_TUPLES, 0>;
    // k[i] <- ring offset of the tuple's bucket (or dummy ring), bk[i] <- its hash (the low 16 bits are what is stored)
    const uint32_t chk_bits = (uint32_t)__popc(s.chk_mask);  // (the mask is a run of low bits, or 0: every tuple in range is real)
    auto hash_tile = [&](Regs& t) {
      auto body = [&](auto use_mask, auto hashed) {
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          const uint32_t h = decltype(hashed)::value ? t.k[i] : wang_hash(t.k[i]);  // (hashed: the first pass left the hash in place of the key)
          bool ok = __builtin_amdgcn_ubfe(h, s.chk_shift, chk_bits) == expect;
          if (decltype(use_mask)::value) ok &= ((t.mask >> i) & 1u) != 0;
          t.k[i] = ok ? __builtin_amdgcn_ubfe(h, shift, fbits) << rsh : dummy_bs;
          t.bk[i] = h;
          if (CHECK) {
            acc_or |= ok ? h : 0u;
            acc_and &= ok ? h : 0xFFFFFFFFu;
          }
        }
      };
      const bool whole = t.mask == (1u << TPT) - 1u;  // (uniform: whole tiles do not look at the mask again)
      if (s.hashed) {
        if (whole) body(std::false_type{}, std::true_type{});
        else body(std::true_type{}, std::true_type{});
      } else {
        if (whole) body(std::false_type{}, std::false_type{});
        else body(std::true_type{}, std::false_type{});
      }
    };
    ColsCursor cur;
    Regs A, B, Cc;
    A.mask = B.mask = Cc.mask = 0;
#pragma unroll
    for (int i = 0; i < TPT; ++i) A.k[i] = B.k[i] = Cc.k[i] = dummy_bs;
    if (r.t0 < r.t1) A.template load<kProd, TILE, OLAP_PACK6_NT_LOAD != 0>(s, r, r.t0, cur);
    if (r.t0 + 1 < r.t1) B.template load<kProd, TILE, OLAP_PACK6_NT_LOAD != 0>(s, r, r.t0 + 1, cur);
    hash_tile(A);
    auto hw_of = [](uint32_t pay, uint32_t h, int j) -> uint16_t { return j == 0 ? (uint16_t)pay : j == 1 ? (uint16_t)(pay >> 16) : (uint16_t)h; };
#ifdef OLAP_EXP_PHASE_CLOCKS
    const bool clk_on = (blockIdx.x & 15) == 0 && threadIdx.x == 0;
#endif
    auto step = [&](uint64_t tile, Regs& c, Regs& nx, Regs& n2) {
      OLAP_CLK(c0);
      // insert tile `tile`: one returning LDS atomic and three 2-byte LDS stores per tuple
      uint32_t v[TPT];
#pragma unroll
      for (int i = 0; i < TPT; ++i) v[i] = atomicAdd((uint32_t*)(smem + C::kCl + (c.k[i] >> (rsh - 2u))), 6u);
      uint32_t mx = 0;
#pragma unroll
      for (int i = 0; i < TPT; ++i) mx = max(mx, v[i] & 0xFFFFu);
      uint32_t dmask = 0;
      if (!__ballot(mx > lim)) {  // every tuple of this wave fits its ring (dummy rings are reset every tile): no selects
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          const uint32_t p = (v[i] >> 16) + (v[i] & 0xFFFFu), bs = c.k[i];
          *(uint16_t*)(rings + (bs | (p & M))) = (uint16_t)c.p[i];
          *(uint16_t*)(rings + (bs | ((p + 2u) & M))) = (uint16_t)(c.p[i] >> 16);
          *(uint16_t*)(rings + (bs | ((p + 4u) & M))) = (uint16_t)c.bk[i];
        }
      } else {  // a ring of this wave overflows (or a dummy ring ran over in an edge tile): halfword by halfword
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          const uint32_t cnt = v[i] & 0xFFFFu, hd = v[i] >> 16, bs = c.k[i];
          const bool valid = bs < (F << rsh);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const uint32_t q = cnt + 2u * j;
            // what lies beyond the ring is placed after the flush; a dummy ring just wraps
            const uint32_t at = (!valid | (q < RB)) ? (bs | ((hd + q) & M)) : (dummy_bs | (2u * j));
            *(uint16_t*)(rings + at) = hw_of(c.p[i], c.bk[i], j);
          }
          dmask |= (valid & (cnt > lim)) ? 1u << i : 0u;
        }
        if (dmask) misc[0] = (uint32_t)tile + 1u;
      }
#ifdef OLAP_EXP_PHASE_CLOCKS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      OLAP_CLK(c1);
      __syncthreads();  // ---- barrier 1: this tile's inserts are in the rings
      OLAP_CLK(c2);
#ifdef OLAP_EXP_PHASE_CLOCKS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      OLAP_CLK(c2a);
      nx.pin();
      hash_tile(nx);
#pragma unroll
      for (int i = 0; i < TPT; ++i) asm volatile("" : "+v"(nx.k[i]) : : "memory");
      n2.mask = 0;
      if (tile + 2 < r.t1) n2.template load<kProd, TILE, OLAP_PACK6_NT_LOAD != 0>(s, r, tile + 2, cur);
      OLAP_CLK(c3);
      __syncthreads();  // ---- barrier 2: the rings' words are up to date
      OLAP_CLK(c4);
      OLAP_CLK_ADD(1, 0, c1 - c0);
      OLAP_CLK_ADD(1, 1, c2 - c1);
      OLAP_CLK_ADD(1, 2, c3 - c2);
      OLAP_CLK_ADD(1, 3, c4 - c3);
      OLAP_CLK_ADD(1, 8, c2a - c2);
      if (misc[0] == (uint32_t)tile + 1u) {  // some bucket was hot in this tile (uniform)
        OLAP_CLK_ADD(1, 9, 1);
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          if (dmask & (1u << i)) {
            const uint32_t bs = c.k[i], b = bs >> rsh;
            const uint2 info = hot[b];
            const uint32_t full = info.y & 0xFFFFu, hd = info.y >> 16, cnt = v[i] & 0xFFFFu;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const uint32_t q = cnt + 2u * j;
              if (q < RB) continue;  // already in the ring (flushed with it)
              const uint16_t val = hw_of(c.p[i], c.bk[i], j);
              if (q < full) {  // straight to the range the owner reserved (dropped when the region is full)
                if (info.x != kRingNone) gstore_t(out + (((uint64_t)oseg * F + b) * cap_l * LB + info.x + (q - RB)) / 2u, val);
              } else {         // tail of the tile: the bucket's new residue
                *(uint16_t*)(rings + (bs | ((hd + (q - full)) & M))) = val;
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
    const uint32_t p_own = oseg * F + b_own;
    const uint32_t line0 = p_own * cap_l;                    // the region's first line
    const uint32_t ring_own = C::kRing + (b_own << rsh);     // LDS address of the bucket's ring
    uint2* list = (uint2*)(smem + C::kList) + fw * C::kListPerWave;
    uint32_t pos = 0;     // next line of the region
    bool dead = !owner;   // region exhausted (skew): tuples are dropped, *overflow raised
    // The flush runs ONE TILE BEHIND the insert: the lines that are complete after tile t are listed between the two barriers
    // of tile t and leave (LDS -> HBM) while the producers insert tile t + 1; they stay in the ring until the bookkeeping of
    // tile t + 1 releases them.  `pend`: this owner's lines listed for the coming emission; n_list: the wave's.
    uint32_t pend = 0, n_list = 0;
    const uint32_t lines_per_ring = RB / LB;
    // The wave emits the listed lines eight per store instruction, kFU instructions' worth at a time -- list entries, then
    // ring reads, then stores, each kind back to back, so that a batch costs two LDS round trips instead of two per instruction.
    constexpr int kFU = TPT >= 8 ? 4 : 2;
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
    // Lists `count` lines of every owner (ring lines from byte `ring0` on, output from line `pos` on) in rounds: round q takes
    // line q of the owners that have one (a ballot and a lane count per round; two rounds at 512 buckets).  Returns the total.
    auto list_rounds = [&](uint32_t count, uint32_t ring0) -> uint32_t {
      uint32_t n = 0;
      for (uint32_t q = 0;; ++q) {
        const bool has = count > q;
        const unsigned long long m = __ballot(has);
        if (!m) break;
        if (has) list[n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
            make_uint2(line0 + pos + q, ring_own + ((ring0 + q * LB) & M));
        n += (uint32_t)__popcll(m);
      }
      pos += count;
      return n;
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
      if (fw == 0 && lane < nd) cl[F + lane] = 0;  // the dummy rings start every tile empty
      const uint32_t tot0 = vv & 0xFFFFu, hd0 = vv >> 16;
      const uint32_t cnt1 = tot0 - pend * LB;  // release the lines that went out during the insert
      const uint32_t hd1 = (hd0 + pend * LB) & M;
      const uint32_t nl = cnt1 / LB;           // complete lines now: they go out during the next insert
      uint32_t mine = nl;
      // (tot0 > RB: the window of this tile's insert -- RB bytes from hd0, the listed lines included -- was overrun)
      if (__ballot(owner & ((tot0 > RB) | dead | (pos + nl > cap_l)))) {  // rare: a hot bucket, an exhausted region
        const bool hotb = tot0 > RB;
        if (__ballot(hotb)) {
          // A hot bucket's ring is exactly full; its first `pend` lines have just left.  The rest leaves NOW, so that the
          // ring is empty for what did not fit; whole lines' worth of that goes out directly, the rest is the new residue.
          uint32_t imm = 0;
          if (hotb && !dead) {
            imm = lines_per_ring - pend;
            if (pos + imm > cap_l) {
              dead = true;
              atomicAdd(co.overflow, 1ull);
              imm = 0;
            }
          }
          const uint32_t n_imm = list_rounds(imm, hd0 + pend * LB);
          __builtin_amdgcn_wave_barrier();
          emit_lines(n_imm);
          __builtin_amdgcn_wave_barrier();
          if (hotb) {
            const uint32_t full = tot0 & ~(LB - 1);
            const uint32_t ndl = (full - RB) / LB;  // lines that go out directly
            uint32_t dbase = kRingNone;
            if (ndl && !dead) {
              if (pos + ndl > cap_l) {
                dead = true;
                atomicAdd(co.overflow, 1ull);
              } else {
                dbase = pos * LB;
                pos += ndl;
              }
            }
            hot[b_own] = make_uint2(dbase, full | (hd0 << 16));
            cl[b_own] = (tot0 - full) | (hd0 << 16);
          }
        }
        mine = 0;
        if (owner && !hotb) {
          if (nl && !dead) {
            if (pos + nl > cap_l) {
              dead = true;
              atomicAdd(co.overflow, 1ull);
            } else {
              mine = nl;
            }
          }
          if (dead || !PIPE) cl[b_own] = (cnt1 - nl * LB) | (((hd1 + nl * LB) & M) << 16);  // dropped (or, not pipelined, emitted) at once
          else cl[b_own] = cnt1 | (hd1 << 16);
        }
      } else if (owner) {
        cl[b_own] = PIPE ? cnt1 | (hd1 << 16) : (cnt1 - nl * LB) | (((hd1 + nl * LB) & M) << 16);
      }
      n_list = list_rounds(mine, hd1);
      if (PIPE) {
        pend = mine;
      } else {  // large tiles (a ring cannot hold a tile's lines beside the next tile's tuples): the lines leave at once
        __builtin_amdgcn_wave_barrier();
        emit_lines(n_list);
        n_list = 0;
      }
      OLAP_CLK(f2);
      __syncthreads();  // ---- barrier 2: the rings' words are up to date
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
      const uint32_t rr = (vv & 0xFFFFu) - pend * LB, hd = ((vv >> 16) + pend * LB) & M;
      uint32_t count_b = pos * LB;
      if (!dead) {
        if (rr && pos >= cap_l) {
          atomicAdd(co.overflow, 1ull);
        } else {
          uint16_t* dst = out + ((uint64_t)(line0 + pos) * LB) / 2u;
          for (uint32_t j = 0; j < rr; j += 2) dst[j / 2] = *(const uint16_t*)(smem + ring_own + ((hd + j) & M));
          count_b += rr;
        }
      }
      ((uint32_t*)hot)[b_own] = count_b / 6u;  // tuples, for the closing scan
    }
  }
  // partition sizes, dense starts and leading hash bits (exclusive scan over this segment's buckets, one per thread)
  if (CHECK && acc_and != 0xFFFFFFFFu) misc[1] = acc_or >> (shift + fbits);  // (any thread's: they all agree, or `foreign` says so)
  __syncthreads();
  const uint32_t segtop = misc[1];
  if (CHECK && acc_and != 0xFFFFFFFFu && (((acc_or ^ acc_and) >> (shift + fbits)) != 0 || (acc_or >> (shift + fbits)) != segtop))
    atomicAdd(foreign, 1ull);
  const uint32_t t = threadIdx.x;
  const uint64_t count = t < F ? ((const uint32_t*)hot)[t] : 0u;
  uint64_t* wtot = (uint64_t*)(smem + C::kMisc + 64);
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
    co.cnt[(uint64_t)oseg * F + t] = (uint32_t)count;
    co.dense[(uint64_t)oseg * F + t] = (co.seg_total ? 0 : r.lo) + pre;
    ptop[(uint64_t)oseg * F + t] = segtop == kRingNone ? 0u : ((segtop << fbits) | t);
  }
  if (co.seg_total && threadIdx.x == 0) {
    uint64_t total = 0;
    for (int w = 0; w < BLOCK / kWave; ++w) total += wtot[w];
    co.seg_total[oseg] = total;
  }
}

// ------------------------------------------------------------------ build + probe over packed partitions
// A wave walks the chunks (CHUNK tuples = CHUNK * 6 bytes, a whole number of 16-byte vectors) of a packed region that
// fall to it: vectors in with coalesced 16-byte loads (the next chunk in flight meanwhile), through its private staging
// strip, out as PAIRS of tuples -- lane l takes pairs l, l + 64, ... : 12 bytes at dword 3 * pair, so the 64 lanes of a
// read hit 64 different banks and the pair {2m, 2m + 1} needs no byte shuffling beyond one funnel shift.
// f(i, pay0, h0, pay1, h1): tuples i (even) and i + 1 of the region.
// A wave's FIRST chunk of a region, requested ahead of the walk (k_join_probe_direct6_pf: the next partition's loads leave while
// the current one is probed).
template <int CHUNK>
struct PackedChunk {
  static constexpr int NP = CHUNK * 6 / 16;
  static constexpr int LPL = (NP + kWave - 1) / kWave;
  uint4 v[LPL];
  __device__ __forceinline__ void load(const char* region, uint32_t lo, uint32_t hi, unsigned wave, unsigned lane) {
    const uint32_t c = lo / CHUNK + wave;
#pragma unroll
    for (int k = 0; k < LPL; ++k) {
      const uint32_t piece = k * kWave + lane;
      v[k] = make_uint4(0, 0, 0, 0);
      if (c * CHUNK < hi && (NP % kWave == 0 || piece < (uint32_t)NP)) v[k] = OLAP_PROBE6_LOAD(region + ((uint64_t)c * NP + piece) * 16);
    }
  }
};

template <int CHUNK, class Fn>
__device__ __forceinline__ void scan_packed(const char* region, uint32_t lo, uint32_t hi, char* stage, unsigned wave, unsigned nwaves,
                                            unsigned lane, Fn&& f, const PackedChunk<CHUNK>* pre = nullptr) {
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
      if (NP % kWave == 0 || piece < (uint32_t)NP) v[k] = OLAP_PROBE6_LOAD(region + ((uint64_t)cc * NP + piece) * 16);
    }
  };
  if (pre) {
#pragma unroll
    for (int k = 0; k < LPL; ++k) cur[k] = pre->v[k];
  } else if (c < c1) {
    load(c, cur);
  }
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

// The same walk over SEVERAL packed regions (the sub-partitions of one partition, PartView::nsub): the chunks of all regions
// form one sequence that the waves share round-robin, and a wave's next chunk -- of whichever region -- is in flight while it
// works on the current one, so that short regions (a few chunks each) do not cost a load latency apiece.
// region(q) -> {base, tuples}: region q as bytes and its length n; f(q, n, i, pay0, h0, pay1, h1): tuples i, i + 1 of region q.
struct PackedRegion {
  const char* base;
  uint32_t n;
};
template <int CHUNK, class Region, class Fn>
__device__ __forceinline__ void scan_packed_multi(uint32_t nregions, Region&& region, char* stage, unsigned wave, unsigned nwaves, unsigned lane,
                                                  Fn&& f) {
  static_assert(CHUNK % 128 == 0, "whole pairs per lane");
  constexpr int NP = CHUNK * 6 / 16;       // 16-byte vectors per chunk
  constexpr int LPL = (NP + kWave - 1) / kWave;
  uint4 cur[LPL], nxt[LPL];
#pragma unroll
  for (int k = 0; k < LPL; ++k) cur[k] = nxt[k] = make_uint4(0, 0, 0, 0);
  // this wave's chunks in order: (q, c) = chunk c of region q; `skip` chunks of the shared sequence lie in front of the next one
  uint32_t q = 0, c0 = 0, skip = wave;
  PackedRegion rg = nregions ? region(0u) : PackedRegion{nullptr, 0u};
  uint32_t c1 = (rg.n + CHUNK - 1) / CHUNK;
  auto advance = [&](uint32_t& oq, uint32_t& oc, const char*& obase, uint32_t& on) -> bool {
    for (;;) {
      if (q >= nregions) return false;
      const uint32_t avail = c1 - c0;
      if (skip < avail) {
        oq = q;
        oc = c0 + skip;
        obase = rg.base;
        on = rg.n;
        c0 = oc + 1;
        skip = nwaves - 1;
        return true;
      }
      skip -= avail;
      ++q;
      c0 = 0;
      if (q < nregions) {
        rg = region(q);
        c1 = (rg.n + CHUNK - 1) / CHUNK;
      }
    }
  };
  auto load = [&](const char* base, uint32_t cc, uint4 (&v)[LPL]) {
#pragma unroll
    for (int k = 0; k < LPL; ++k) {
      const uint32_t piece = k * kWave + lane;
      if (NP % kWave == 0 || piece < (uint32_t)NP) v[k] = OLAP_PROBE6_LOAD(base + ((uint64_t)cc * NP + piece) * 16);
    }
  };
  uint32_t cq = 0, cc = 0, cn = 0, nq = 0, nc = 0, nn = 0;
  const char *cb = nullptr, *nb = nullptr;
  bool have = advance(cq, cc, cb, cn);
  if (have) load(cb, cc, cur);
  while (have) {
    const bool more = advance(nq, nc, nb, nn);
    if (more) load(nb, nc, nxt);
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
      f(cq, cn, cc * CHUNK + 2 * m, d0, d1 & 0xFFFFu, (d1 >> 16) | (d2 << 16), d2 >> 16);
    }
#pragma unroll
    for (int k = 0; k < LPL; ++k) cur[k] = nxt[k];
    cq = nq;
    cc = nc;
    cn = nn;
    have = more;
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

// k_join_probe_direct6<.., true, true> with the loads running AHEAD of the partition's phases: a small partition (2048 + 2048
// tuples at 2^29 rows per side: one chunk per wave and side) is a chain of latencies -- sizes, build chunk, probe chunk -- so the
// next work item's sizes are fetched while the table is built, its build chunk while the current one is probed, and the probe chunk
// before the barrier that completes the table.  Work items: the partitions, then the (partition, slice >= 1) pairs.
template <int BLOCK, int CHUNK>
__global__ __launch_bounds__(BLOCK) void k_join_probe_direct6_pf(PartView RV, PartView LV, uint32_t nparts, uint32_t table_bits, uint32_t* out_fk,
                                                                   uint32_t* out_y, uint32_t* out_x, uint32_t pair_store_ok,
                                                                   unsigned long long* counters, uint32_t* miss_bits, ProbeSlices slices) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* s_vals = (uint32_t*)smem;
  uint32_t* s_used = (uint32_t*)(smem + kDSlots * 4);
  const unsigned lane = lane_id(), wave = threadIdx.x / kWave;
  constexpr unsigned kNW = BLOCK / kWave;
  char* stage = smem + kDSlots * 4 + kDSlots / 8 + wave * (CHUNK * 6);
  const char* R = (const char*)RV.t;
  const char* L = (const char*)LV.t;
  const uint32_t slots = 1u << table_bits, smask = slots - 1;
  uint32_t my_miss = 0, my_dup = 0, my_foreign = 0;
  uint64_t my_max = 0;
  const uint32_t n_extra = slices.count ? min(*slices.count, slices.capacity) : 0u;
  const uint32_t total = nparts + n_extra;
  struct Item {
    uint32_t p, slice, n_r, n_l, top, rtop;
    uint64_t dense;
  };
  auto fetch = [&](uint32_t w) -> Item {
    Item it{0u, 0u, 0u, 0u, 0u, 0u, 0ull};
    if (w < total) {
      if (w < nparts) {
        it.p = w;
      } else {
        const uint2 e = slices.list[w - nparts];
        it.p = e.x;
        it.slice = e.y;
      }
      it.n_r = RV.cnt[it.p];
      it.n_l = LV.cnt[it.p];
      it.top = LV.top[it.p];
      it.rtop = RV.top[it.p];
      it.dense = LV.dense[it.p];
    }
    return it;
  };
  auto range = [&](const Item& it, uint32_t& lb, uint32_t& le) {  // probe rows [lb, le) of the item's partition
    lb = it.slice * kProbeSlice;
    le = slices.count ? min(it.n_l, lb + kProbeSlice) : it.n_l;
  };
  PackedChunk<CHUNK> rpre, lpre;
  Item cur = fetch(blockIdx.x);
  {
    uint32_t lb, le;
    range(cur, lb, le);
    if (lb < le) rpre.load(R + (uint64_t)cur.p * RV.cap * 6, 0u, cur.n_r, wave, lane);
  }
  for (uint32_t w = blockIdx.x; w < total; w += gridDim.x) {
    const Item nxt = fetch(w + gridDim.x);  // (its figures arrive while this item's table is built)
    uint32_t lb, le;
    range(cur, lb, le);
    if (cur.n_r > my_max) my_max = cur.n_r;
    const bool work = lb < le;
    if (work) {
      lpre.load(L + (uint64_t)cur.p * LV.cap * 6, lb, le, wave, lane);  // (in flight while the table is built)
      if (cur.n_r && cur.rtop != cur.top && threadIdx.x == 0) my_foreign += cur.n_l;
      for (uint32_t i = threadIdx.x; i < (slots + 31) / 32; i += BLOCK) s_used[i] = 0;
      __syncthreads();
      const uint32_t n_r = cur.n_r, slice = cur.slice;
      scan_packed<CHUNK>(R + (uint64_t)cur.p * RV.cap * 6, 0u, n_r, stage, wave, kNW, lane,
                         [&](uint32_t i, uint32_t p0, uint32_t h0, uint32_t p1, uint32_t h1) {
#pragma unroll
                           for (int h = 0; h < 2; ++h) {
                             if (i + h >= n_r) continue;
                             const uint32_t slot = (h ? h1 : h0) & smask;
                             const uint32_t bit = 1u << (slot & 31);
                             if ((atomicOr(&s_used[slot >> 5], bit) & bit) && slice == 0) ++my_dup;  // equal key overwrites (hashtable.c:133)
                             s_vals[slot] = h ? p1 : p0;
                           }
                         },
                         &rpre);
      __syncthreads();
    }
    {  // the next item's first build chunk
      uint32_t nlb, nle;
      range(nxt, nlb, nle);
      if (nlb < nle) rpre.load(R + (uint64_t)nxt.p * RV.cap * 6, 0u, nxt.n_r, wave, lane);
    }
    if (work) {
      const uint64_t dense = cur.dense;
      const uint32_t pair_store = pair_store_ok && ((dense + (pair_store_ok >> 1)) & 1) == 0;
      const uint32_t top_bits = cur.top << table_bits;
      scan_packed<CHUNK>(L + (uint64_t)cur.p * LV.cap * 6, lb, le, stage, wave, kNW, lane,
                         [&](uint32_t i, uint32_t p0, uint32_t h0, uint32_t p1, uint32_t h1) {
                           uint32_t xs[2], fk[2];
                           bool ok[2];
#pragma unroll
                           for (int h = 0; h < 2; ++h) {
                             ok[h] = i + h < le;
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
                         },
                         &lpre);
      __syncthreads();
    }
    cur = nxt;
  }
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

// k_join_probe_direct6 over sub-partitions (PartView::nsub; both sides packed): what a rank joins after the all-to-all, when its
// last pass ran chunk by chunk (one workgroup per (segment, sender) chunk, PassSrc::out_by_block) because the segments alone are
// too few to fill the chip.  Partition p = (segment j, bucket b): its table is built from the right side's sub-partitions of all
// chunks of segment j, then the left side's sub-partitions are probed one after the other; result rows are (segment, chunk,
// bucket)-major.  No probe slices (a skewed side overruns its regions first and is redone with the exact layout).
constexpr int kSubBlocks = 4;  // sub-partitions per partition: at most kSubBlocks x 64 (the host falls back to the histogram pipeline beyond)
template <int BLOCK, int CHUNK>
__global__ __launch_bounds__(BLOCK) void k_join_probe_direct6_sub(PartView RV, PartView LV, uint32_t nparts, uint32_t table_bits, uint32_t* out_fk,
                                                                    uint32_t* out_y, uint32_t* out_x, uint32_t pair_store_ok,
                                                                    unsigned long long* counters, uint32_t* miss_bits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* s_vals = (uint32_t*)smem;
  uint32_t* s_used = (uint32_t*)(smem + kDSlots * 4);
  const unsigned lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  constexpr unsigned kNW = BLOCK / kWave;
  char* stage = smem + kDSlots * 4 + kDSlots / 8 + wave * (CHUNK * 6);
  const char* R = (const char*)RV.t;
  const char* L = (const char*)LV.t;
  const uint32_t slots = 1u << table_bits, smask = slots - 1;
  const uint32_t rn = RV.nsub ? RV.nsub : 1u, ln = LV.nsub ? LV.nsub : 1u;
  uint32_t my_miss = 0, my_dup = 0, my_foreign = 0;
  uint64_t my_max = 0;
  for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) {
    const uint32_t rj = p >> RV.sub_fbits, rb = p & ((1u << RV.sub_fbits) - 1u);
    const uint32_t lj = p >> LV.sub_fbits, lbk = p & ((1u << LV.sub_fbits) - 1u);
    auto rsub = [&](uint32_t q) { return ((rj * rn + q) << RV.sub_fbits) | rb; };
    auto lsub = [&](uint32_t q) { return ((lj * ln + q) << LV.sub_fbits) | lbk; };
    // The sub-partitions' sizes, dense starts and leading hash bits, one sub-partition per LANE (kSubBlocks x 64 of them at most):
    // a handful of vector loads per partition instead of a chain of scalar ones per sub-partition; a region's figures are then
    // read back with readlane.  Every non-empty sub-partition of either side must agree on the bits above the table bits.
    uint32_t cr[kSubBlocks], cl[kSubBlocks], tl[kSubBlocks], tr[kSubBlocks];
    uint64_t dl[kSubBlocks];
#pragma unroll
    for (int k = 0; k < kSubBlocks; ++k) {
      const uint32_t q = k * kWave + lane;
      cr[k] = cl[k] = tl[k] = tr[k] = 0;
      dl[k] = 0;
      if (k * kWave < (int)rn && q < rn) {
        const uint32_t sp = rsub(q);
        cr[k] = RV.cnt[sp];
        tr[k] = RV.top[sp];
      }
      if (k * kWave < (int)ln && q < ln) {
        const uint32_t sp = lsub(q);
        cl[k] = LV.cnt[sp];
        tl[k] = LV.top[sp];
        dl[k] = LV.dense[sp];
      }
    }
    uint64_t n_r = 0, n_l = 0;
    uint32_t top = 0;
    bool have_top = false;
#pragma unroll
    for (int k = 0; k < kSubBlocks; ++k) {
      n_r += wave_reduce_add((uint64_t)cr[k]);
      n_l += wave_reduce_add((uint64_t)cl[k]);
      const unsigned long long m = __ballot(cl[k] != 0);
      if (!have_top && m) {
        top = __builtin_amdgcn_readlane(tl[k], (int)__builtin_ctzll(m));
        have_top = true;
      }
    }
    n_r = uniform_u64(n_r);
    n_l = uniform_u64(n_l);
    if (n_r > my_max) my_max = n_r;
    if (n_l == 0) continue;
    uint64_t bad = 0;
#pragma unroll
    for (int k = 0; k < kSubBlocks; ++k)
      bad += wave_reduce_add((uint64_t)((cl[k] && tl[k] != top) ? cl[k] : 0u) + (uint64_t)((cr[k] && tr[k] != top) ? cr[k] : 0u));
    if (bad && threadIdx.x == 0) my_foreign += (uint32_t)min(uniform_u64(bad), (uint64_t)0xFFFFFFFFu);
    auto pick32 = [&](const uint32_t (&v)[kSubBlocks], uint32_t q) -> uint32_t {
      uint32_t r = __builtin_amdgcn_readlane(v[0], (int)(q & 63u));
#pragma unroll
      for (int k = 1; k < kSubBlocks; ++k)
        if ((q >> 6) == (uint32_t)k) r = __builtin_amdgcn_readlane(v[k], (int)(q & 63u));
      return r;
    };
    auto pick_dense = [&](uint32_t q) -> uint64_t {
      uint32_t lo32 = __builtin_amdgcn_readlane((uint32_t)dl[0], (int)(q & 63u)), hi32 = __builtin_amdgcn_readlane((uint32_t)(dl[0] >> 32), (int)(q & 63u));
#pragma unroll
      for (int k = 1; k < kSubBlocks; ++k)
        if ((q >> 6) == (uint32_t)k) {
          lo32 = __builtin_amdgcn_readlane((uint32_t)dl[k], (int)(q & 63u));
          hi32 = __builtin_amdgcn_readlane((uint32_t)(dl[k] >> 32), (int)(q & 63u));
        }
      return ((uint64_t)hi32 << 32) | lo32;
    };
    for (uint32_t i = threadIdx.x; i < (slots + 31) / 32; i += BLOCK) s_used[i] = 0;
    __syncthreads();
    scan_packed_multi<CHUNK>(
        rn, [&](uint32_t q) { return PackedRegion{R + (uint64_t)rsub(q) * RV.cap * 6, pick32(cr, q)}; }, stage, wave, kNW, lane,
        [&](uint32_t, uint32_t n, uint32_t i, uint32_t p0, uint32_t h0, uint32_t p1, uint32_t h1) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (i + h >= n) continue;
            const uint32_t slot = (h ? h1 : h0) & smask;
            const uint32_t bit = 1u << (slot & 31);
            if (atomicOr(&s_used[slot >> 5], bit) & bit) ++my_dup;  // equal key overwrites (hashtable.c:133)
            s_vals[slot] = h ? p1 : p0;
          }
        });
    __syncthreads();
    const uint32_t top_bits = top << table_bits;
    scan_packed_multi<CHUNK>(
        ln,
        [&](uint32_t q) { return PackedRegion{L + (uint64_t)lsub(q) * LV.cap * 6, pick32(cl, q)}; }, stage, wave, kNW, lane,
        [&](uint32_t q, uint32_t n, uint32_t i, uint32_t p0, uint32_t h0, uint32_t p1, uint32_t h1) {
          const uint64_t dense = pick_dense(q);
          // 8-byte pair stores need an 8-byte aligned element (tuple indices here are even): as k_join_probe_direct
          const uint32_t pair_store = pair_store_ok && ((dense + (pair_store_ok >> 1)) & 1) == 0;
          uint32_t xs[2], fk[2];
          bool ok[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            ok[h] = i + h < n;
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
    __syncthreads();
  }
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
