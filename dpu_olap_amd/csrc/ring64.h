// ring64.h -- the histogram-free radix passes and the probe of olap_join_u64 on 12-BYTE tuples {key lo, key hi, payload}
// (included by join.hip only; host side: join64.h).
//
// Reference: HT_64BIT_KEYS (dpu/shared/hashtable/hashtable.h:13-18, hashtable.c:23-26) with the partition / build / probe
// pipeline of dpu/shared/kernels/partition.c:167-294, hash_build.c:9-35, hash_probe.c:9-46 and the shuffle of
// host/partition/partitioner.cc:280-375.  radix64.h holds the exact-layout pipeline (histogram + scan + tile-staged scatter
// on 16-byte tuples: 224 B moved per row pair); this file is the large-join plan of the 32-bit path (ring_scatter.h,
// ring_claim.h, pack6.h) for 64-bit keys: two ring-scatter passes without any histogram and a probe over the capacity
// layout, 24 (columns) + 24 + 24 + 24 + 24 (tuples) + 16 (result) = 136 B per row pair.
//
// A tuple is three dwords, so -- as in pack6.h, with dwords instead of halfwords -- the rings and the regions are
// STREAMS OF DWORDS: a tuple is one returning LDS atomic (+12 on its bucket's byte count) and three 4-byte LDS stores,
// tuples straddle the 128-byte lines the flushers emit, and only the region geometry keeps tuple boundaries:
//   * claimed layout (pass 0): an allotment is 128 tuples = 1536 B = 12 lines, and a workgroup's stream for a bucket fills
//     its allotments one after the other without gaps, so every allotment starts on a tuple; only the LAST allotment of a
//     (workgroup, bucket) is padded, with filler tuples whose key hashes to another bucket (the next pass drops them);
//   * capacity layout (last pass): partition p owns [p * cap * 12, (p + 1) * cap * 12), cap a multiple of 32 tuples (3 lines).
#pragma once
#include "radix64.h"
#include "ring_scatter.h"

namespace olap {

constexpr uint32_t k64TupleBytes = 12;
constexpr uint32_t k64Allot = 128;       // tuples per claimed allotment
constexpr uint32_t k64AllotLines = 12;   // ... = 1536 B = 12 lines of 128 B
constexpr uint32_t k64CapAlign = 32;     // tuples: capacity regions are whole lines (32 x 12 B = 3 lines)

template <int FMAX, int BLOCK, int NPROD>
struct Ring64Cfg {
  static constexpr int kRingLog2 = 17;             // all rings together: 128 KiB (FMAX buckets x two 128-byte lines)
  static_assert(FMAX * 256 == 1 << kRingLog2, "two lines per bucket at FMAX buckets");
  static constexpr int kNW = BLOCK / kWave;
  static constexpr int kNF = kNW - NPROD;
  static constexpr int kProd = NPROD * kWave;
  static_assert(FMAX <= kNF * kWave, "one bucket per flusher lane");
  static_assert(FMAX <= kProd, "the producers serve the claims, one bucket per thread");
  static_assert(FMAX <= BLOCK, "the closing scan runs one bucket per thread");
  static constexpr int kSlots = FMAX + FMAX / 16;  // rings: the buckets' and the dummies'
  static constexpr uint32_t kCl = 0;                                   // per ring: bytes in the ring | ring head << 16
  static constexpr uint32_t kMisc = 2304;                              // [0] hot flag; scan scratch
  static_assert(kSlots * 4 <= kMisc, "bucket words");
  static constexpr uint32_t kHot = kMisc + 64 + kNW * 8;               // per bucket: hot-tile hand-off {base 1, full | head << 16}
  static constexpr uint32_t kHot2 = kHot + FMAX * 8;                   // ... {bytes at base 1, base 2}
  static constexpr uint32_t kRsv = kHot2 + FMAX * 8;                   // CLAIM, per bucket: the allotment claimed ahead (or kRingNone)
  static constexpr uint32_t kWant = kRsv + FMAX * 4;                   // CLAIM, per bucket: 1 = claim one more, 2 = claim in flight
  static constexpr int kListPerWave = (1 << kRingLog2) / 128 / kNF;    // a wave's buckets x lines per ring: the same for every fan-out
  static constexpr uint32_t kList = kWant + FMAX * 4;                  // per flusher wave: lines to emit {line index, ring address}
  static constexpr uint32_t kRing = (kList + kNF * kListPerWave * 8 + 255) / 256 * 256;
  static constexpr uint32_t kBytes = kRing + (kSlots << 8);
  static_assert(kRing < 65536, "the ring area's offset is an immediate of the LDS stores");
  static_assert(kBytes <= 160 * 1024, "LDS of one CU");
};

template <int TPT>
struct Regs64 {
  uint32_t lo[TPT], hi[TPT], p[TPT], bk[TPT];
  uint32_t mask;
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int i = 0; i < TPT; ++i) asm volatile("" : "+v"(lo[i]), "+v"(hi[i]) : : "memory");
  }
};

struct Cols64Cursor {
  uint32_t batch = 0xFFFFFFFFu;
  const uint64_t* kp = nullptr;
  const uint32_t* pp = nullptr;
  uint64_t rows = 0, row0 = 0;
  uint64_t tile = ~0ull, local_unit = 0, units = 0;
};

struct Tuple12 {
  uint32_t lo, hi, pay;
};
// a 12-byte tuple through ONE global (not flat) load; the tuple is only 4-byte aligned
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef u32x3_t u32x3_a4_t __attribute__((aligned(4)));
__device__ __forceinline__ Tuple12 gload_tuple12(const Tuple12* p) {
  const u32x3_t v = *(const OLAP_GLOBAL u32x3_a4_t*)p;
  return Tuple12{v.x, v.y, v.z};
}

// One tile of the u64 key column + u32 payload column (null: global row index).  Thread t holds rows
// 2 * (j * KPROD + t), + 1 of the tile, j < TPT / 2.
template <int KPROD, int TPT>
__device__ __forceinline__ void load_cols64(const PassSrc& s, uint64_t tile, Cols64Cursor& cur, Regs64<TPT>& t) {
  constexpr int TILE = KPROD * TPT;
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
    cur.kp = (const uint64_t*)uniform_u64((uint64_t)s.key_ptrs[l.batch]);
    cur.pp = s.pay_ptrs ? (const uint32_t*)uniform_u64((uint64_t)s.pay_ptrs[l.batch]) : nullptr;
    cur.rows = uniform_u64(batch_rows(s.v, l.batch));
    cur.row0 = uniform_u64(batch_row0(s.v, l.batch));
    cur.units = (cur.rows + TILE - 1) / TILE;
  }
  const uint64_t r0 = l.local_unit * TILE;
  const uint32_t n = (uint32_t)min((uint64_t)TILE, cur.rows - r0);
  const uint64_t* kp = cur.kp + r0;
  const uint32_t* pp = cur.pp ? cur.pp + r0 : nullptr;
  const uint32_t rowid0 = (uint32_t)(cur.row0 + r0);
  const unsigned tid = threadIdx.x;  // (a producer thread: < KPROD)
  t.mask = 0;
  if (n == (uint32_t)TILE && (((uintptr_t)kp) & 15) == 0 && (((uintptr_t)pp) & 7) == 0) {
#pragma unroll
    for (int j = 0; j < TPT / 2; ++j) {
      const uint32_t e = (j * KPROD + tid) * 2;
      const uint4 kv = gload4(kp + e);
      t.lo[2 * j] = kv.x;
      t.hi[2 * j] = kv.y;
      t.lo[2 * j + 1] = kv.z;
      t.hi[2 * j + 1] = kv.w;
      if (pp) {
        const uint2 pv = gload2(pp + e);
        t.p[2 * j] = pv.x;
        t.p[2 * j + 1] = pv.y;
      } else {
        t.p[2 * j] = rowid0 + e;
        t.p[2 * j + 1] = rowid0 + e + 1;
      }
    }
    t.mask = (1u << TPT) - 1u;
  } else {
#pragma unroll
    for (int j = 0; j < TPT / 2; ++j) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t e = (j * KPROD + tid) * 2 + h;
        const bool ok = e < n;
        const uint64_t k = ok ? gload1_u64(kp + e) : 0ull;
        t.lo[2 * j + h] = (uint32_t)k;
        t.hi[2 * j + h] = (uint32_t)(k >> 32);
        t.p[2 * j + h] = ok ? (pp ? gload1(pp + e) : rowid0 + e) : 0u;
        t.mask |= (ok ? 1u : 0u) << (2 * j + h);
      }
    }
  }
}

// One tile of a claimed region of 12-byte tuples (PassSrc::tuples reinterpreted; fillers included): thread t holds tuples
// j * KPROD + t of the tile, one 12-byte load each (64 lanes: 768 contiguous bytes).
template <int KPROD, int TPT>
__device__ __forceinline__ void load_tuples12(const PassSrc& s, const TileRange& r, uint64_t tile, Regs64<TPT>& t) {
  constexpr int TILE = KPROD * TPT;
  const Tuple12* tv = (const Tuple12*)s.tuples;
  const uint64_t base = r.lo + tile * (uint64_t)TILE;
  t.mask = 0;
#ifndef OLAP_R64_NO_VEC_LOAD
  if (TPT == 4 && base + TILE <= r.hi && (base & 3u) == 0) {
    // a whole tile: thread t takes tuples 4 t .. 4 t + 3 = 48 contiguous, 16-byte aligned bytes as three 16-byte loads
    // (regions start on 1536-byte boundaries)
    const uint4* p = (const uint4*)(tv + base) + 3 * threadIdx.x;
    const uint4 a = gload4(p), b = gload4(p + 1), c = gload4(p + 2);
    t.lo[0] = a.x; t.hi[0] = a.y; t.p[0] = a.z;
    t.lo[1] = a.w; t.hi[1] = b.x; t.p[1] = b.y;
    t.lo[2] = b.z; t.hi[2] = b.w; t.p[2] = c.x;
    t.lo[3] = c.y; t.hi[3] = c.z; t.p[3] = c.w;
    t.mask = 0xFu;
    return;
  }
#endif
#pragma unroll
  for (int j = 0; j < TPT; ++j) {
    const uint64_t i = base + (uint64_t)j * KPROD + threadIdx.x;
    const bool ok = i < r.hi;
    Tuple12 v{0u, 0u, 0u};
    if (ok) v = gload_tuple12(tv + i);
    t.lo[j] = v.lo;
    t.hi[j] = v.hi;
    t.p[j] = v.pay;
    t.mask |= (ok ? 1u : 0u) << j;
  }
}

// CLAIM: pass 0 from the columns, output = claimed allotments (ClaimOut; cap a multiple of k64Allot, cursors in tuples).
// !CLAIM: last pass from the claimed regions of pass 0 (PassSrc::tuples / seg_cnt / seg_cap / chk_*), one workgroup per
// input segment, output = capacity layout (CapOut; cap a multiple of k64CapAlign).  out: the output buffer as dwords.
// Tiles of NPROD * 64 * TPT tuples, one workgroup per CU, 16 <= F <= FMAX buckets.
template <bool CLAIM, int FMAX, int BLOCK, int NPROD, int TPT>
__global__ __launch_bounds__(BLOCK) void k64_ring(PassSrc s, uint32_t shift, uint32_t fbits, uint32_t* out, CapOut co, ClaimOut cq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = Ring64Cfg<FMAX, BLOCK, NPROD>;
  constexpr int kNF = C::kNF, kProd = C::kProd;
  constexpr int TILE = kProd * TPT;
  constexpr uint32_t LB = 128;        // bytes per line
  constexpr int LPL = 8;              // lanes per line (16 B each)
  constexpr int LPI = kWave / LPL;    // lines per store instruction
  uint32_t* cl = (uint32_t*)(smem + C::kCl);
  uint2* hot = (uint2*)(smem + C::kHot);
  uint2* hot2 = (uint2*)(smem + C::kHot2);
  uint32_t* misc = (uint32_t*)(smem + C::kMisc);
  uint32_t* rsv = (uint32_t*)(smem + C::kRsv);
  uint32_t* want = (uint32_t*)(smem + C::kWant);
  char* rings = smem + C::kRing;
  const uint32_t F = 1u << fbits;
  const uint32_t rsh = (uint32_t)C::kRingLog2 - fbits;  // log2 of a ring's bytes (>= 8)
  const uint32_t RB = 1u << rsh, M = RB - 1u;
  const uint32_t nd = F >> 4;                           // dummy rings (F >= 16)
  const TileRange r = CLAIM ? block_tiles<SRC_COLS, TILE>(s) : block_tiles<SRC_TUPLES, TILE>(s);
  if (CLAIM && r.t0 >= r.t1) return;  // nothing to read: claim nothing (uniform)
  const unsigned lane = lane_id();
  const unsigned wave = threadIdx.x / kWave;
  const uint32_t cap = CLAIM ? cq.cap : co.cap;
  const uint32_t cap_l = cap / 32u * 3u;  // lines per region (the host keeps regions x cap_l below 2^32)
  unsigned long long* const overflow = CLAIM ? cq.overflow : co.overflow;
  for (uint32_t b = threadIdx.x; b < F + nd; b += BLOCK) cl[b] = 0;
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
    const uint32_t chk_bits = (uint32_t)__popc(s.chk_mask);      // (the mask is a run of low bits, or 0: every tuple in range is real)
    const uint32_t dummy_bs = (F + (lane & (nd - 1u))) << rsh;  // ring offset of this lane's dummy ring
    const uint32_t lim = RB - k64TupleBytes;                    // a tuple fits while the ring holds at most this many bytes
    using Regs = Regs64<TPT>;
    // bk[i] <- ring offset of the tuple's bucket, or of a dummy ring (ragged tails, fillers)
    auto hash_tile = [&](Regs& t) {
#pragma unroll
      for (int i = 0; i < TPT; ++i) {
        const uint32_t h = part_hash64(((uint64_t)t.hi[i] << 32) | t.lo[i]);
        bool ok = ((t.mask >> i) & 1u) != 0;
        if (!CLAIM) ok &= __builtin_amdgcn_ubfe(h, s.chk_shift, chk_bits) == expect;
        t.bk[i] = ok ? __builtin_amdgcn_ubfe(h, shift, fbits) << rsh : dummy_bs;
      }
    };
    Cols64Cursor cur;
    auto load = [&](Regs& t, uint64_t tile) {
      if (CLAIM) load_cols64<kProd, TPT>(s, tile, cur, t);
      else load_tuples12<kProd, TPT>(s, r, tile, t);
    };
    Regs A, B, Cc;
    A.mask = B.mask = Cc.mask = 0;
#pragma unroll
    for (int i = 0; i < TPT; ++i) A.bk[i] = B.bk[i] = Cc.bk[i] = dummy_bs;
    if (r.t0 < r.t1) load(A, r.t0);
    if (r.t0 + 1 < r.t1) load(B, r.t0 + 1);
    hash_tile(A);
    const uint32_t tb = threadIdx.x;  // CLAIM: thread t refills the reserve of bucket t
    uint32_t claimed = 0;             // the atomic's answer, in flight until the next tile's loads have been awaited
    bool pending = false;
    auto dword_of = [](const Regs& c, int i, int j) -> uint32_t { return j == 0 ? c.lo[i] : j == 1 ? c.hi[i] : c.p[i]; };
    auto step = [&](uint64_t tile, Regs& c, Regs& nx, Regs& n2) {
      // insert tile `tile`: one returning LDS atomic and three 4-byte LDS stores per tuple
      uint32_t v[TPT];
#pragma unroll
      for (int i = 0; i < TPT; ++i) v[i] = atomicAdd((uint32_t*)(smem + C::kCl + (c.bk[i] >> (rsh - 2u))), k64TupleBytes);
      uint32_t mx = 0;
#pragma unroll
      for (int i = 0; i < TPT; ++i) mx = max(mx, v[i] & 0xFFFFu);
      uint32_t dmask = 0;
      if (!__ballot(mx > lim)) {  // every tuple of this wave fits its ring (dummy rings are reset every tile): no selects
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          const uint32_t p = (v[i] >> 16) + (v[i] & 0xFFFFu), bs = c.bk[i];
          *(uint32_t*)(rings + (bs | (p & M))) = c.lo[i];
          *(uint32_t*)(rings + (bs | ((p + 4u) & M))) = c.hi[i];
          *(uint32_t*)(rings + (bs | ((p + 8u) & M))) = c.p[i];
        }
      } else {  // a ring of this wave overflows (or a dummy ring ran over in an edge tile): dword by dword
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          const uint32_t cnt = v[i] & 0xFFFFu, hd = v[i] >> 16, bs = c.bk[i];
          const bool valid = bs < (F << rsh);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const uint32_t q = cnt + 4u * j;
            // what lies beyond the ring is placed after the flush; a dummy ring just wraps
            const uint32_t at = (!valid | (q < RB)) ? (bs | ((hd + q) & M)) : (dummy_bs | (4u * j));
            *(uint32_t*)(rings + at) = dword_of(c, i, j);
          }
          dmask |= (valid & (cnt > lim)) ? 1u << i : 0u;
        }
        if (dmask) misc[0] = (uint32_t)tile + 1u;
      }
      __syncthreads();  // ---- barrier 1: the flushers take over the rings
      // tile + 1 is awaited and hashed HERE, beside the flush, and tile + 2 is requested only after tile + 1 has arrived
      // (ring_scatter.h)
      nx.pin();
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
          claimed = atomicAdd(&cq.cursor[(uint64_t)r.seg * F + tb], k64Allot);
          want[tb] = 2u;
          pending = true;
        }
      }
      n2.mask = 0;
      if (tile + 2 < r.t1) load(n2, tile + 2);
      __syncthreads();  // ---- barrier 2: complete lines are out, the rings have room again
      if (misc[0] == (uint32_t)tile + 1u) {  // some bucket was hot in this tile (uniform)
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
          if (dmask & (1u << i)) {
            const uint32_t bs = c.bk[i], b = bs >> rsh;
            const uint2 info = hot[b], info2 = hot2[b];
            const uint32_t full = info.y & 0xFFFFu, hd = info.y >> 16, cnt = v[i] & 0xFFFFu;
            const uint64_t region = ((uint64_t)r.seg * F + b) * cap_l * LB;  // byte offset of the bucket's region
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const uint32_t q = cnt + 4u * j;
              if (q < RB) continue;  // already in the ring (flushed with it)
              const uint32_t val = dword_of(c, i, j);
              if (q < full) {  // straight to the range(s) the owner reserved (dropped when the region is full)
                const uint32_t d = q - RB;
                const uint32_t base = d < info2.x ? info.x : info2.y, off = d < info2.x ? d : d - info2.x;
                if (base != kRingNone) gstore_t(out + (region + base + off) / 4u, val);
              } else {         // tail of the tile: the bucket's new residue
                *(uint32_t*)(rings + (bs | ((hd + (q - full)) & M))) = val;
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
    const uint32_t p_own = r.seg * F + b_own;
    const uint32_t line0 = p_own * cap_l;                    // the region's first line
    const uint32_t ring_own = C::kRing + (b_own << rsh);     // LDS address of the bucket's ring
    uint2* list = (uint2*)(smem + C::kList) + fw * C::kListPerWave;
    uint32_t pos = 0, rem = 0;  // next line of the region; CLAIM: lines left in the allotment in use
    bool dead = !owner;         // region exhausted (skew): tuples are dropped, *overflow raised
    uint32_t fil0 = 0u, fil1 = 0u;  // CLAIM: the key of a tuple of another bucket (filler tuples: {key, payload 0})
    auto line_of = [](uint32_t allot) { return allot / 32u * 3u; };  // first line of the allotment that starts at tuple `allot`
    if (CLAIM && owner) {
      uint64_t fk = 0;
      while (__builtin_amdgcn_ubfe(part_hash64(fk), shift, fbits) == b_own) ++fk;
      fil0 = (uint32_t)fk;
      fil1 = (uint32_t)(fk >> 32);
      const uint32_t got = atomicAdd(&cq.cursor[p_own], 2 * k64Allot);
      if ((uint64_t)got + 2 * k64Allot > cap) {
        dead = true;
        atomicAdd(overflow, 1ull);
      } else {
        pos = line_of(got);
        rem = k64AllotLines;
        rsv[b_own] = got + k64Allot;
      }
    }
    // CLAIM: the next allotment -- the reserve (a producer refills it), or claimed on the spot if that has not arrived
    auto next_allotment = [&]() {
      uint32_t a = rsv[b_own];
      if (a != kRingNone) {
        rsv[b_own] = kRingNone;
        want[b_own] = 1u;
      } else {
        a = atomicAdd(&cq.cursor[p_own], k64Allot);
      }
      if ((uint64_t)a + k64Allot > cap) {
        dead = true;
        atomicAdd(overflow, 1ull);
      } else {
        pos = line_of(a);
        rem = k64AllotLines;
      }
    };
    // The wave emits the listed lines eight per store instruction, kFU instructions' worth at a time (ring_claim.h)
    constexpr int kFU = 2;
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
    for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
      __syncthreads();  // ---- barrier 1
      const uint32_t vv = owner ? cl[b_own] : 0u;
      if (fw == 0 && lane < nd) cl[F + lane] = 0;  // the dummy rings start every tile empty
      const uint32_t total = vv & 0xFFFFu, hd = vv >> 16;
      const bool hotb = total > RB;
      const uint32_t placed = hotb ? RB : total;
      const uint32_t nl = placed / LB;  // complete lines in the ring
      if (!CLAIM && owner && !dead && pos + nl > cap_l) {
        dead = true;
        atomicAdd(overflow, 1ull);
      }
      // list them in rounds (round q: line q of the owners that have one), every line at the next line of its bucket's
      // allotment / region; then emit
      uint32_t n = 0;
      for (uint32_t q = 0;; ++q) {
        bool has = q < nl && !dead;
        if (CLAIM && has) {
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
              make_uint2(line0 + pos, ring_own + ((hd + q * LB) & M));
          ++pos;
          if (CLAIM) --rem;
        }
        n += (uint32_t)__popcll(m);
      }
      // CLAIM: an allotment that has just run out is replaced NOW from the reserve, beside the emission (ring_claim.h)
      if (CLAIM && owner && !dead && rem == 0) {
        const uint32_t a = rsv[b_own];
        if (a != kRingNone) {
          rsv[b_own] = kRingNone;
          want[b_own] = 1u;
          if ((uint64_t)a + k64Allot > cap) {
            dead = true;
            atomicAdd(overflow, 1ull);
          } else {
            pos = line_of(a);
            rem = k64AllotLines;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      emit_lines(n);
      __builtin_amdgcn_wave_barrier();
      if (owner) {
        if (hotb) {  // the bytes beyond the ring: whole lines' worth goes out directly, the rest is the new residue
          const uint32_t full = total & ~(LB - 1);
          const uint32_t ndl = (full - RB) / LB;  // lines that go out directly
          uint32_t base1 = kRingNone, len1 = 0, base2 = kRingNone;
          if (ndl && !dead) {
            if (CLAIM) {
              // The bucket's stream goes on without a gap: what is left of the allotment in use, then the next one(s)
              const uint32_t take1 = min(rem, ndl), rest = ndl - take1;
              if (take1) {
                base1 = pos * LB;
                len1 = take1 * LB;
                pos += take1;
                rem -= take1;
              }
              if (rest) {
                if (rest <= k64AllotLines) {
                  next_allotment();
                  if (!dead) {
                    base2 = pos * LB;
                    pos += rest;
                    rem -= rest;
                  }
                } else {  // a very hot bucket: a block of whole allotments, the stream ends inside its last one
                  const uint32_t blk = (rest + k64AllotLines - 1) / k64AllotLines;
                  const uint32_t got = atomicAdd(&cq.cursor[p_own], blk * k64Allot);
                  if ((uint64_t)got + (uint64_t)blk * k64Allot > cap) {
                    dead = true;
                    atomicAdd(overflow, 1ull);
                  } else {
                    base2 = line_of(got) * LB;
                    pos = line_of(got) + rest;
                    rem = blk * k64AllotLines - rest;
                  }
                }
              }
              if (dead) base1 = base2 = kRingNone;
            } else if (pos + ndl > cap_l) {
              dead = true;
              atomicAdd(overflow, 1ull);
            } else {
              base1 = pos * LB;
              len1 = ndl * LB;
              pos += ndl;
            }
          }
          hot[b_own] = make_uint2(base1, full | (hd << 16));
          hot2[b_own] = make_uint2(len1, base2);
          cl[b_own] = (total - full) | (hd << 16);
        } else {
          cl[b_own] = (placed - nl * LB) | (((hd + nl * LB) & M) << 16);
        }
      }
      __syncthreads();  // ---- barrier 2
      if (misc[0] == (uint32_t)tile + 1u) __syncthreads();  // ---- barrier 3 (hot tiles only)
    }
    if (CLAIM) __syncthreads();  // the producers have delivered the claims that were in flight
    // tail: what is left in the rings (less than a line per bucket)
    if (owner) {
      const uint32_t vv = cl[b_own];
      const uint32_t rr = vv & 0xFFFFu, hd = vv >> 16;  // bytes
      if (CLAIM) {
        if (!dead && rr && rem == 0) next_allotment();
        if (!dead) {  // the residue, then fillers up to the end of the allotment (the stream ends on a tuple boundary)
          uint32_t* dst = out + (uint64_t)(line0 + pos) * (LB / 4);
          const uint32_t nres = rr / 4u, ntot = rem * (LB / 4);
          uint32_t ph = 0;
          for (uint32_t j = 0; j < ntot; ++j) {
            uint32_t val;
            if (j < nres) {
              val = *(const uint32_t*)(smem + ring_own + ((hd + 4u * j) & M));
            } else {
              val = ph == 0u ? fil0 : ph == 1u ? fil1 : 0u;
              ph = ph == 2u ? 0u : ph + 1u;
            }
            dst[j] = val;
          }
        }
        const uint32_t ahead = rsv[b_own];  // the allotment claimed ahead holds only fillers
        if (ahead != kRingNone && (uint64_t)ahead + k64Allot <= cap) {
          uint32_t* dst = out + (uint64_t)(line0 + line_of(ahead)) * (LB / 4);
          uint32_t ph = 0;
          for (uint32_t j = 0; j < k64AllotLines * (LB / 4); ++j) {
            dst[j] = ph == 0u ? fil0 : ph == 1u ? fil1 : 0u;
            ph = ph == 2u ? 0u : ph + 1u;
          }
        }
      } else {
        uint32_t count_b = pos * LB;
        if (!dead) {
          if (rr && pos >= cap_l) {
            atomicAdd(overflow, 1ull);
          } else {
            uint32_t* dst = out + (uint64_t)(line0 + pos) * (LB / 4);
            for (uint32_t j = 0; j < rr; j += 4) dst[j / 4] = *(const uint32_t*)(smem + ring_own + ((hd + j) & M));
            count_b += rr;
          }
        }
        ((uint32_t*)hot)[b_own] = count_b / k64TupleBytes;  // tuples, for the closing scan
      }
    }
  }
  if (!CLAIM) {  // partition sizes and dense starts (exclusive scan over this segment's buckets, one per thread)
    __syncthreads();
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
      co.cnt[(uint64_t)r.seg * F + t] = (uint32_t)count;
      co.dense[(uint64_t)r.seg * F + t] = pre;  // inside the segment: k_dense_fix adds the segments' bases
    }
    if (threadIdx.x == 0) {
      uint64_t total = 0;
      for (int w = 0; w < BLOCK / kWave; ++w) total += wtot[w];
      co.seg_total[r.seg] = total;
    }
  }
}

// ------------------------------------------------------------------ build + probe over the capacity layout
// k64_probe (radix64.h) over partitions of 12-byte tuples: partition p = tuples [p * cap, p * cap + cnt[p]) of its side,
// result rows at dense[p] + i (left side).  Same table: 4096 slots, 64-bit keys, open addressing from slot = ht_hash64(key) &
// (slots - 1) (hashtable.c:99-101), an equal key overwrites (hashtable.c:133).
//   W   tuples of a thread whose probing sequences advance TOGETHER: one LDS round trip per step for all of them instead of one
//       divergent loop per tuple after the other (1 | 2; three and more spill at six waves per SIMD -- profiles/r3_probe64_ab.md)
//   DH  double hashing: a key steps through the table by an odd number taken from its hash bits above the slot bits, so keys
//       that meet part again.  The slowest lane of a wave sets the pace of its loop, and linear probing's clusters make that
//       lane's sequence long.  (The table is internal: the rows do not depend on where a key sits.)
struct Part12 {
  const Tuple12* t;
  const uint32_t* cnt;
  const uint64_t* dense;
  uint32_t cap;
};

template <int W, bool DH>
__global__ __launch_bounds__(k64Block) __attribute__((amdgpu_waves_per_eu(6, 6))) void k64_probe12(Part12 RV, Part12 LV, uint32_t nparts, uint64_t* out_fk, uint32_t* out_y, uint32_t* out_x,
                                                         unsigned long long* counters, uint32_t* miss_bits, ProbeSlices slices) {
  __shared__ unsigned long long s_keys[k64Slots];
  __shared__ uint32_t s_vals[k64Slots];
  __shared__ uint32_t s_sp[2];  // the one key equal to the empty marker: {present, value}
  uint32_t my_miss = 0, my_dup = 0;
  uint64_t my_max = 0;
  // A partition is small (2048 + 2048 tuples on average): its three phases -- clear, build, probe -- are latency chains, so the
  // loads run AHEAD of them: the build tuples of the next work item are requested before the probe of the current one, its
  // sizes one phase earlier still, and the first probe tuples before the barrier that completes the table.
  constexpr int kU = 6;                              // tuples per thread and chunk: one chunk holds any build side the table admits
  constexpr uint32_t kChunk = kU * k64Block;
  static_assert(kChunk >= k64TableCap, "a build side is one chunk");
  static_assert(kU % W == 0, "whole groups");
  const uint32_t n_extra = slices.count ? min(*slices.count, slices.capacity) : 0u;
  const uint32_t total = nparts + n_extra;           // work items: the partitions, then the (partition, slice >= 1) pairs
  struct Item {
    uint32_t p, slice, n_r, n_l;
    uint64_t dense;
  };
  auto fetch = [&](uint32_t w) -> Item {
    Item it{0u, 0u, 0u, 0u, 0ull};
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
      it.dense = LV.dense[it.p];
    }
    return it;
  };
  // probe rows [lb, le) of the item's partition; nothing to do: lb >= le
  auto range = [&](const Item& it, uint32_t& lb, uint32_t& le) {
    lb = it.slice * kProbeSlice;
    le = slices.count ? min(it.n_l, lb + kProbeSlice) : it.n_l;
    if (it.n_r > k64TableCap) le = lb;  // (reported where the item is handled)
  };
  auto load_chunk = [&](const Tuple12* base, uint32_t lo, uint32_t hi, Tuple12 (&v)[kU]) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const uint32_t i = lo + u * k64Block + threadIdx.x;
      v[u] = Tuple12{0u, 0u, 0u};
      if (i < hi) v[u] = gload_tuple12(base + i);
    }
  };
  auto key_of = [](const Tuple12& t) { return ((unsigned long long)t.hi << 32) | t.lo; };
  // where a key's sequence starts and how far it steps
  auto seq_of = [](unsigned long long k, uint32_t& slot, uint32_t& step) {
    const uint32_t h = ht_hash64(k);
    slot = h & (k64Slots - 1);  // hashtable.c:99-101
    step = DH ? ((h >> 12) | 1u) & (k64Slots - 1) : 1u;
  };
  Tuple12 Rv[kU], Lv[kU];
  Item cur = fetch(blockIdx.x);
  {
    uint32_t lb, le;
    range(cur, lb, le);
    if (lb < le) load_chunk(RV.t + (uint64_t)cur.p * RV.cap, 0u, cur.n_r, Rv);
  }
  for (uint32_t w = blockIdx.x; w < total; w += gridDim.x) {
    const Item nxt = fetch(w + gridDim.x);  // (its sizes arrive while this item's table is built)
    uint32_t lb, le;
    range(cur, lb, le);
    if (cur.n_r > my_max) my_max = cur.n_r;
    if (cur.n_r > k64TableCap && cur.n_l && threadIdx.x == 0 && cur.slice == 0) atomicAdd(&counters[2], 1ull);
    const bool work = lb < le;
    const Tuple12* L = LV.t + (uint64_t)cur.p * LV.cap;
    if (work) {
      for (uint32_t i = threadIdx.x; i < k64Slots; i += k64Block) s_keys[i] = k64Empty;
      if (threadIdx.x == 0) s_sp[0] = 0;
      __syncthreads();
#pragma unroll
      for (int g = 0; g < kU; g += W) {  // build: table[pk] = x
        uint32_t slot[W], step[W], act = 0;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const unsigned long long k = key_of(Rv[g + j]);
          seq_of(k, slot[j], step[j]);
          if ((g + j) * k64Block + threadIdx.x >= cur.n_r) continue;
          if (k == k64Empty) {
            s_sp[1] = Rv[g + j].pay;
            if (atomicExch(&s_sp[0], 1u) && cur.slice == 0) ++my_dup;
          } else {
            act |= 1u << j;
          }
        }
        while (act) {  // (ends: a table admits at most 3/4 of its slots)
          unsigned long long old[W];
#pragma unroll
          for (int j = 0; j < W; ++j)
            if (act >> j & 1u) old[j] = atomicCAS(&s_keys[slot[j]], k64Empty, key_of(Rv[g + j]));
#pragma unroll
          for (int j = 0; j < W; ++j) {
            if (!(act >> j & 1u)) continue;
            const unsigned long long k = key_of(Rv[g + j]);
            if (old[j] == k64Empty || old[j] == k) {  // free cell, or an equal key overwrites (hashtable.c:133)
              s_vals[slot[j]] = Rv[g + j].pay;
              if (old[j] == k && cur.slice == 0) ++my_dup;
              act &= ~(1u << j);
            } else {
              slot[j] = (slot[j] + step[j]) & (k64Slots - 1);
            }
          }
        }
      }
      load_chunk(L, lb, le, Lv);
      __syncthreads();
    }
    {  // the next item's build side
      uint32_t nlb, nle;
      range(nxt, nlb, nle);
      if (nlb < nle) load_chunk(RV.t + (uint64_t)nxt.p * RV.cap, 0u, nxt.n_r, Rv);
    }
    if (work) {
      for (uint32_t base = lb; base < le; base += kChunk) {  // probe + take of the right payload
        if (base != lb) load_chunk(L, base, le, Lv);
#pragma unroll
        for (int g = 0; g < kU; g += W) {
          uint32_t slot[W], step[W], xs[W], act = 0, hit = 0;
#pragma unroll
          for (int j = 0; j < W; ++j) {
            const unsigned long long k = key_of(Lv[g + j]);
            seq_of(k, slot[j], step[j]);
            xs[j] = 0;
            if (base + (g + j) * k64Block + threadIdx.x >= le) continue;
            if (k == k64Empty) {
              if (s_sp[0] != 0) {
                hit |= 1u << j;
                xs[j] = s_sp[1];
              }
            } else {
              act |= 1u << j;
            }
          }
          uint32_t found = 0;
          while (act) {
            unsigned long long kk[W];
#pragma unroll
            for (int j = 0; j < W; ++j)
              if (act >> j & 1u) kk[j] = s_keys[slot[j]];
#pragma unroll
            for (int j = 0; j < W; ++j) {
              if (!(act >> j & 1u)) continue;
              if (kk[j] == key_of(Lv[g + j])) {
                found |= 1u << j;
                act &= ~(1u << j);
              } else if (kk[j] == k64Empty) {
                act &= ~(1u << j);
              } else {
                slot[j] = (slot[j] + step[j]) & (k64Slots - 1);
              }
            }
          }
#pragma unroll
          for (int j = 0; j < W; ++j)
            if (found >> j & 1u) xs[j] = s_vals[slot[j]];
          hit |= found;
#pragma unroll
          for (int j = 0; j < W; ++j) {
            const uint32_t i = base + (g + j) * k64Block + threadIdx.x;
            if (i >= le) continue;
            const uint64_t o = cur.dense + i;
            if (!(hit >> j & 1u)) {
              ++my_miss;
              if (miss_bits) atomicOr(&miss_bits[o >> 5], 1u << (o & 31));
            }
            out_fk[o] = key_of(Lv[g + j]);
            out_y[o] = Lv[g + j].pay;
            out_x[o] = xs[j];
          }
        }
      }
      __syncthreads();
    }
    cur = nxt;
  }
  const uint64_t wm = wave_reduce_add((uint64_t)my_miss), wd = wave_reduce_add((uint64_t)my_dup);
  if (lane_id() == 0) {
    if (wm) atomicAdd(&counters[0], (unsigned long long)wm);
    if (wd) atomicAdd(&counters[1], (unsigned long long)wd);
  }
  if (threadIdx.x == 0) atomicMax(&counters[3], (unsigned long long)my_max);
}

}  // namespace olap
