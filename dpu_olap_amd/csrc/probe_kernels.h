// probe_kernels.h -- build + probe of one partition pair in LDS, and the row compaction of
// the inner-join mode (included by join.hip only).
//
// Reference behaviour: dpu/shared/kernels/hash_build.c:9-35, hash_probe.c:9-46,
// hashtable.c:89-192 (open addressing, linear probing, equal key overwrites), fused with the
// take of the right payload (host/join/join_dpu.cc:332-362).
#pragma once
#include "radix_kernels.h"

namespace olap {

// ------------------------------------------------------------ the probing table's sequences
// Open addressing in an LDS table of SLOTS cells.  A key's sequence starts at the reference's cell (low hash bits,
// hashtable.c:99-101) and steps by an ODD number of its own (double hashing; the reference probes linearly, hashtable.c:120-141):
// the slowest lane of a wave sets the pace of its loop, and linear probing's clusters make that lane's sequence long.  The table
// is internal -- which cell a key sits in never shows in the rows.  The loops run under an active flag, without a step bound
// (a table admits at most 3/4 of its cells) and without a break: the bounded `for (...) { ...; break; }` form of rounds 1-2
// compiled to far slower code (profiles/r3_probe64_ab.md).
template <uint32_t SLOTS>
__device__ __forceinline__ void seq_of_hash(uint32_t h, uint32_t& slot, uint32_t& step) {
  slot = h & (SLOTS - 1);
  step = ((h * 0x9E3779B1u) >> (32 - __builtin_ctz(SLOTS))) | 1u;  // (partition-mates share the hash's TOP bits: mix the low ones up)
}
// table[k] = val for a valid tuple; true: the key was there already (it is overwritten, hashtable.c:133)
template <uint32_t SLOTS, typename K>
__device__ __forceinline__ bool table_put(K* keys, uint32_t* vals, K empty, K k, uint32_t h, uint32_t val, bool valid) {
  uint32_t slot, step;
  seq_of_hash<SLOTS>(h, slot, step);
  bool act = valid, dup = false;
  while (act) {
    const K old = atomicCAS(&keys[slot], empty, k);
    if (old == empty || old == k) {
      vals[slot] = val;
      dup = old == k;
      act = false;
    } else {
      slot = (slot + step) & (SLOTS - 1);
    }
  }
  return dup;
}
template <uint32_t SLOTS, typename K>
__device__ __forceinline__ bool table_get(const K* keys, const uint32_t* vals, K empty, K k, uint32_t h, bool valid, uint32_t& x) {
  uint32_t slot, step;
  seq_of_hash<SLOTS>(h, slot, step);
  bool act = valid, found = false;
  while (act) {
    const K kk = keys[slot];
    if (kk == k) {
      found = true;
      act = false;
    } else if (kk == empty) {
      act = false;
    } else {
      slot = (slot + step) & (SLOTS - 1);
    }
  }
  if (found) x = vals[slot];
  return found;
}

// ------------------------------------------------------------ build + probe
// counters: [0] missing keys, [1] duplicate build keys, [2] overflowing
// partitions, [3] max build rows, [4] region overflows (radix passes), [5] tuples whose
// leading (skipped) hash bits differ from their partition's (direct table only)
// A partitioned side: exact layout (cap == 0: partition p = [off[p], off[p+1])) or
// capacity layout (partition p = [p * cap, p * cap + cnt[p]), dense start dense[p]).
struct PartView {
  const uint2* t;
  const uint64_t* off;
  const uint32_t* cnt;
  const uint64_t* dense;
  uint32_t cap;
  const uint32_t* top;  // packed side (pack6.h: t = 6-byte tuples, capacity layout): per partition, the hash bits above the table bits
  // packed side written chunk by chunk from pre-split input (PassSrc::out_by_block): partition p = (segment j, bucket b) of a
  // last pass of 2^sub_fbits buckets is the union of nsub sub-partitions, number ((j * nsub + q) << sub_fbits) | b for chunk q of
  // its segment; cnt / dense / top are indexed by the sub-partition.  nsub <= 1: plain partitions.
  uint32_t nsub;
  uint32_t sub_fbits;
};

// A probe partition far larger than the others (one hot foreign key) would be streamed by ONE workgroup while the rest of
// the chip idles (2^27 left rows, half of them one key: 52 ms of probe).  The probe side of a partition is therefore cut
// into slices of kProbeSlice rows: slice 0 is handled where the partition is, every further slice is a work item of its own
// -- (partition, slice) pairs listed by k_probe_slices, taken up by the probe kernels' workgroups after their partitions;
// each rebuilds the partition's table (the build side is not the skewed one) and probes its slice.  Uniform data has no
// partition beyond one slice: the list stays empty.
constexpr uint32_t kProbeSlice = 1u << 16;
struct ProbeSlices {
  uint2* list;          // (partition, slice >= 1)
  uint32_t* count;      // entries wanted (may exceed capacity: clamp)
  uint32_t capacity;
};

__global__ __launch_bounds__(256) void k_probe_slices(PartView LV, uint32_t nparts, ProbeSlices ps) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= nparts) return;
  const uint64_t n_l = LV.cap ? (uint64_t)LV.cnt[p] : LV.off[p + 1] - LV.off[p];
  const uint64_t ns = (n_l + kProbeSlice - 1) / kProbeSlice;
  for (uint64_t sl = 1; sl < ns; ++sl) {
    const uint32_t at = atomicAdd(ps.count, 1u);
    if (at < ps.capacity) ps.list[at] = make_uint2(p, (uint32_t)sl);
  }
}

// SLOTS: 8192 (64 KiB: two workgroups per CU) | 4096 (32 KiB: four per CU -- the kernel is bound by the waves it has in flight,
// not by the table's load; for partitions of <= 2048 build rows on average, join.hip probe_slots).
template <uint32_t SLOTS>
__global__ __launch_bounds__(kJBlock, SLOTS == 4096 ? 8 : 4) void k_join_probe(PartView RV, PartView LV, uint32_t nparts,
                                                           uint32_t* out_fk, uint32_t* out_y,
                                                           uint32_t* out_x, uint32_t pair_store_ok,
                                                           unsigned long long* counters, uint32_t* miss_bits, ProbeSlices slices) {
  const uint2* R = RV.t;
  const uint2* L = LV.t;
  __shared__ uint32_t s_keys[SLOTS];
  __shared__ uint32_t s_vals[SLOTS];
  __shared__ uint32_t s_sp[2];  // the one key equal to the empty marker: {present, value}
  uint32_t my_miss = 0, my_dup = 0;
  uint64_t my_max = 0;
  auto body = [&](const uint32_t p, const uint32_t slice) {
    const uint64_t rlo = RV.cap ? (uint64_t)p * RV.cap : RV.off[p];
    const uint64_t rhi = RV.cap ? rlo + RV.cnt[p] : RV.off[p + 1];
    const uint64_t llo = LV.cap ? (uint64_t)p * LV.cap : LV.off[p];
    const uint64_t lhi = LV.cap ? llo + LV.cnt[p] : LV.off[p + 1];
    const uint64_t n_r = rhi - rlo, n_l = lhi - llo;
    // output rows are dense: tuple at sparse index i goes to row i + delta
    const uint64_t delta = (LV.cap ? LV.dense[p] : llo) - llo;
    // 8-byte pair stores need an 8-byte aligned element: pair_store_ok 1 = the output columns start 8-byte
    // aligned, 2 = they all start at 4 mod 8 (a slice of a larger result), 0 = mixed -> 4-byte stores
    const uint32_t pair_store = pair_store_ok && ((delta + (pair_store_ok >> 1)) & 1) == 0;
    if (n_r > my_max) my_max = n_r;
    if (n_l == 0) return;
    if (n_r > (SLOTS / 4 * 3)) {
      if (threadIdx.x == 0 && slice == 0) atomicAdd(&counters[2], 1ull);
      return;
    }
    // this work item's share of the probe side (slice 0 with the partition, the others from the slice list)
    const uint64_t lb = llo + (uint64_t)slice * kProbeSlice, le = slices.count ? min(lhi, lb + kProbeSlice) : lhi;  // (no slice list: all of it)
    if (lb >= le) return;
    for (uint32_t i = threadIdx.x; i < SLOTS; i += kJBlock) s_keys[i] = kEmpty;
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
              if (atomicExch(&s_sp[0], 1u) && slice == 0) ++my_dup;
            } else if (table_put<SLOTS>(s_keys, s_vals, kEmpty, k, wang_hash(k), val, true) && slice == 0) {
              ++my_dup;
            }
          }
        }
      }
    }
    __syncthreads();
    // probe (hash_probe.c:26-36) fused with the take of the right payload
    {
      const uint64_t a = lb & ~1ull;
      const uint64_t span = le - a;
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
            ok[h] = i0 + h >= lb && i0 + h < le;
            if (!ok[h]) continue;
            bool hit = false;
            if (k == kEmpty) {
              hit = s_sp[0] != 0;
              xs[h] = s_sp[1];
            } else {
              hit = table_get<SLOTS>(s_keys, s_vals, kEmpty, k, wang_hash(k), true, xs[h]);
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
  };
  for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) body(p, 0u);
  const uint32_t n_extra = slices.count ? min(*slices.count, slices.capacity) : 0u;
  for (uint32_t w = blockIdx.x; w < n_extra; w += gridDim.x) body(slices.list[w].x, slices.list[w].y);
  const uint64_t wm = wave_reduce_add((uint64_t)my_miss);
  const uint64_t wd = wave_reduce_add((uint64_t)my_dup);
  if (lane_id() == 0) {
    if (wm) atomicAdd(&counters[0], (unsigned long long)wm);
    if (wd) atomicAdd(&counters[1], (unsigned long long)wd);
  }
  if (threadIdx.x == 0) atomicMax(&counters[3], (unsigned long long)my_max);
}

// ------------------------------------------------------- small build side: no partitioning
// A right table of <= kTableCap rows fits ONE LDS table: every workgroup builds the whole table from the right
// COLUMNS and streams its share of the LEFT COLUMNS through it -- no radix pass, no tuple buffers, and the result
// keeps the left row order.  20 B per left row (8 read + 12 written) instead of 60 through one radix pass; and
// the chip stays busy, where 2^1 partitions of a one-bit plan would leave the whole probe side to two workgroups
// (2^28 x 2^12 rows: 197 ms partitioned by one bit).  The reference always partitions
// (host/join/join_dpu.cc:82-142); Arrow's hash join -- the Native path -- has this shape.  Table semantics as
// k_join_probe (hashtable.c:89-192: first cell = low hash bits, an equal key overwrites; table_put / table_get above).
constexpr int kBcTile = kJBlock * 8;  // rows per tile of the column views (8 per thread)

__global__ __launch_bounds__(kJBlock, 4) void k_join_broadcast(PassSrc rs, PassSrc ls, uint32_t* out_fk, uint32_t* out_y,
                                                               uint32_t* out_x, unsigned long long* counters,
                                                               uint32_t* miss_bits) {
  constexpr int TPT = kBcTile / kJBlock;
  __shared__ uint32_t s_keys[kSlots];
  __shared__ uint32_t s_vals[kSlots];
  __shared__ uint32_t s_sp[2];  // the one key equal to the empty marker: {present, value}
  uint32_t my_miss = 0, my_dup = 0;
  for (uint32_t i = threadIdx.x; i < kSlots; i += kJBlock) s_keys[i] = kEmpty;
  if (threadIdx.x == 0) s_sp[0] = 0;
  __syncthreads();
  {  // build: the whole right table (every workgroup; the host has checked rows <= kTableCap)
    ColsCursor cur;
    const TileRange none{};
    for (uint64_t tile = 0; tile < rs.v.total_units; ++tile) {
      uint32_t key[TPT], pay[TPT];
      const uint32_t mask = load_tile<SRC_COLS, kJBlock, true, kBcTile>(rs, none, tile, cur, key, pay);
#pragma unroll
      for (int i = 0; i < TPT; ++i) {
        if (!(mask & (1u << i))) continue;
        const uint32_t k = key[i];
        if (k == kEmpty) {
          s_sp[1] = pay[i];
          if (atomicExch(&s_sp[0], 1u)) ++my_dup;
          continue;
        }
        if (table_put<kSlots>(s_keys, s_vals, kEmpty, k, wang_hash(k), pay[i], true)) ++my_dup;
      }
    }
  }
  __syncthreads();
  {  // probe: this workgroup's run of left tiles, results at the rows' own positions
    ColsCursor cur;
    const TileRange r = block_tiles<SRC_COLS, kBcTile>(ls);
    for (uint64_t tile = r.t0; tile < r.t1; ++tile) {
      uint32_t key[TPT], pay[TPT], xs[TPT];
      const uint32_t mask = load_tile<SRC_COLS, kJBlock, true, kBcTile>(ls, r, tile, cur, key, pay);
      const uint64_t row0 = cur.row0 + cur.local_unit * (uint64_t)kBcTile;  // global row of the tile's first element
#pragma unroll
      for (int i = 0; i < TPT; ++i) {
        xs[i] = 0;
        if (!(mask & (1u << i))) continue;
        const uint32_t k = key[i];
        bool hit = false;
        if (k == kEmpty) {
          hit = s_sp[0] != 0;
          xs[i] = s_sp[1];
        } else {
          hit = table_get<kSlots>(s_keys, s_vals, kEmpty, k, wang_hash(k), true, xs[i]);
        }
        if (!hit) {
          ++my_miss;
          const uint64_t row = row0 + (uint64_t)((i >> 2) * kJBlock + threadIdx.x) * 4 + (i & 3);
          if (miss_bits) atomicOr(&miss_bits[row >> 5], 1u << (row & 31));  // drop mode: mark the row
        }
      }
      const bool vec = mask == (1u << TPT) - 1u && ((((uintptr_t)(out_fk + row0)) | ((uintptr_t)(out_y + row0)) | ((uintptr_t)(out_x + row0))) & 15) == 0;
#pragma unroll
      for (int j = 0; j < TPT / 4; ++j) {
        const uint64_t o = row0 + (uint64_t)(j * kJBlock + threadIdx.x) * 4;
        if (vec) {  // (uniform: a whole tile whose rows start 16-byte aligned in all three columns)
          gstore4(out_fk + o, make_uint4(key[4 * j], key[4 * j + 1], key[4 * j + 2], key[4 * j + 3]));
          gstore4(out_y + o, make_uint4(pay[4 * j], pay[4 * j + 1], pay[4 * j + 2], pay[4 * j + 3]));
          gstore4(out_x + o, make_uint4(xs[4 * j], xs[4 * j + 1], xs[4 * j + 2], xs[4 * j + 3]));
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (!(mask & (1u << (4 * j + k)))) continue;
            out_fk[o + k] = key[4 * j + k];
            out_y[o + k] = pay[4 * j + k];
            out_x[o + k] = xs[4 * j + k];
          }
        }
      }
    }
  }
  const uint64_t wm = wave_reduce_add((uint64_t)my_miss);
  const uint64_t wd = wave_reduce_add((uint64_t)(blockIdx.x == 0 ? my_dup : 0u));  // every workgroup saw the same build side
  if (lane_id() == 0) {
    if (wm) atomicAdd(&counters[0], (unsigned long long)wm);
    if (wd) atomicAdd(&counters[1], (unsigned long long)wd);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&counters[3], (unsigned long long)rs.v.total_rows);
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
// skip_bits > 0 (olap_join_set_hash_skip_bits: an upstream owner split fixed the leading hash bits): the slot of a
// key is only its identity if every tuple of the partition shares those leading bits.  That precondition is CHECKED:
// tuples whose leading bits differ from the partition's first build tuple are counted in counters[5] and the join
// fails with OLAP_ERR_INVALID instead of returning the x of another key.
__global__ __launch_bounds__(kDBlock, 4) void k_join_probe_direct(PartView RV, PartView LV, uint32_t nparts, uint32_t run,
                                                               uint32_t table_bits, uint32_t skip_bits, uint32_t* out_fk,
                                                               uint32_t* out_y, uint32_t* out_x,
                                                               uint32_t pair_store_ok, unsigned long long* counters,
                                                               uint32_t* miss_bits, ProbeSlices slices) {
  __shared__ uint32_t s_vals[kDSlots];
  __shared__ uint32_t s_used[kDSlots / 32];
  const uint2* R = RV.t;
  const uint2* L = LV.t;
  const uint32_t slots = 1u << table_bits, smask = slots - 1;
  uint32_t my_miss = 0, my_dup = 0, my_foreign = 0;
  uint64_t my_max = 0;
  const uint32_t top_shift = skip_bits ? 32u - skip_bits : 0u;
  auto body = [&](const uint32_t p, const uint32_t slice) {
    const uint64_t rlo = RV.cap ? (uint64_t)p * RV.cap : RV.off[p];
    const uint64_t rhi = RV.cap ? rlo + RV.cnt[p] : RV.off[p + 1];
    const uint64_t llo = LV.cap ? (uint64_t)p * LV.cap : LV.off[p];
    const uint64_t lhi = LV.cap ? llo + LV.cnt[p] : LV.off[p + 1];
    const uint64_t n_r = rhi - rlo, n_l = lhi - llo;
    const uint64_t delta = (LV.cap ? LV.dense[p] : llo) - llo;
    // 8-byte pair stores need an 8-byte aligned element: pair_store_ok 1 = the output columns start 8-byte
    // aligned, 2 = they all start at 4 mod 8 (a slice of a larger result), 0 = mixed -> 4-byte stores
    const uint32_t pair_store = pair_store_ok && ((delta + (pair_store_ok >> 1)) & 1) == 0;
    if (n_r > my_max) my_max = n_r;
    if (n_l == 0) return;
    // this work item's share of the probe side (slice 0 with the partition, the others from the slice list)
    const uint64_t lb = llo + (uint64_t)slice * kProbeSlice, le = slices.count ? min(lhi, lb + kProbeSlice) : lhi;  // (no slice list: all of it)
    if (lb >= le) return;
    const uint32_t top_ref = (skip_bits && n_r) ? wang_hash(R[rlo].x) >> top_shift : 0u;  // leading bits every tuple here must share
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
            const uint32_t hk = wang_hash(k);
            if (skip_bits && (hk >> top_shift) != top_ref) ++my_foreign;
            const uint32_t slot = hk & smask;
            const uint32_t bit = 1u << (slot & 31);
            if ((atomicOr(&s_used[slot >> 5], bit) & bit) && slice == 0) ++my_dup;  // equal key overwrites (hashtable.c:133)
            s_vals[slot] = val;
          }
        }
      }
    }
    __syncthreads();
    {  // probe + take of the right payload
      const uint64_t a = lb & ~1ull;
      const uint64_t span = le - a;
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
            ok[h] = i0 + h >= lb && i0 + h < le;
            if (!ok[h]) continue;
            const uint32_t hk = wang_hash(k);
            if (skip_bits && n_r && (hk >> top_shift) != top_ref) ++my_foreign;
            const uint32_t slot = hk & smask;
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
  };
  // partitions are dealt to the workgroups in runs of `run` consecutive ones (1: round-robin)
  const uint32_t n_runs = (nparts + run - 1) / run;
  for (uint32_t q = blockIdx.x * run; q < n_runs * run; q += gridDim.x * run)
    for (uint32_t p = q; p < min(q + run, nparts); ++p) body(p, 0u);
  const uint32_t n_extra = slices.count ? min(*slices.count, slices.capacity) : 0u;
  for (uint32_t w = blockIdx.x; w < n_extra; w += gridDim.x) body(slices.list[w].x, slices.list[w].y);
  const uint64_t wm = wave_reduce_add((uint64_t)my_miss);
  const uint64_t wd = wave_reduce_add((uint64_t)my_dup);
  const uint64_t wf = wave_reduce_add((uint64_t)my_foreign);
  if (lane_id() == 0) {
    if (wm) atomicAdd(&counters[0], (unsigned long long)wm);
    if (wd) atomicAdd(&counters[1], (unsigned long long)wd);
    if (wf) atomicAdd(&counters[5], (unsigned long long)wf);
  }
  if (threadIdx.x == 0) atomicMax(&counters[3], (unsigned long long)my_max);
}

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

}  // namespace olap
