// common.h -- context, error handling, batched-column views and device helpers
// shared by the HIP translation units of libolap_gpu.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstring>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <unordered_set>
#include <vector>

#include "olap_gpu.h"

namespace olap {

constexpr int kWave = 64;  // CDNA wavefront
constexpr uint32_t kXcds = 8;  // MI355X: workgroup i of a launch runs on XCD i % 8 (each XCD has its own L2)

// ---------------------------------------------------------------- phases
enum Phase : int {
  PH_COPY_TO = 0,
  PH_COPY_FROM,
  PH_SUM,
  PH_FILTER_COUNT,
  PH_FILTER_COMPACT,
  PH_TAKE,
  PH_PART_HIST,
  PH_PART_SCAN,
  PH_PART_SCATTER,
  PH_PART_SCATTER_CLAIM,  // pass 0 of the large-join plan (claimed allotments, no histogram)
  PH_PART_SCATTER_CAP,    // last pass in the capacity layout (no histogram)
  PH_JOIN_PROBE,
  PH_GEN,
  PH_COUNT
};

extern const char* const kPhaseNames[PH_COUNT];

struct PhaseTimer {
  double ms = 0.0;
  uint64_t launches = 0;
};

}  // namespace olap

// The opaque handle of include/olap_gpu.h.
namespace olap {
class Stager;  // pinned-ring host <-> HBM staging (stage.hip)
}

struct olap_ctx {
  int device = 0;
  int compute_units = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  mutable std::string err;

  // scratch arena (device) -- one allocation, bump-carved per operator call
  char* ws = nullptr;
  uint64_t ws_bytes = 0;
  uint64_t ws_used = 0;
  // pinned host scratch for pointer tables / small results
  char* pinned = nullptr;
  uint64_t pinned_bytes = 0;
  uint64_t pinned_used = 0;

  // profiling
  bool profiling = false;
  olap::PhaseTimer timers[olap::PH_COUNT];
  struct Pending {
    int phase;
    hipEvent_t a, b;
  };
  std::vector<Pending> pending;
  std::string phase_kernel[olap::PH_COUNT];  // name of the kernel last launched under each phase, as rocprofv3 prints it
  std::vector<hipEvent_t> event_pool;

  std::unordered_set<const void*> lds_raised;  // kernels whose dynamic-LDS limit was raised on this device

  // join tuning
  int radix_bits_total = -1;
  int max_bits_per_pass = 0;
  int scatter_variant = 0;  // 0 write-combining scatter, 1 tile-staged scatter (A/B)
  int take_variant = 0;    // 0 auto, 1 never the LDS-window kernel, 3 small tiles, 4 the window-bucketed path
  int line16 = 0;          // 1: 128-byte write-combining lines for passes of <= 512 buckets (measured: no gain yet, the kernel is not write-bound there)
  int direct_table = 1;    // collision-free direct-address LDS table when <= 14 hash bits remain (0: always probe)
  int wc_geometry = 2;     // 2: passes of <= 512 buckets use 512-thread workgroups on 4096-tuple tiles, two per CU
  int capacity_mode = 1;   // last radix pass without histogram (fixed-capacity regions + exact fallback)
  int ring_scatter = 3;    // bit 0: claiming pass, bit 1: capacity pass: passes of <= 512 buckets run the ring scatter (ring_scatter.h): 128-byte lines, one workgroup per CU
  int filter_whole_batches = 2;  // filter: from this many long batches per CU on, one workgroup walks a whole batch (no count pass); 0 off
  int balance_segments = 1; // exact radix passes over segments: workgroups per segment in proportion to the segment sizes (skew)
  int plan_first_bits = 0;   // two-pass plans: radix bits of pass 0 (0: even split)
  int probe_small_table = 1; // probing-table probe: a 4096-slot table (four workgroups per CU) when the build partitions hold <= 2048 rows on average (0: always 8192 slots)
  bool probe_big_once = false;   // (join_run: a partition overflowed the small table -- the retry takes the large one)
  uint32_t last_probe_slots = 0; // table slots of the last probing-table probe
  int probe_slices = 1;    // probe partitions longer than 64 Ki rows are shared by several workgroups (skewed foreign keys)
  int broadcast_join = 1;  // a right table that fits one LDS table (<= 6144 rows) is joined without partitioning
  int probe_run = 1;       // direct-table probe: consecutive partitions per workgroup turn
  int ring_split_ragged = 1; // ring scatter from ragged columns: whole aligned tiles and the rest in two launches (0: one launch, generic loader)
  int ring_full_tiles = 1; // ring scatter from columns: the straight-line loader when all batches are whole aligned tiles
  int claim_weave = 1;     // claiming ring scatter writes woven regions (radix_kernels.h woven_at)
  int ring_interleave = -1; // ring scatter from columns: tiles dealt to the workgroups round-robin in chunks of this many (0: one contiguous run each; -1: automatic)
  int pack6 = 1;           // the last (capacity, ring) pass hands the direct-table probe 6-byte tuples {payload, low 16 hash bits} (pack6.h)
  int contig_columns = 1;  // pass 0's straight-line loader addresses batches that are slices of one buffer without the pointer tables
  int claim_lean = 1;      // claiming pass from columns: the kernel of ring_claim.h (0: ring_scatter.h's)
  int pack6_hash = 1;      // ... and the claiming pass in front of it leaves wang_hash(key) in place of the key (one hash per tuple less)
  int ring_exact_blocks = 2; // ... from the columns: workgroups per CU of histogram and scatter (2: two rounds of ring workgroups behind a 512-thread histogram; 1: one round behind a 1024-thread histogram)
  int ring_exact = 1;      // histogram + scan passes of <= 512 buckets over >= 2^26 rows scatter through the ring kernel into the exact layout (0: write-combining kernel, 2: any size)
  int filter_stores = 1;   // filter: 0 every survivor its own store | 1 tiles where most rows pass leave through an LDS strip as 16-byte stores, for segments of >= 32 Ki rows (k_filter_compact_lds) | 2 ... for every shape
  int presplit_refine = 1; // ... its chunk places are cut finer until this many non-empty ones per CU exist (2 measured slower: 7.2 vs 6.6 ms per rank at G=8)
  int presplit_sub = 1;    // pre-split joins with one pass left: that pass chunk by chunk into packed sub-partitions (0 never, 1 when the chunks fill the chip, 2 always)
  int join64_ring = 1;     // olap_join_u64: the histogram-free plan on 12-byte tuples (ring64.h): 0 never, 1 automatic, 2 every two-pass plan of 4..9 bits per pass
  int join64_tpt = 8;      // ... tuples per producer thread and tile of its passes of <= 256 buckets (8 | 4; 512 buckets always take 4)
  int probe64_variant = 5; // k64_probe12 (ring64.h): W + 4 * DH -- W tuples of a thread probe together (1 | 2), DH double hashing instead of linear probing
  int probe64_grid = 24;   // ... workgroups per CU of its grid (three are resident; 8: +5 %, profiles/r3_probe64_ab.md)
  int ring6_tpt = 8;       // packed last pass: tuples per producer thread and tile (8: 4096-tuple tiles flushed at once; 4: 2048-tuple tiles, the flush one tile behind the insert)
  int probe6_prefetch = 0; // packed probe: the next partition's sizes and first chunks are requested ahead (k_join_probe_direct6_pf)
  int probe6_geometry = 0; // packed probe: 0 two workgroups of 512 per CU, 1 one workgroup of 1024
  int ring_tpt = 8;        // tuples per producer thread and tile of the ring scatter (8: 4096-tuple tiles, 4: 2048)
  int claim_mode = 1;      // pass 0 of a two-pass plan without histogram (claimed allotments): 0 off, 1 auto, 2 always
  int hash_skip_bits = 0;  // top hash bits already fixed by an upstream owner split
  int join_drop_unmatched = 0;       // 1: inner-join semantics (rows without a match are dropped, not an error)
  hipEvent_t right_ready = nullptr;  // one-shot: the next join waits for it before reading its right table
  // diagnostics of the last join call (olap_get_counter)
  uint64_t last_join_attempts = 0;        // 1 = the first plan went through
  uint64_t last_join_region_overflows = 0;  // capacity / claimed regions that overflowed in the first attempt (skew -> exact rerun)

  // OLAP_MEM_HOST operands
  olap::Stager* stager = nullptr;
  int host_stage = 1;         // 1: pinned ring + copy threads; 0: plain hipMemcpyAsync per buffer (A/B)
  int host_copy_threads = 0;  // 0: min(hardware threads, 8)

  int fail(int code, const char* fmt, ...) const {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
};

#define OLAP_HIP(ctx, call)                                                              \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess)                                                               \
      return (ctx)->fail(OLAP_ERR_DEVICE, "%s failed: %s (%s:%d)", #call,                \
                         hipGetErrorString(e__), __FILE__, __LINE__);                    \
  } while (0)

#define OLAP_TRY(expr)            \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != OLAP_OK) return rc__; \
  } while (0)

namespace olap {

// ---- context services (ctx.hip)
int ws_begin(olap_ctx* ctx, uint64_t bytes);                 // ensure capacity, reset bump pointer
void* ws_alloc(olap_ctx* ctx, uint64_t bytes);               // 256-B aligned carve (nullptr if exhausted)
int pinned_begin(olap_ctx* ctx, uint64_t bytes);
void* pinned_alloc(olap_ctx* ctx, uint64_t bytes);
void phase_begin(olap_ctx* ctx, int phase);                  // record start event if profiling
void phase_end(olap_ctx* ctx, int phase);                    // record stop event if profiling
int resolve_timers(olap_ctx* ctx);                           // after a sync: fold pending events

// ---- host <-> HBM staging (stage.hip)
struct StageItem {
  void* dev;
  void* host;
  uint64_t bytes;
};
int stage_h2d(olap_ctx* ctx, const std::vector<StageItem>& items);  // enqueued on ctx->stream when it returns
int stage_d2h(olap_ctx* ctx, const std::vector<StageItem>& items);  // host buffers filled when it returns
void stager_destroy(olap_ctx* ctx);

struct PhaseScope {
  olap_ctx* ctx;
  int phase;
  PhaseScope(olap_ctx* c, int p) : ctx(c), phase(p) { phase_begin(ctx, phase); }
  ~PhaseScope() { phase_end(ctx, phase); }
};

inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
inline uint32_t dtype_bytes(int dtype) { return (dtype == OLAP_U32 || dtype == OLAP_I32) ? 4u : 8u; }
inline bool dtype_ok(int dtype) { return dtype >= OLAP_U32 && dtype <= OLAP_I64; }

// A set of batches (one data buffer per batch) cut into fixed-size units of
// `unit_rows` rows that never straddle a batch.  unit u of the view belongs to
// batch b = locate(u) and covers rows [lu*unit_rows, ...) of that batch.
struct ColView {
  const uint64_t* unit_start;  // [nb+1] prefix sum of units per batch (device)
  const uint64_t* row_start;   // [nb+1] prefix sum of rows per batch (device)
  uint64_t total_units;
  uint64_t total_rows;
  uint32_t nb;
  uint32_t unit_rows;
  uint32_t units_per_batch;  // > 0 when every batch has the same row count
  uint32_t uniform_rows;     // that row count (valid when units_per_batch > 0)
  const uint64_t* row0;      // optional [nb]: global index of each entry's first row when the entries are PIECES of batches
                             // (split_column_set); null: row_start[b]
};

// Host-side staging of a batched column set: builds the device tables (and, for
// OLAP_MEM_HOST inputs, copies the buffers into the arena).
struct ColumnSet {
  ColView view{};
  std::vector<const void* const*> dev_ptr_tables;  // one device pointer table per column
  bool full_units = false;  // every batch is a whole number of units and every buffer is 16-byte aligned
  bool contiguous = false;  // ... and in every column batch b + 1 starts where batch b ends (slices of one buffer)
  // host copies (device addresses, rows, element sizes) for views derived from this one (split_column_set)
  std::vector<std::vector<const void*>> h_ptrs;
  std::vector<uint64_t> h_rows;
  std::vector<uint32_t> h_elem;
};

// Two views of the same buffers: `whole` = the whole units of every batch that start at its first 16-byte boundary
// (full_units holds), `rest` = everything else (the rows in front of that boundary, the tails; batches whose columns are
// misaligned differently entirely).  Both number their rows
// globally like `cs` (ColView::row0).  Tables come from the arena / pinned scratch of the current call.
int split_column_set(olap_ctx* ctx, const ColumnSet& cs, ColumnSet* whole, ColumnSet* rest);
uint64_t split_column_set_bytes(uint64_t nb, int ncols);
inline uint64_t split_column_set_pinned(uint64_t nb, int ncols) { return 2 * (nb + 2) * 8 * (3 + (uint64_t)ncols) + 1024; }

// Builds view tables in the arena.  `cols` = ncols arrays of nb host-side pointers.
// For in_mem == HOST the data is copied to the arena and the tables point there.
int build_column_set(olap_ctx* ctx, const void* const* const* cols, int ncols,
                     const uint32_t* elem_bytes, const uint64_t* rows, uint64_t nb, int in_mem,
                     uint32_t unit_rows, ColumnSet* out);
uint64_t column_set_bytes(const uint64_t* rows, uint64_t nb, int ncols, const uint32_t* elem_bytes,
                          int in_mem);

}  // namespace olap

// ---------------------------------------------------------------- device side
#if defined(__HIPCC__)
namespace olap {

// Wang hash -- bit-exact with dpu/shared/kernels/partition.c:20-28 and
// dpu/shared/hashtable/hashtable.c:29-37 of the reference (checked against the
// known-answer values in tests/golden/reference_kats.json).
__host__ __device__ __forceinline__ uint32_t wang_hash(uint32_t key) {
  key += ~(key << 15);
  key ^= (key >> 10);
  key += (key << 3);
  key ^= (key >> 6);
  key += ~(key << 11);
  key ^= (key >> 16);
  return key;
}

// 64-bit finaliser (splitmix64) used by the generator and the digests.
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Counter-based generator: value of stream `s` at global row `r`.
__host__ __device__ __forceinline__ uint64_t gen64(uint64_t seed, uint32_t s, uint64_t r) {
  return mix64(mix64(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(s + 1))) + r);
}
__host__ __device__ __forceinline__ uint32_t gen32(uint64_t seed, uint32_t s, uint64_t r) {
  return (uint32_t)(gen64(seed, s, r) >> 32);
}

struct Loc {
  uint32_t batch;
  uint64_t local_unit;
};

__device__ __forceinline__ Loc locate(const ColView& v, uint64_t unit) {
  Loc l;
  if (v.units_per_batch) {
    l.batch = (uint32_t)(unit / v.units_per_batch);
    l.local_unit = unit - (uint64_t)l.batch * v.units_per_batch;
  } else {
    uint32_t lo = 0, hi = v.nb;  // find last b with unit_start[b] <= unit
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (v.unit_start[mid] <= unit) lo = mid; else hi = mid;
    }
    l.batch = lo;
    l.local_unit = unit - v.unit_start[lo];
  }
  return l;
}

__device__ __forceinline__ uint64_t batch_rows(const ColView& v, uint32_t b) {
  return v.units_per_batch ? (uint64_t)v.uniform_rows : v.row_start[b + 1] - v.row_start[b];
}

__device__ __forceinline__ uint64_t batch_row0(const ColView& v, uint32_t b) {
  if (v.row0) return v.row0[b];
  return v.units_per_batch ? (uint64_t)b * v.uniform_rows : v.row_start[b];
}

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & (kWave - 1); }

// A value every lane of the wave holds alike (a batch-table entry read with a vector load): moved to
// scalar registers, which frees two VGPRs per pointer in the partition kernels' tile loops.
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// Loads through a pointer that was itself read from memory (the batch pointer tables).  The compiler
// cannot tell that such a pointer is global and emits FLAT loads for it, which count in lgkmcnt as
// well as vmcnt: every `s_waitcnt lgkmcnt(0)` in front of an s_barrier or an LDS access then waits for
// the loads in flight.  These helpers state the address space, so the loads are global_load_*.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#define OLAP_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ uint4 gload4(const void* p) {
  const u32x4_t v = *(const OLAP_GLOBAL u32x4_t*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 gload2(const void* p) {
  const u32x2_t v = *(const OLAP_GLOBAL u32x2_t*)p;
  return make_uint2(v.x, v.y);
}
// streaming variants (read once / written once): the `nt` cache policy
__device__ __forceinline__ uint4 gload4_nt(const void* p) {
  const u32x4_t v = __builtin_nontemporal_load((const OLAP_GLOBAL u32x4_t*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 gload2_nt(const void* p) {
  const u32x2_t v = __builtin_nontemporal_load((const OLAP_GLOBAL u32x2_t*)p);
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void gstore4(void* p, uint4 v) {
  u32x4_t t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  *(OLAP_GLOBAL u32x4_t*)p = t;
}
__device__ __forceinline__ void gstore4_nt(void* p, uint4 v) {
  u32x4_t t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, (OLAP_GLOBAL u32x4_t*)p);
}
__device__ __forceinline__ void gstore2(void* p, uint2 v) {
  u32x2_t t;
  t.x = v.x;
  t.y = v.y;
  *(OLAP_GLOBAL u32x2_t*)p = t;
}
// any 16-byte vector type / any integer scalar through a global (not flat) access
template <class V>
__device__ __forceinline__ V gload_vec16(const V* p) {
  static_assert(sizeof(V) == 16, "16-byte vectors");
  // The column streams of sum / filter / take are read ONCE: non-temporal loads (k_sum 2.77 -> 2.55 ms per 16 GiB = 6.7 TB/s,
  // int64 5.8 -> 5.2 ms per 32 GiB; profiles/r3_sum_nt_shapes.txt).  -DOLAP_VEC16_PLAIN_LOAD=1 builds the A/B partner.
#ifndef OLAP_VEC16_PLAIN_LOAD
  const u32x4_t t = __builtin_nontemporal_load((const OLAP_GLOBAL u32x4_t*)p);
#else
  const u32x4_t t = *(const OLAP_GLOBAL u32x4_t*)p;
#endif
  V v;
  __builtin_memcpy(&v, &t, 16);
  return v;
}
template <class V>
__device__ __forceinline__ void gstore_vec16(V* p, const V& v) {
  static_assert(sizeof(V) == 16, "16-byte vectors");
  u32x4_t t;
  __builtin_memcpy(&t, &v, 16);
  *(OLAP_GLOBAL u32x4_t*)p = t;
}
template <class T>
__device__ __forceinline__ T gload_t(const T* p) { return *(const OLAP_GLOBAL T*)p; }
template <class T>
__device__ __forceinline__ void gstore_t(T* p, T v) { *(OLAP_GLOBAL T*)p = v; }
__device__ __forceinline__ uint32_t gload1(const void* p) { return *(const OLAP_GLOBAL uint32_t*)p; }
__device__ __forceinline__ uint64_t gload1_u64(const void* p) { return *(const OLAP_GLOBAL uint64_t*)p; }

__device__ __forceinline__ uint64_t wave_reduce_add(uint64_t x) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) x += __shfl_down(x, o, kWave);
  return x;
}

}  // namespace olap
#endif
