/*
 * olap_gpu.h -- C ABI of the MI355X (gfx950) operator hot path that replaces the
 * UPMEM DPU path of upmem/dpu_olap: filter, take, sum-aggregate, radix partition
 * and radix-partitioned PK-FK hash join over Arrow fixed-width integer columns.
 *
 * This header IS the drop-in boundary.  It replaces the reference's
 * host<->device protocol, i.e. shared/umq/kernels.h:12-51 (`enum Kernel`, the
 * per-kernel INPUT/OUTPUT structs) together with the libdpu calls that carry it
 * (host/dpuext/dpuext.hpp:148-927, host/dpuext/arrow_utils.cc:25-266).  Where the
 * reference addresses DPU memory by symbol name ("buffer", "output_buffer",
 * "selection_indices_vector", "INPUT", "kernel") and launches one DPU program per
 * operator (dpu/{filter,take,aggr,partition,join}/main.c), this ABI takes plain
 * pointers and sizes and launches hand-written HIP kernels.
 *
 * Conventions
 *  - Every function returns an olap_status (0 = ok, negative = error class) and
 *    never throws or aborts (the reference: DpuError exceptions / DPU_ASSERT,
 *    host/dpuext/dpuext.hpp:54-90, host/dpuext/status.h:7-11).  A message is
 *    kept per context: olap_last_error().
 *  - Columns are passed the way the reference's operators see them: a vector of
 *    batches (arrow::RecordBatchVector), one contiguous data buffer per (batch,
 *    column), no validity bitmap (all benchmark columns are non-nullable,
 *    host/dpuext/arrow_utils.h:27-29).  `ptrs[b]` is the data buffer of batch b
 *    (Arrow buffer index 1, host/dpuext/arrow_utils.cc:64-66), `rows[b]` its
 *    length in elements.  The pointer/row TABLES are always host arrays; `mem`
 *    says where the buffers they point to live.
 *  - OLAP_MEM_DEVICE buffers must be 4-byte (32-bit types) / 8-byte (64-bit
 *    types) aligned; 16-byte alignment takes the vector fast path.
 *  - A context is single-caller (one host thread drives it); distinct contexts
 *    are independent.  One context == one GPU; a device GROUP (olap_group_*, below)
 *    owns one context per device and runs the join across them.
 *  - All calls are synchronous on return unless stated otherwise.
 */
#ifndef OLAP_GPU_H_
#define OLAP_GPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OLAP_ABI_VERSION 1

typedef struct olap_ctx olap_ctx;

typedef enum {
  OLAP_OK = 0,
  OLAP_ERR_INVALID = -1,   /* bad argument */
  OLAP_ERR_DEVICE = -2,    /* HIP runtime error (message has hipGetErrorString) */
  OLAP_ERR_NOMEM = -3,     /* device / pinned allocation failed */
  OLAP_ERR_NO_MATCH = -4,  /* join: a foreign key has no primary key (reference asserts, dpu/shared/kernels/hash_probe.c:32-33) */
  OLAP_ERR_OVERFLOW = -5,  /* join: a final partition's build side exceeds the LDS table (reference: ht_put false -> assert, hash_build.c:30-31) */
  OLAP_ERR_CAPACITY = -6   /* caller-provided output buffer too small */
} olap_status;

/* Element types (the reference hard-wires uint32_t, dpu/shared/kernels/common.h:3-4) */
typedef enum { OLAP_U32 = 0, OLAP_I32 = 1, OLAP_U64 = 2, OLAP_I64 = 3 } olap_dtype;

typedef enum { OLAP_MEM_HOST = 0, OLAP_MEM_DEVICE = 1 } olap_mem;

/* ------------------------------------------------------------------ context
 * Replaces dpu::DpuSet::allocate()/load() (host/dpuext/dpuext.hpp:709-715,
 * 738-744): there is no program-load step, kernels are linked in. */
int olap_abi_version(void);
int olap_create(int device_id, olap_ctx** out_ctx);
void olap_destroy(olap_ctx* ctx);
const char* olap_last_error(const olap_ctx* ctx);
/* name[] receives the device name; any out pointer may be NULL. */
int olap_device_info(olap_ctx* ctx, char* name, size_t name_cap, int* compute_units,
                     uint64_t* total_mem_bytes, uint64_t* free_mem_bytes);
/* Run the context's kernels on a caller-owned hipStream_t (NULL = the context's
 * own stream).  Lets a host runtime order our kernels with its own work. */
int olap_set_stream(olap_ctx* ctx, void* hip_stream);
void* olap_get_stream(olap_ctx* ctx);
int olap_synchronize(olap_ctx* ctx);
/* Pre-size the internal scratch arena (partition ping-pong buffers, histograms,
 * staging).  Optional: the arena grows on demand.  Mirrors the role of
 * Prepare() (host/join/join_dpu.cc:144-156). */
int olap_reserve_workspace(olap_ctx* ctx, uint64_t bytes);
uint64_t olap_workspace_bytes(const olap_ctx* ctx);

/* -------------------------------------------------------- device memory
 * For callers that keep columns resident in HBM (the benchmark harness).
 * Replaces the static per-DPU MRAM buffers (dpu/filter/main.c:20-27). */
int olap_malloc(olap_ctx* ctx, uint64_t bytes, void** out_dptr);
int olap_free(olap_ctx* ctx, void* dptr);
int olap_memcpy_h2d(olap_ctx* ctx, void* dst_device, const void* src_host, uint64_t bytes);
int olap_memcpy_d2h(olap_ctx* ctx, void* dst_host, const void* src_device, uint64_t bytes);
int olap_memset(olap_ctx* ctx, void* dst_device, int byte_value, uint64_t bytes);

/* ------------------------------------------------------------------ timers
 * Replaces timer::Timers (host/timer/timer.h:28-39; names copy-to-dpu, dpu-work,
 * copy-from-dpu of host/filter/filter_dpu.cc:107-110 and the join phases of
 * host/join/join_dpu.cc:146-148).  With profiling on, every kernel launch is
 * bracketed by HIP events on the context's stream; the accumulated device time
 * and launch count per phase are returned.  Phase names: "copy-to-gpu",
 * "copy-from-gpu", "sum", "filter-count", "filter-compact", "take",
 * "partition-hist", "partition-scan", "partition-scatter" (exact layout, behind a histogram),
 * "partition-scatter-claim" (pass 0 without histogram), "partition-scatter-capacity" (last pass
 * without histogram), "join-probe", "gen". */
int olap_set_profiling(olap_ctx* ctx, int enabled);
int olap_reset_timers(olap_ctx* ctx);
/* Fills up to `cap` entries; *out_n = number of phases with launches. names[i]
 * points to static storage. */
int olap_get_timers(olap_ctx* ctx, const char** names, double* total_ms, uint64_t* launches,
                    int cap, int* out_n);

/* Name of the kernel last launched under a phase of olap_get_timers, spelled as rocprofv3 --kernel-trace prints it
 * (e.g. "olap::k_part_scatter_ring<0, true, 512, 16, 1024, 8, 8>"), so a bench line can be matched against a
 * profile; "" for phases that are not a single kernel.  Static storage of the context. */
const char* olap_phase_kernel(olap_ctx* ctx, const char* phase_name);

/* ------------------------------------------------------------------ filter
 * Replaces kernel_filter (dpu/shared/kernels/filter.c:57-177, predicate :25) and
 * FilterDpu::GetResult's per-DPU result pull (host/filter/filter_dpu.cc:50-101).
 * For every batch b: out_ptrs[b][0 .. out_rows[b]) = the values v of batch b with
 * v < *threshold (compared in `dtype`, so unsigned for U32/U64), input order
 * preserved.  out_ptrs[b] must have room for rows[b] elements (worst case, as the
 * reference sizes its transfer, filter_dpu.cc:67-83) and lives in `out_mem`.
 * out_rows (host array, n_batches) and *out_total (may be NULL) are filled. */
int olap_filter_lt(olap_ctx* ctx, int dtype, const void* const* ptrs, const uint64_t* rows,
                   uint64_t n_batches, int in_mem, const void* threshold, void* const* out_ptrs,
                   int out_mem, uint64_t* out_rows, uint64_t* out_total);

/* --------------------------------------------------------------------- sum
 * Replaces kernel_aggr + sum (dpu/shared/kernels/aggr.c:16-33,
 * dpu/aggr/main.c:44-50,76-88) and the host add (host/aggr/aggr_dpu.cc:82-84).
 * U32 -> uint64 sum, I32 -> int64 sum, U64/I64 wrap mod 2^64 (Arrow "sum"
 * semantics).  The 64-bit result bit pattern is stored in *out_sum. */
int olap_sum(olap_ctx* ctx, int dtype, const void* const* ptrs, const uint64_t* rows,
             uint64_t n_batches, int in_mem, uint64_t* out_sum);

/* -------------------------------------------------------------------- take
 * Replaces kernel_take (dpu/shared/kernels/take.c:12-47, params
 * shared/umq/kernels.h:39-42): for every batch b and i < idx_rows[b]:
 *   out_ptrs[b][i] = data_ptrs[b][ idx_ptrs[b][i] ]      (uint32 indices)
 * No bounds check, like the reference (take.c:36) and TakeNative's
 * NoBoundsCheck (host/take/take_native.cc:27).  dtype gives the data width. */
int olap_take(olap_ctx* ctx, int dtype, const void* const* data_ptrs, const uint64_t* data_rows,
              const uint32_t* const* idx_ptrs, const uint64_t* idx_rows, uint64_t n_batches,
              int in_mem, void* const* out_ptrs, int out_mem);

/* --------------------------------------------------------------- partition
 * Replaces kernel_partition (dpu/shared/kernels/partition.c:296-341, params
 * shared/umq/kernels.h:27-37), the payload re-gather (Partitioner::TakeKernel,
 * host/partition/partitioner.cc:185-207) and the shuffle (GetOffsets /
 * LoadPartitions, partitioner.cc:280-375) in one device-side step.
 * Tuples (key, payload) of all batches are grouped by
 *   bucket = wang_hash(key) >> (32 - log2(nr_partitions))   (partition.c:45-46)
 * into out_keys / out_payload (contiguous, capacity `out_capacity` rows, in
 * out_mem); out_offsets (host, nr_partitions + 1) receives the start of every
 * partition (out_offsets[p+1] is the END offset the reference calls metadata,
 * host/join/join_dpu.cc:54-66).  Order inside a partition is unspecified, as in
 * the reference.  nr_partitions: power of two, 2 .. 2^20.  payload_ptrs may be
 * NULL (keys only; out_payload then receives the global row index). */
int olap_partition_u32(olap_ctx* ctx, const uint32_t* const* key_ptrs,
                       const uint32_t* const* payload_ptrs, const uint64_t* rows,
                       uint64_t n_batches, int in_mem, uint32_t nr_partitions, uint32_t* out_keys,
                       uint32_t* out_payload, uint64_t out_capacity, int out_mem,
                       uint64_t* out_offsets);

/* Same partitioning, but the (key, payload) tuples stay zipped: out_tuples[i] =
 * key | (uint64_t)payload << 32 (device memory, 16-byte aligned, capacity in tuples).
 * This is the send buffer of the multi-GPU owner split: partition p is the contiguous
 * region [out_offsets[p], out_offsets[p+1]) for peer p. */
int olap_partition_tuples_u32(olap_ctx* ctx, const uint32_t* const* key_ptrs,
                              const uint32_t* const* payload_ptrs, const uint64_t* rows,
                              uint64_t n_batches, int in_mem, uint32_t nr_partitions,
                              uint64_t* out_tuples, uint64_t out_capacity, uint64_t* out_offsets);

/* -------------------------------------------------------------------- join
 * Replaces JoinDpu::Run_internal (host/join/join_dpu.cc:168-400): radix
 * partition of both tables, then per partition kernel_hash_build
 * (dpu/shared/kernels/hash_build.c:9-35) on the right key, kernel_hash_probe
 * (hash_probe.c:9-46) with the left key and the take of the right payload.
 * left = {fk, y}, right = {pk, x}; output rows {fk, y, x} (left columns then the
 * right non-key column, join_dpu.cc:314-363), partition-major, order inside a
 * partition unspecified (tests sort, host/join/join_test.cc:27-38).  A right table
 * of <= 6144 rows (one LDS table) is joined without partitioning (stats.passes == 0,
 * nr_partitions == 1): the rows then come in LEFT ROW ORDER.  A small right table
 * against a large left one otherwise gets at least 1024 partitions, to keep the chip busy.
 * Contract of the reference kept: pk unique (a duplicate overwrites,
 * hashtable.c:133 -- counted in stats), every fk has a match (else
 * OLAP_ERR_NO_MATCH).  out_* have capacity `out_capacity` >= total left rows. */
typedef struct {
  uint32_t passes;           /* radix passes run per side */
  uint32_t radix_bits[4];    /* bits per pass */
  uint32_t nr_partitions;    /* final partitions = 2^sum(bits) */
  uint32_t table_slots;      /* LDS hash table slots per partition */
  uint64_t left_rows, right_rows;
  uint64_t max_build_rows;   /* largest final build partition */
  uint64_t duplicate_keys;   /* build keys that overwrote an equal key */
  uint64_t missing_keys;     /* probe keys without a match */
} olap_join_stats;

/* Selection-vector form: left_y_ptrs == NULL makes out_y the GLOBAL LEFT ROW INDEX of every
 * result row (rows numbered across the batches in order), right_x_ptrs == NULL makes out_x
 * the global right row index of its match -- what kernel_hash_probe returns
 * (dpu/shared/kernels/hash_probe.c:9-46) and what the host then feeds to the take kernel for
 * every further column (host/join/join_dpu.cc:325-363). */
int olap_join_u32(olap_ctx* ctx, const uint32_t* const* left_fk_ptrs,
                  const uint32_t* const* left_y_ptrs, const uint64_t* left_rows,
                  uint64_t n_left_batches, const uint32_t* const* right_pk_ptrs,
                  const uint32_t* const* right_x_ptrs, const uint64_t* right_rows,
                  uint64_t n_right_batches, int in_mem, uint32_t* out_fk, uint32_t* out_y,
                  uint32_t* out_x, uint64_t out_capacity, int out_mem, uint64_t* out_rows,
                  olap_join_stats* stats /* may be NULL */);

/* The join on 64-BIT KEYS (uint64 / int64 bit patterns) with 32-bit payloads: the reference's HT_64BIT_KEYS build
 * (dpu/shared/hashtable/hashtable.h:13-18: ht_key_t = uint64_t, values stay uint32_t; hash of a 64-bit key
 * hashtable.c:23-26).  left = {fk: u64, y: u32}, right = {pk: u64, x: u32}, output {fk: u64, y, x}; NULL payload
 * tables give selection vectors as in olap_join_u32.  Same contract (pk unique, every fk matches unless
 * OLAP_UNMATCHED_DROP), same statistics; partition = top bits of wang_hash(hash64(key)) (dpu/shared/kernels/partition.c:45-46
 * applied to the table hash), probing LDS table with 64-bit keys, slot = hash64(key) & (slots - 1) (hashtable.c:99-101).
 * Large joins (from ~2^26 rows per side) take two histogram-free passes on 12-byte tuples {key, payload} (tuning knob
 * join64_ring), smaller ones and skewed ones histogram + scan + scatter passes on 16-byte tuples. */
int olap_join_u64(olap_ctx* ctx, const uint64_t* const* left_fk_ptrs, const uint32_t* const* left_y_ptrs,
                  const uint64_t* left_rows, uint64_t n_left_batches, const uint64_t* const* right_pk_ptrs,
                  const uint32_t* const* right_x_ptrs, const uint64_t* right_rows, uint64_t n_right_batches,
                  int in_mem, uint64_t* out_fk, uint32_t* out_y, uint32_t* out_x, uint64_t out_capacity,
                  int out_mem, uint64_t* out_rows, olap_join_stats* stats /* may be NULL */);

/* The same join on zipped tuples already resident in HBM (what a rank holds after the
 * all-to-all): left_tuples[i] = fk | y << 32, right_tuples[i] = pk | x << 32, 16-byte
 * aligned device arrays; outputs are device columns. */
int olap_join_tuples_u32(olap_ctx* ctx, const uint64_t* left_tuples, uint64_t n_left,
                         const uint64_t* right_tuples, uint64_t n_right, uint32_t* out_fk,
                         uint32_t* out_y, uint32_t* out_x, uint64_t out_capacity, uint64_t* out_rows,
                         olap_join_stats* stats /* may be NULL */);

/* The multi-device split: olap_partition_tuples_u32 with 2^(owner_bits + group_bits + low_bits)
 * buckets in ONE radix pass, but with the buckets laid out group-major.  A tuple's bucket id, read
 * from the top bits of its hash below the skipped ones, is [owner | group | low]; the output (and
 * out_offsets, 2^bits + 1 entries) is ordered [group | owner | low].  So what a rank sends to
 * all owners for one sub-bucket GROUP is a single contiguous run -- one all_to_all_single per
 * group -- and a receiver can join group g (olap_join_presplit_u32 with
 * olap_join_set_hash_skip_bits(owner_bits + group_bits) and 2^low_bits segments) while group
 * g + 1 is still on the links.  group_bits = 0 gives the natural order. */
int olap_split_tuples_u32(olap_ctx* ctx, const uint32_t* const* key_ptrs,
                          const uint32_t* const* payload_ptrs, const uint64_t* rows,
                          uint64_t n_batches, int in_mem, uint32_t owner_bits, uint32_t group_bits,
                          uint32_t low_bits, uint64_t* out_tuples, uint64_t out_capacity,
                          uint64_t* out_offsets);

/* The join on zipped tuples that arrive PRE-SPLIT by their leading hash bits -- what a rank
 * holds when every sender ran olap_partition_tuples_u32 with nr_partitions = ranks x
 * n_segments and shipped each owner its n_segments runs in one message.  Segment j fixes
 * the log2(n_segments) hash bits directly below the skipped ones
 * (olap_join_set_hash_skip_bits) and is the union of chunks_per_segment chunks
 * [chunk_lo[j * chunks_per_segment + c], chunk_hi[...]) -- tuple indices into the device
 * array, host tables, one chunk per sender.  The first radix pass reads the chunks in
 * place (one workgroup per chunk), so the exchanged data is not re-grouped first.  This
 * replaces the host-side regrouping of Partitioner::GetOffsets / LoadPartitions
 * (host/partition/partitioner.cc:280-375) for the multi-device case.  Row order of the result: segment-major; inside a
 * segment partition-major, or -- when the one remaining pass runs chunk by chunk (tuning knob presplit_sub) -- chunk-major,
 * then partition-major. */
int olap_join_presplit_u32(olap_ctx* ctx, const uint64_t* left_tuples, uint64_t n_left,
                           const uint64_t* left_chunk_lo, const uint64_t* left_chunk_hi,
                           const uint64_t* right_tuples, uint64_t n_right,
                           const uint64_t* right_chunk_lo, const uint64_t* right_chunk_hi,
                           uint32_t n_segments, uint32_t chunks_per_segment, uint32_t* out_fk,
                           uint32_t* out_y, uint32_t* out_x, uint64_t out_capacity, uint64_t* out_rows,
                           olap_join_stats* stats /* may be NULL */);

/* What a foreign key without a matching primary key does.  OLAP_UNMATCHED_ERROR (default):
 * the join fails with OLAP_ERR_NO_MATCH -- the reference asserts on it
 * (dpu/shared/kernels/hash_probe.c:33-34).  OLAP_UNMATCHED_DROP: inner-join semantics, like
 * the Arrow hash join behind JoinNative (host/join/join_native.cc:31-49): such rows are left
 * out, *out_rows < left rows, stats->missing_keys says how many. */
enum { OLAP_UNMATCHED_ERROR = 0, OLAP_UNMATCHED_DROP = 1 };
int olap_join_set_unmatched(olap_ctx* ctx, int mode);

/* One-shot: the NEXT join on this context partitions its left table first and makes the
 * context's stream wait for `hip_event` (a recorded hipEvent_t) before it reads a byte of the
 * right table -- so the right table's exchange can still be in flight when the join is
 * called, and its transfer overlaps the left table's radix pass.  NULL clears it. */
int olap_join_set_right_ready_event(olap_ctx* ctx, void* hip_event);

/* Radix plan override.  radix_bits_total < 0 keeps the automatic choice: 18 - skip bits
 * (16 Ki-row partitions, direct-address LDS table) for large build sides, else
 * ceil(log2(right_rows / 4096)) (probing table); max_bits_per_pass 0 = up to 10. */
int olap_join_configure(olap_ctx* ctx, int radix_bits_total, int max_bits_per_pass);
/* Kernel-variant switches for A/B measurements in one process; every combination gives the same
 * rows (covered by the randomized shape / plan / variant test in tests/test_gpu_parity.py).
 *   scatter_variant    0 write-combining scatter (default) | 1 tile-staged scatter
 *   wc_geometry        2 two 512-thread workgroups per CU for passes of <= 512 buckets (default) | 1 one of 1024
 *   line16             0 64-byte lines (default) | 1 128-byte lines for passes of <= 512 buckets (one workgroup per CU)
 *   capacity_mode      1 last pass without histogram, fixed-capacity regions (default) | 0 exact layout
 *   claim_mode         1 pass 0 of large two-pass plans as a claiming scatter (default) | 0 never | 2 always
 *   ring_scatter       3 claiming AND capacity passes of <= 512 buckets run the ring scatter: per-bucket LDS rings, 128-byte
 *                      lines, one 1024-thread workgroup per CU (default) | 1 claiming pass only | 2 capacity pass only
 *                      | 0 the write-combining kernels
 *   ring_exact         1 histogram + scan + scatter passes of <= 512 buckets over >= 2^26 rows scatter through the ring kernel as
 *                      well, into the exact layout (a run's first and last, partial lines as 8-byte stores; default) | 0 the
 *                      write-combining kernel | 2 the ring kernel at any size
 *   ring_tpt           8 (default) | 4 tuples per producer thread and tile of the ring scatter
 *   filter_whole_batches N filter: from N batches per CU on (default 2), a batch longer than 64 Ki rows is still walked by ONE
 *                      workgroup (one pass over the data) | 0 always count + scan + compact over 64 Ki-row segments
 *   filter_stores      1 filter: tiles in which at least half of the rows pass leave through an LDS strip as whole 16-byte vectors,
 *                      for segments of >= 32 Ki rows in 16-byte aligned output buffers (default) | 0 every surviving element its own
 *                      store | 2 the strip for every batch geometry
 *   balance_segments   1 histogram + scatter passes over the segments of an earlier pass give every segment workgroups in
 *                      proportion to its size (a table computed on the device): a segment far larger than the rest -- a hot
 *                      key -- gets its share of the chip (default) | 0 the same number of workgroups for every segment
 *   probe_slices       1 the probe side of a partition is handled in slices of 64 Ki rows, every slice beyond the first a work
 *                      item of its own: a hot foreign key no longer leaves one workgroup with most of the table (default) | 0 off
 *   broadcast_join     1 a right table of <= 6144 rows (one LDS hash table) is joined WITHOUT partitioning: every workgroup
 *                      builds the whole table and streams its share of the left columns; the result keeps the left row
 *                      order (default; only with the automatic radix plan) | 0 always partition
 *   ring_full_tiles    1 column input whose batches are all whole 4096-row tiles in 16-byte aligned buffers is read by a
 *                      loader without the ragged fallback (keeps the prefetch asynchronous; default) | 0 generic loader
 *   ring_split_ragged  1 pass 0 over ragged batches (or some unaligned buffers) runs as two launches on the same cursors: the
 *                      whole aligned tiles of every batch through the whole-tile loader, tails and unaligned batches through
 *                      the generic one (default) | 0 one launch with the generic loader
 *   ring_interleave    -1 automatic (default: pass 0's column tiles are dealt to the workgroups round-robin with the
 *                      whole-tile loader, as contiguous runs otherwise) | 0 contiguous runs | N round-robin in chunks of N tiles
 *   claim_weave        1 the claiming ring scatter weaves its regions in 64 KiB chunks, so that the lines the chip writes at
 *                      any moment lie within ~32 MiB (address-translation locality; default) | 0 regions one after the other
 *   presplit_sub       1 olap_join_presplit_u32 with ONE radix pass left and enough chunks to fill the chip: the pass runs chunk by
 *                      chunk without a histogram (one workgroup per (segment, sender) chunk, 6-byte tuples, every chunk a set of
 *                      sub-partitions of its own) and the probe builds a partition's table from the sub-partitions of all of its
 *                      segment's chunks; the rows then come (segment, chunk, bucket)-major (default) | 0 histogram + scan + scatter
 *                      over the chunks | 2 whenever the plan allows it, whatever the sizes (tests)
 *   presplit_refine    1 ... chunk places are cut into pieces until this many non-empty ones per CU exist (default 1; 1..8)
 *   join64_ring        1 olap_join_u64: two-pass plans of large inputs run without histograms on 12-byte tuples -- claiming ring
 *                      scatter from the columns, capacity-layout ring scatter, capacity-layout probe (csrc/ring64.h; default) |
 *                      0 always the exact-layout pipeline (histogram + scan + scatter per pass, 16-byte tuples) | 2 every plan of
 *                      8..18 radix bits (tests)
 *   join64_tpt         8 its passes of <= 256 buckets run 4096-tuple tiles (default) | 4 2048-tuple tiles, as passes of 512 buckets always do
 *   plan_first_bits    0 two-pass plans split their bits evenly, pass 0 taking at least log2(#CU) when the last pass keeps >= 6
 *                      (default) | N pass 0 takes N bits (experiments)
 *   probe_small_table  1 probing-table probe (k_join_probe): a 4096-slot LDS table, four workgroups per CU, when the build
 *                      partitions hold <= 2048 rows on average; a partition above 3072 rows repeats the plan with the 8192-slot
 *                      table (default) | 0 always 8192 slots, two workgroups per CU
 *   probe64_variant    5 the probe of that plan (k64_probe12): W + 4 * DH -- W = 1 | 2 tuples of a thread step through the table
 *                      together, DH = 1 double hashing (default) | 0 linear probing as in hashtable.c (the table is internal: rows
 *                      do not depend on it)
 *   probe64_grid       24 ... its workgroups per CU (1..64)
 *   probe6_prefetch    0 (default) | 1 the packed probe requests the next partition's sizes and first chunks ahead of the
 *                      current partition's phases (k_join_probe_direct6_pf: SF=256 probe 3.0 -> 2.85 ms, SF=2048 18.1 -> 17.8 ms)
 *   direct_table       1 direct-address LDS table when <= 14 hash bits remain (default) | 0 probing table
 *   take_variant       0 automatic | 1 never the LDS-window kernel | 3 small gather tiles | 4 window-bucketed path
 *                      for batches whose data buffer is far larger than LDS (an alternative that measured slower)
 *   host_stage         1 pinned ring + copy threads for OLAP_MEM_HOST operands (default) | 0 plain hipMemcpyAsync
 *   host_copy_threads  0 = min(hardware threads, 8) */
int olap_set_tuning(olap_ctx* ctx, const char* key, int value);
/* Diagnostics of the LAST join call on this context (the reference has no analog: its regions are sized by a
 * histogram round trip, host/join/join_dpu.cc:54-66).  Names:
 *   "join_attempts"          1 = the first plan went through; > 1 = it was redone (exact partitioning after a region
 *                            overflow, or more radix bits after an LDS-table overflow)
 *   "join_region_overflows"  (workgroup, bucket) regions of the histogram-free passes that overflowed in the first attempt
 *   "arena_base", "arena_bytes"  device address and size of the context's scratch arena (placement studies) */
int olap_get_counter(olap_ctx* ctx, const char* name, uint64_t* out_value);
/* Multi-GPU: when the inputs of this context were already split by the top
 * `bits` bits of wang_hash(key) (olap_partition_u32 with nr_partitions = #GPUs,
 * one partition per GPU), all keys here share those bits; the join's own radix
 * passes then start below them.  0 (default) = use the hash from its top bit. */
int olap_join_set_hash_skip_bits(olap_ctx* ctx, int bits);

/* ------------------------------------------------------------ device group
 * Several devices behind one handle: replaces dpu::DpuSet::allocate(n, profile) over MANY DPUs
 * (host/dpuext/dpuext.hpp:687-715) for the operator that needs them all at once, the join
 * (JoinDpu::Run_internal spans the whole set, host/join/join_dpu.cc:82-142,168-400).
 * n_ranks: a power of two in [1, 256]; device_ids[r] = HIP device of rank r.  The ids may repeat:
 * several logical ranks on one device is the loopback mode that runs the whole multi-device path on
 * a one-GPU box.  Single process; the group is single-caller like a context.  filter / take / sum
 * shard by batch without communication: run them per rank on olap_group_ctx(g, r). */
typedef struct olap_group olap_group;
int olap_group_create(int n_ranks, const int* device_ids, olap_group** out_group);
void olap_group_destroy(olap_group* g);
int olap_group_size(const olap_group* g);
olap_ctx* olap_group_ctx(olap_group* g, int rank);   /* the context of one rank (owned by the group) */
int olap_group_device(const olap_group* g, int rank);
const char* olap_group_last_error(const olap_group* g);
/* How the tuples travel between the ranks: RCCL (ncclSend / ncclRecv in one group call over a communicator made
 * by ncclCommInitAll -- needs one device per rank; librccl.so is opened on first use) or plain copies
 * (hipMemcpyPeerAsync over xGMI, same-device copies in loopback).  AUTO = RCCL when every rank has its own
 * device, else copies. */
enum { OLAP_GROUP_TRANSPORT_AUTO = 0, OLAP_GROUP_TRANSPORT_COPY = 1, OLAP_GROUP_TRANSPORT_RCCL = 2 };
int olap_group_set_transport(olap_group* g, int transport);
int olap_group_last_transport(const olap_group* g);   /* what the last join used (COPY or RCCL; 0: none yet) */
/* Wall milliseconds of the last olap_group_join_u32: {split, exchange, join, copy-out}. */
int olap_group_last_phase_ms(const olap_group* g, double out_ms[4]);
/* olap_join_u32 across the group.  The batches are dealt to the ranks (batch b -> rank b mod G, as the reference
 * deals batches to DPUs); with in_mem == OLAP_MEM_DEVICE batch b must live on the device of rank b mod G.  Every
 * rank splits its share by owner = top log2 G bits of wang_hash(key) (olap_split_tuples_u32), the runs are
 * exchanged -- the shuffle of host/partition/partitioner.cc:280-375 -- and every rank joins the keys it owns
 * (olap_join_presplit_u32).  Output: HOST columns, rank-major (rank r's rows at [sum of out_rank_rows[< r], ...)),
 * inside a rank partition-major like olap_join_u32.  out_rank_rows (G entries) may be NULL.  Both payload
 * columns are required.  The unmatched-row mode and tuning knobs are those of the ranks' contexts. */
int olap_group_join_u32(olap_group* g, const uint32_t* const* left_fk_ptrs, const uint32_t* const* left_y_ptrs,
                        const uint64_t* left_rows, uint64_t n_left_batches, const uint32_t* const* right_pk_ptrs,
                        const uint32_t* const* right_x_ptrs, const uint64_t* right_rows, uint64_t n_right_batches,
                        int in_mem, uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x, uint64_t out_capacity,
                        uint64_t* out_rows, uint64_t* out_rank_rows /* may be NULL */, olap_join_stats* stats /* may be NULL */);

/* --------------------------------------------------------------- generator
 * Synthetic columns with the distributions of host/generator/generator.cc:22-71
 * (uniform u32 payloads, pk = 0,1,2,..., fk of left batch i uniform in
 * [i*R, (i+1)*R-1]).  Counter based: element at global row r of stream s is
 *   mix32(seed, s, r)   (see dpu_olap_amd/datagen.py for the host mirror)
 * so SF=2048 columns are produced in HBM and any slice can be re-derived. */
int olap_gen_uniform_u32(olap_ctx* ctx, uint32_t* out_device, uint64_t n, uint64_t seed,
                         uint32_t stream_id, uint64_t first_row);
int olap_gen_uniform_u64(olap_ctx* ctx, uint64_t* out_device, uint64_t n, uint64_t seed,
                         uint32_t stream_id, uint64_t first_row);
int olap_gen_index_u32(olap_ctx* ctx, uint32_t* out_device, uint64_t n, uint32_t first_value);
/* out[r - first_row] = (r / left_batch_rows) * right_batch_rows
 *                      + mix32(seed, stream, r) mod right_batch_rows (multiply-high) */
int olap_gen_fk_u32(olap_ctx* ctx, uint32_t* out_device, uint64_t n, uint64_t seed,
                    uint32_t stream_id, uint64_t first_row, uint64_t left_batch_rows,
                    uint64_t right_batch_rows);
/* out[i] = mix32(seed, stream, rows[i])  -- the generator evaluated at given row ids */
int olap_gen_uniform_u32_at(olap_ctx* ctx, uint32_t* out_device, const uint32_t* rows_device,
                            uint64_t n, uint64_t seed, uint32_t stream_id);
/* Order-independent digest of a device column: sum, xor and sum of mix64 of the
 * elements, all mod 2^64 (out[0..2]).  elem_bytes 4 or 8. */
int olap_checksum(olap_ctx* ctx, const void* col_device, uint64_t n, int elem_bytes,
                  uint64_t out[3]);
/* Digest of the rows of a 3-column u32 table, independent of row order:
 * out[0] = sum_i mix64(a_i, b_i, c_i), out[1] = xor_i of the same. */
int olap_checksum_rows3_u32(olap_ctx* ctx, const uint32_t* a, const uint32_t* b,
                            const uint32_t* c, uint64_t n, uint64_t out[2]);

#ifdef __cplusplus
}
#endif
#endif /* OLAP_GPU_H_ */
