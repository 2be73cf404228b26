/*
 * olap_oracle.h -- CPU restatement of upmem/dpu_olap's data-parallel operator
 * kernels (filter, take, sum, radix partition, hash build/probe, join).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product path (dpu_olap_amd/, include/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / reported CPU baseline.
 *
 * Every function cites the reference file:line (relative to the upmem/dpu_olap
 * tree) whose semantics it restates.  The reference's device code targets the
 * UPMEM DPU toolchain (mram_read, barrier_wait, handshake_*, mutex pools) and
 * cannot be compiled here, so this is a restatement, pinned by the reference's
 * own known-answer tests (tests/golden/reference_kats.json) and by vectors
 * produced with Apache Arrow (the library the reference's "Native" path
 * delegates to) in tests/golden/.
 *
 * Element type: the reference hard-wires T = uint32_t
 * (dpu/shared/kernels/common.h:3-4) and sums into uint64_t
 * (dpu/aggr/main.c:44-50).  The 64-bit / signed variants below apply the same
 * algorithm to the other Arrow fixed-width integer types BASELINE.json names.
 */
#ifndef OLAP_ORACLE_H_
#define OLAP_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- hash / bucket: dpu/shared/kernels/partition.c:20-28,45-46;
 *      dpu/shared/hashtable/hashtable.c:29-37 (same function) ---- */
uint32_t oracle_wang_hash_u32(uint32_t key);
/* nr_partitions must be a power of two >= 2 (dpu/join/main.c:87-89). */
uint32_t oracle_bucket_of(uint32_t key, uint32_t nr_partitions);

/* ---- filter: dpu/shared/kernels/filter.c:25 (predicate), :87-95 (scan),
 *      :172-174 (out length).  Order preserved; returns the output length. ---- */
uint64_t oracle_filter_lt_u32(const uint32_t* in, uint64_t n, uint32_t threshold, uint32_t* out);
uint64_t oracle_filter_lt_i32(const int32_t* in, uint64_t n, int32_t threshold, int32_t* out);
uint64_t oracle_filter_lt_u64(const uint64_t* in, uint64_t n, uint64_t threshold, uint64_t* out);
uint64_t oracle_filter_lt_i64(const int64_t* in, uint64_t n, int64_t threshold, int64_t* out);

/* ---- take: dpu/shared/kernels/take.c:33-39 (out[i] = buffer[idx[i]], no
 *      bounds check, take.c:36 TODO) ---- */
void oracle_take_u32(const uint32_t* data, const uint32_t* idx, uint64_t n_idx, uint32_t* out);
void oracle_take_u64(const uint64_t* data, const uint32_t* idx, uint64_t n_idx, uint64_t* out);

/* ---- sum: dpu/aggr/main.c:44-50 (block sum into uint64_t), :76-88 (reduce of
 *      the per-tasklet partials); host add host/aggr/aggr_dpu.cc:82-84 ---- */
uint64_t oracle_sum_u32(const uint32_t* in, uint64_t n);
int64_t oracle_sum_i32(const int32_t* in, uint64_t n);
uint64_t oracle_sum_u64(const uint64_t* in, uint64_t n); /* wraps mod 2^64 */
int64_t oracle_sum_i64(const int64_t* in, uint64_t n);   /* wraps (two's complement) */

/* ---- per-batch radix partition: dpu/shared/kernels/partition.c:296-341
 *      (build_histogram :67-92, prefix_sum :94-137, partition_array :233-294,
 *      output rotation output_offset_shift :139-147).
 *
 * out_keys / out_rowid receive the keys and their original row indices grouped
 * by bucket; the whole output is rotated so that bucket `o_shift` starts at
 * index 0 (partition.c:316-321).  hist_end[b] is the END offset of bucket b in
 * the UN-rotated numbering, exactly what the host reads back as "metadata"
 * (host/partition/partitioner.cc:167-180, host/join/join_dpu.cc:54-66).
 * Inside a bucket the reference's order depends on tasklet interleaving; the
 * oracle uses input order (a valid schedule: one tasklet).
 * Returns 0, or -1 if nr_partitions is not a power of two >= 2. ---- */
int oracle_partition_u32(const uint32_t* keys, uint32_t n, uint32_t nr_partitions, uint32_t o_shift,
                         uint32_t* out_keys, uint32_t* out_rowid, uint32_t* hist_end);

/* ---- hash table: dpu/shared/hashtable/hashtable.c:63-78 (init), :89-165
 *      (put: linear probing, used-bitmap, equal key overwrites :133), :167-192
 *      (get).  n_buckets must be a power of two. ---- */
typedef struct {
  uint32_t key;
  uint32_t value;
} oracle_ht_entry_t;

typedef struct {
  oracle_ht_entry_t* items;
  uint32_t* used;
  uint32_t n_buckets;
} oracle_ht_t;

int oracle_ht_init(oracle_ht_t* t, uint32_t n_buckets);
void oracle_ht_free(oracle_ht_t* t);
int oracle_ht_put(oracle_ht_t* t, uint32_t key, uint32_t value); /* 1 ok, 0 full */
int oracle_ht_get(const oracle_ht_t* t, uint32_t key, uint32_t* value); /* 1 hit, 0 miss */

/* ---- hash build / probe: dpu/shared/kernels/hash_build.c:9-35 (value = index
 *      in the partition buffer), hash_probe.c:9-46 (sel[i] = table[fk[i]]; the
 *      reference asserts a hit, :32-33).  probe returns the number of misses
 *      (sel[i] = 0xFFFFFFFF on a miss). ---- */
int oracle_hash_build_u32(oracle_ht_t* t, const uint32_t* keys, uint32_t n);
uint64_t oracle_hash_probe_u32(const oracle_ht_t* t, const uint32_t* keys, uint32_t n, uint32_t* sel);

/* ---- join composition: host/join/join_dpu.cc:168-400 (+ do_partition :82-142,
 *      shuffle host/partition/partitioner.cc:280-375).
 *
 * left = {fk, y}, right = {pk, x}, given as n_batches batches per side (batch
 * pointer arrays + row counts).  nr_partitions plays the role of the
 * reference's "#partitions == #batches" (join_dpu.cc:93,168) but is a free
 * parameter here (power of two >= 2).  Phase 1 partitions every batch of both
 * sides (oracle_partition_u32), gathers the payload column by row id
 * (kernel_take reuse, partitioner.cc:185-207) and concatenates bucket p of all
 * batches in batch order (the shuffle).  Phase 2, per partition: build on pk,
 * probe with fk, take x.  Output rows {fk, y, x} are partition-major
 * (join_dpu.cc:314-363,371-399); out_* must hold total left rows.
 * Returns the number of output rows, or (uint64_t)-1 if some fk had no match
 * (reference asserts) or on bad arguments.  Runs batches / partitions in
 * parallel with OpenMP when built with -fopenmp; `threads` <= 0 means default. */
uint64_t oracle_join_u32(const uint32_t* const* left_fk, const uint32_t* const* left_y,
                         const uint32_t* left_rows, uint32_t n_left_batches,
                         const uint32_t* const* right_pk, const uint32_t* const* right_x,
                         const uint32_t* right_rows, uint32_t n_right_batches,
                         uint32_t nr_partitions, int threads, uint32_t* out_fk, uint32_t* out_y,
                         uint32_t* out_x);

/* ---- 64-bit keys: the hash table's HT_64BIT_KEYS mode, dpu/shared/hashtable/hashtable.h:13-18 (ht_key_t =
 *      uint64_t, ht_value_t stays uint32_t) with its hash hashtable.c:23-26:
 *      (ht_index_t)((key) >> 33 ^ (key) ^ (key) << 11).  Table semantics as above (hashtable.c:89-192). ---- */
uint32_t oracle_ht_hash_u64(uint64_t key);
/* Join on 64-bit keys: the composition of oracle_join_u32 with the 64-bit table.  The reference's partition kernel is
 * written for T = uint32_t (partition.c:20-28,45-46) and defines nothing for 64-bit keys; the restatement applies the
 * same BUCKET_OF to the table hash: bucket = wang_hash(ht_hash_u64(key)) >> (32 - log2 P).  Partitions keep input
 * order; rows {fk, y, x} partition-major.  Returns rows, or (uint64_t)-1 on a missing key / bad arguments. */
uint64_t oracle_join_u64(const uint64_t* const* left_fk, const uint32_t* const* left_y, const uint32_t* left_rows,
                         uint32_t n_left_batches, const uint64_t* const* right_pk, const uint32_t* const* right_x,
                         const uint32_t* right_rows, uint32_t n_right_batches, uint32_t nr_partitions, int threads,
                         uint64_t* out_fk, uint32_t* out_y, uint32_t* out_x);

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif /* OLAP_ORACLE_H_ */
