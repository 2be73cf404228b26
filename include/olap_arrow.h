/* olap_arrow.h -- Arrow C Data Interface front end of the C ABI in olap_gpu.h.
 *
 * The reference moves Arrow arrays to and from its device by their raw buffers
 * (arrow_copy_to_dpus / arrow_copy_from_dpus, host/dpuext/arrow_utils.cc:47-73,147-266) and
 * wraps device-filled buffers as ArrayData{nullptr, buffer} (host/filter/filter_dpu.cc:91-92).
 * These entry points do the same across the Arrow C Data Interface, so any Arrow
 * implementation (C++, pyarrow, Go cdata, arrow-rs FFI, Java C Data) can call the GPU
 * operators without copying and without linking Arrow C++ here:
 *   - inputs are struct ArrowArray of a primitive 32/64-bit integer type (format "I", "i",
 *     "L", "l"), without nulls (the reference assumes non-nullable columns,
 *     host/dpuext/arrow_utils.h:27-29); the arrays stay owned by the caller;
 *   - outputs are struct ArrowArray the library fills; the consumer owns them and calls
 *     release() when done (all outputs of one call share one allocation, freed with the last
 *     release).
 * The struct definitions below are the Arrow project's published ABI, guarded the standard way. */
#ifndef OLAP_ARROW_H_
#define OLAP_ARROW_H_

#include "olap_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE

#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4

struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};

struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};

#endif /* ARROW_C_DATA_INTERFACE */

/* filter: out_batches[b] (caller-provided structs, n_batches of them) receives the values of
 * batches[b] below *threshold, same type, in input order (olap_filter_lt). */
int olap_arrow_filter_lt(olap_ctx* ctx, const struct ArrowSchema* type, const struct ArrowArray* const* batches,
                         uint64_t n_batches, const void* threshold, struct ArrowArray* out_batches,
                         uint64_t* out_total /* may be NULL */);

/* sum over all batches (olap_sum); the 64-bit result bit pattern in *out_sum. */
int olap_arrow_sum(olap_ctx* ctx, const struct ArrowSchema* type, const struct ArrowArray* const* batches,
                   uint64_t n_batches, uint64_t* out_sum);

/* take: out_batches[b][i] = data[b][ indices[b][i] ], indices of type uint32 ("I"), no bounds
 * check (olap_take). */
int olap_arrow_take(olap_ctx* ctx, const struct ArrowSchema* data_type, const struct ArrowArray* const* data,
                    const struct ArrowArray* const* indices, uint64_t n_batches, struct ArrowArray* out_batches);

/* PK-FK join of left {fk, y} with right {pk, x}, all uint32 / int32 columns given as per-batch
 * arrays (olap_join_u32): out_columns[0..3) = fk, y, x as uint32 arrays of left-row length,
 * partition-major order.  left_y == NULL / right_x == NULL select the selection-vector form
 * (global row indices instead of payloads). */
int olap_arrow_join_u32(olap_ctx* ctx, const struct ArrowArray* const* left_fk, const struct ArrowArray* const* left_y,
                        uint64_t n_left_batches, const struct ArrowArray* const* right_pk,
                        const struct ArrowArray* const* right_x, uint64_t n_right_batches,
                        struct ArrowArray* out_columns /* 3 */, olap_join_stats* stats /* may be NULL */);

/* The same join on 64-BIT KEYS (olap_join_u64; the reference's HT_64BIT_KEYS table, dpu/shared/hashtable/hashtable.h:13-18):
 * fk / pk uint64 or int64 arrays, payloads uint32 / int32 arrays or NULL (selection vectors).  out_columns[0] = fk (64-bit
 * values: import it with the key column's type), out_columns[1..3) = y, x (32-bit). */
int olap_arrow_join_u64(olap_ctx* ctx, const struct ArrowArray* const* left_fk, const struct ArrowArray* const* left_y,
                        uint64_t n_left_batches, const struct ArrowArray* const* right_pk,
                        const struct ArrowArray* const* right_x, uint64_t n_right_batches,
                        struct ArrowArray* out_columns /* 3 */, olap_join_stats* stats /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* OLAP_ARROW_H_ */
