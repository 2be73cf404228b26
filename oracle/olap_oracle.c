/*
 * olap_oracle.c -- CPU restatement of upmem/dpu_olap's operator kernels.
 * TEST INFRASTRUCTURE ONLY (see olap_oracle.h).  Each function cites the
 * reference file:line it follows.
 */
#include "olap_oracle.h"

#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* dpu/shared/kernels/partition.c:20-28 == dpu/shared/hashtable/hashtable.c:29-37 */
uint32_t oracle_wang_hash_u32(uint32_t key) {
  key += ~(key << 15);
  key ^= (key >> 10);
  key += (key << 3);
  key ^= (key >> 6);
  key += ~(key << 11);
  key ^= (key >> 16);
  return key;
}

/* partition.c:45-46: BUCKET_SHIFT(n) = 1 + clz(n); BUCKET_OF(x) = wang(x) >> shift */
uint32_t oracle_bucket_of(uint32_t key, uint32_t nr_partitions) {
  uint32_t shift = 1u + (uint32_t)__builtin_clz(nr_partitions);
  return oracle_wang_hash_u32(key) >> shift;
}

/* filter.c:25 predicate `item < (1 << 30)` generalised to a threshold argument;
 * filter.c:87-95 scan appends passing items in input order; :172-174 length. */
#define ORACLE_FILTER(NAME, T)                                                 \
  uint64_t NAME(const T* in, uint64_t n, T threshold, T* out) {                \
    uint64_t w = 0;                                                            \
    for (uint64_t i = 0; i < n; ++i) {                                         \
      T item = in[i];                                                          \
      if (item < threshold) out[w++] = item;                                   \
    }                                                                          \
    return w;                                                                  \
  }
ORACLE_FILTER(oracle_filter_lt_u32, uint32_t)
ORACLE_FILTER(oracle_filter_lt_i32, int32_t)
ORACLE_FILTER(oracle_filter_lt_u64, uint64_t)
ORACLE_FILTER(oracle_filter_lt_i64, int64_t)

/* take.c:33-39 */
void oracle_take_u32(const uint32_t* data, const uint32_t* idx, uint64_t n_idx, uint32_t* out) {
  for (uint64_t i = 0; i < n_idx; ++i) out[i] = data[idx[i]];
}
void oracle_take_u64(const uint64_t* data, const uint32_t* idx, uint64_t n_idx, uint64_t* out) {
  for (uint64_t i = 0; i < n_idx; ++i) out[i] = data[idx[i]];
}

/* dpu/aggr/main.c:44-50: `uint64_t sum; sum += items[i]` */
uint64_t oracle_sum_u32(const uint32_t* in, uint64_t n) {
  uint64_t s = 0;
  for (uint64_t i = 0; i < n; ++i) s += in[i];
  return s;
}
int64_t oracle_sum_i32(const int32_t* in, uint64_t n) {
  int64_t s = 0;
  for (uint64_t i = 0; i < n; ++i) s += in[i];
  return s;
}
uint64_t oracle_sum_u64(const uint64_t* in, uint64_t n) {
  uint64_t s = 0;
  for (uint64_t i = 0; i < n; ++i) s += in[i];
  return s;
}
int64_t oracle_sum_i64(const int64_t* in, uint64_t n) {
  uint64_t s = 0; /* wrap like Arrow's sum(int64): two's complement */
  for (uint64_t i = 0; i < n; ++i) s += (uint64_t)in[i];
  return (int64_t)s;
}

/* partition.c:296-341 */
int oracle_partition_u32(const uint32_t* keys, uint32_t n, uint32_t P, uint32_t o_shift,
                         uint32_t* out_keys, uint32_t* out_rowid, uint32_t* hist_end) {
  if (P < 2 || (P & (P - 1)) != 0 || o_shift >= P) return -1; /* dpu/join/main.c:87-89 */
  uint32_t* histogram = (uint32_t*)calloc((size_t)P + 1, sizeof(uint32_t)); /* partition.c:303-305 */
  if (!histogram) return -1;
  /* build_histogram, partition.c:67-92: histogram[bucket+1]++ */
  for (uint32_t i = 0; i < n; ++i) histogram[oracle_bucket_of(keys[i], P) + 1]++;
  /* prefix_sum, partition.c:130-135: inclusive scan -> histogram[b] = start of bucket b */
  for (uint32_t i = 1; i < P + 1; ++i) histogram[i] += histogram[i - 1];
  /* partition.c:316-321: output_shift = histogram[o_shift]; nb_elem_input = n */
  uint32_t output_shift = histogram[o_shift];
  /* partition_array, partition.c:233-294: slot = histogram[bucket]++, rotated
   * by output_offset_shift (partition.c:139-147); the WRAM pairing cache
   * (partition.c:180-231) only batches the two 4-byte stores into one 8-byte
   * DMA and does not change where an element lands. */
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t b = oracle_bucket_of(keys[i], P);
    uint32_t off = histogram[b]++;
    uint32_t shifted = (off < output_shift) ? (off - output_shift + n) : (off - output_shift);
    out_keys[shifted] = keys[i];
    if (out_rowid) out_rowid[shifted] = i;
  }
  /* after the scatter histogram[b] is the END offset of bucket b (join_dpu.cc:54-66) */
  memcpy(hist_end, histogram, (size_t)P * sizeof(uint32_t));
  free(histogram);
  return 0;
}

/* hashtable.c:63-78 */
int oracle_ht_init(oracle_ht_t* t, uint32_t n_buckets) {
  if (n_buckets == 0 || (n_buckets & (n_buckets - 1)) != 0) return -1;
  t->n_buckets = n_buckets;
  t->items = (oracle_ht_entry_t*)malloc((size_t)n_buckets * sizeof(oracle_ht_entry_t));
  size_t words = n_buckets < 32 ? 1 : (n_buckets >> 5); /* __ht_used_size, hashtable.c:21 */
  t->used = (uint32_t*)calloc(words, sizeof(uint32_t));
  if (!t->items || !t->used) return -1;
  return 0;
}

void oracle_ht_free(oracle_ht_t* t) {
  free(t->items);
  free(t->used);
  t->items = NULL;
  t->used = NULL;
}

/* hashtable.c:89-165 */
int oracle_ht_put(oracle_ht_t* t, uint32_t key, uint32_t value) {
  uint32_t k = oracle_wang_hash_u32(key);
  uint32_t mask = t->n_buckets - 1;
  uint32_t i = k & mask; /* hashtable.c:99-101 */
  uint32_t last = i;
  for (;;) {
    int used = (t->used[i >> 5] >> (i & 0x1fU)) & 1U;
    if (!used || t->items[i].key == key) { /* free cell, or equal key overwrites (:133) */
      t->used[i >> 5] |= 1U << (i & 0x1fU);
      t->items[i].key = key;
      t->items[i].value = value;
      return 1;
    }
    i = (i + 1U) & mask; /* hashtable.c:136 */
    if (i == last) return 0; /* table full, :137-139 */
  }
}

/* hashtable.c:167-192 */
int oracle_ht_get(const oracle_ht_t* t, uint32_t key, uint32_t* value) {
  uint32_t k = oracle_wang_hash_u32(key);
  uint32_t mask = t->n_buckets - 1;
  uint32_t i = k & mask;
  uint32_t last = i;
  do {
    int used = (t->used[i >> 5] >> (i & 0x1fU)) & 1U;
    if (used && t->items[i].key == key) {
      *value = t->items[i].value;
      return 1;
    }
    i = (i + 1U) & mask;
  } while (i != last);
  return 0;
}

/* hash_build.c:9-35: ht_put(item, index) */
int oracle_hash_build_u32(oracle_ht_t* t, const uint32_t* keys, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i)
    if (!oracle_ht_put(t, keys[i], i)) return -1;
  return 0;
}

/* hash_probe.c:9-46 */
uint64_t oracle_hash_probe_u32(const oracle_ht_t* t, const uint32_t* keys, uint32_t n, uint32_t* sel) {
  uint64_t misses = 0;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t index;
    if (oracle_ht_get(t, keys[i], &index)) {
      sel[i] = index;
    } else {
      sel[i] = 0xFFFFFFFFu;
      ++misses;
    }
  }
  return misses;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static uint32_t next_pow2_u32(uint32_t x) {
  uint32_t p = 1;
  while (p < x && p < 0x80000000u) p <<= 1;
  return p;
}

/* One side of phase 1 (join_dpu.cc:82-142): per batch kernel_partition on the
 * key column, kernel_take of the payload by row id (partitioner.cc:185-207),
 * then the shuffle (GetOffsets :280-312 + LoadPartitions :350-375): bucket p
 * of batch b goes to Partition[p] at offset sum_{b'<b} count[b'][p]. */
typedef struct {
  uint32_t* key;      /* partition-major keys */
  uint32_t* payload;  /* partition-major payload */
  uint64_t* start;    /* [P+1] partition starts */
} side_parts_t;

static int partition_side(const uint32_t* const* key_cols, const uint32_t* const* pay_cols,
                          const uint32_t* rows, uint32_t nb, uint32_t P, side_parts_t* out) {
  uint64_t total = 0;
  for (uint32_t b = 0; b < nb; ++b) total += rows[b];
  /* counts[b][p] from the per-batch END-offset metadata (join_dpu.cc:54-66) */
  uint32_t* meta = (uint32_t*)malloc((size_t)nb * P * sizeof(uint32_t));
  uint32_t** pk = (uint32_t**)calloc(nb, sizeof(uint32_t*));
  uint32_t** pp = (uint32_t**)calloc(nb, sizeof(uint32_t*));
  out->key = (uint32_t*)malloc((size_t)(total ? total : 1) * sizeof(uint32_t));
  out->payload = (uint32_t*)malloc((size_t)(total ? total : 1) * sizeof(uint32_t));
  out->start = (uint64_t*)calloc((size_t)P + 1, sizeof(uint64_t));
  int err = 0;
  if (!meta || !pk || !pp || !out->key || !out->payload || !out->start) return -1;

#pragma omp parallel for schedule(dynamic, 1)
  for (uint32_t b = 0; b < nb; ++b) {
    uint32_t n = rows[b];
    uint32_t* ok = (uint32_t*)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t* orow = (uint32_t*)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t* opay = (uint32_t*)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    if (!ok || !orow || !opay ||
        oracle_partition_u32(key_cols[b], n, P, 0, ok, orow, meta + (size_t)b * P) != 0) {
#pragma omp atomic write
      err = 1;
    } else {
      oracle_take_u32(pay_cols[b], orow, n, opay); /* partitioner.cc:185-207 */
    }
    free(orow);
    pk[b] = ok;
    pp[b] = opay;
  }
  if (!err) {
    /* partition totals -> starts */
    for (uint32_t p = 0; p < P; ++p) {
      uint64_t tot = 0;
      for (uint32_t b = 0; b < nb; ++b) {
        uint32_t lo = p ? meta[(size_t)b * P + p - 1] : 0;
        tot += meta[(size_t)b * P + p] - lo;
      }
      out->start[p + 1] = out->start[p] + tot;
    }
#pragma omp parallel for schedule(static)
    for (uint32_t p = 0; p < P; ++p) {
      uint64_t w = out->start[p];
      for (uint32_t b = 0; b < nb; ++b) {
        uint32_t lo = p ? meta[(size_t)b * P + p - 1] : 0;
        uint32_t cnt = meta[(size_t)b * P + p] - lo;
        memcpy(out->key + w, pk[b] + lo, (size_t)cnt * sizeof(uint32_t));
        memcpy(out->payload + w, pp[b] + lo, (size_t)cnt * sizeof(uint32_t));
        w += cnt;
      }
    }
  }
  for (uint32_t b = 0; b < nb; ++b) {
    free(pk[b]);
    free(pp[b]);
  }
  free(pk);
  free(pp);
  free(meta);
  return err ? -1 : 0;
}

static void free_side(side_parts_t* s) {
  free(s->key);
  free(s->payload);
  free(s->start);
}

/* host/join/join_dpu.cc:168-400 */
uint64_t oracle_join_u32(const uint32_t* const* left_fk, const uint32_t* const* left_y,
                         const uint32_t* left_rows, uint32_t n_left_batches,
                         const uint32_t* const* right_pk, const uint32_t* const* right_x,
                         const uint32_t* right_rows, uint32_t n_right_batches, uint32_t P,
                         int threads, uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x) {
  if (P < 2 || (P & (P - 1)) != 0) return (uint64_t)-1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
  side_parts_t L, R;
  memset(&L, 0, sizeof L);
  memset(&R, 0, sizeof R);
  /* phase 1: both tables (join_dpu.cc:213-220) */
  if (partition_side(left_fk, left_y, left_rows, n_left_batches, P, &L) != 0 ||
      partition_side(right_pk, right_x, right_rows, n_right_batches, P, &R) != 0) {
    free_side(&L);
    free_side(&R);
    return (uint64_t)-1;
  }
  uint64_t total_left = L.start[P];
  uint64_t misses = 0;
  int err = 0;
  /* phase 2: per partition build / probe / take (join_dpu.cc:254-369) */
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : misses)
  for (uint32_t p = 0; p < P; ++p) {
    uint64_t r0 = R.start[p], rn = R.start[p + 1] - r0;
    uint64_t l0 = L.start[p], ln = L.start[p + 1] - l0;
    if (ln == 0) continue;
    if (rn > 0x7FFFFFFFu || ln > 0xFFFFFFFFu) {
#pragma omp atomic write
      err = 1;
      continue;
    }
    oracle_ht_t ht;
    uint32_t cap = next_pow2_u32((uint32_t)(rn * 2 < 64 ? 64 : rn * 2)); /* load <= 0.5 (dpu/join/main.c:23,29) */
    uint32_t* sel = (uint32_t*)malloc((size_t)ln * sizeof(uint32_t));
    if (!sel || oracle_ht_init(&ht, cap) != 0 ||
        oracle_hash_build_u32(&ht, R.key + r0, (uint32_t)rn) != 0) { /* join_dpu.cc:256-277 */
#pragma omp atomic write
      err = 1;
      free(sel);
      continue;
    }
    uint64_t m = oracle_hash_probe_u32(&ht, L.key + l0, (uint32_t)ln, sel); /* :280-301 */
    misses += m;
    if (m == 0) {
      /* left partition columns move into the output (:315-322), right payload
       * taken by the selection vector (:332-362) */
      memcpy(out_fk + l0, L.key + l0, (size_t)ln * sizeof(uint32_t));
      memcpy(out_y + l0, L.payload + l0, (size_t)ln * sizeof(uint32_t));
      oracle_take_u32(R.payload + r0, sel, ln, out_x + l0);
    }
    oracle_ht_free(&ht);
    free(sel);
  }
  free_side(&L);
  free_side(&R);
  if (err || misses) return (uint64_t)-1;
  return total_left;
}


/* ------------------------------------------------------------------ 64-bit keys
 * hashtable.c:23-26 (HT_64BIT_KEYS) */
uint32_t oracle_ht_hash_u64(uint64_t key) { return (uint32_t)((key >> 33) ^ key ^ (key << 11)); }

typedef struct {
  uint64_t* key;
  uint32_t* payload;
  uint64_t* start; /* [P+1] */
} side_parts64_t;

/* phase 1 for one side: bucket of every row (partition.c:45-46 on the table hash), stable counting sort over all
 * batches in batch order (what per-batch kernel_partition + the shuffle of partitioner.cc:280-375 amount to) */
static int partition_side64(const uint64_t* const* key_cols, const uint32_t* const* pay_cols, const uint32_t* rows, uint32_t nb,
                            uint32_t P, side_parts64_t* out) {
  uint64_t total = 0;
  for (uint32_t b = 0; b < nb; ++b) total += rows[b];
  int lg = 0;
  while ((1u << lg) < P) ++lg;
  out->key = (uint64_t*)malloc((size_t)(total ? total : 1) * sizeof(uint64_t));
  out->payload = (uint32_t*)malloc((size_t)(total ? total : 1) * sizeof(uint32_t));
  out->start = (uint64_t*)calloc((size_t)P + 1, sizeof(uint64_t));
  uint64_t* cur = (uint64_t*)calloc((size_t)P, sizeof(uint64_t));
  if (!out->key || !out->payload || !out->start || !cur) return -1;
  for (uint32_t b = 0; b < nb; ++b)
    for (uint32_t i = 0; i < rows[b]; ++i)
      out->start[(oracle_wang_hash_u32(oracle_ht_hash_u64(key_cols[b][i])) >> (32 - lg)) + 1]++;
  for (uint32_t p = 0; p < P; ++p) out->start[p + 1] += out->start[p];
  memcpy(cur, out->start, (size_t)P * sizeof(uint64_t));
  for (uint32_t b = 0; b < nb; ++b)
    for (uint32_t i = 0; i < rows[b]; ++i) {
      const uint32_t p = oracle_wang_hash_u32(oracle_ht_hash_u64(key_cols[b][i])) >> (32 - lg);
      out->key[cur[p]] = key_cols[b][i];
      out->payload[cur[p]] = pay_cols[b][i];
      cur[p]++;
    }
  free(cur);
  return 0;
}

uint64_t oracle_join_u64(const uint64_t* const* left_fk, const uint32_t* const* left_y, const uint32_t* left_rows,
                         uint32_t n_left_batches, const uint64_t* const* right_pk, const uint32_t* const* right_x,
                         const uint32_t* right_rows, uint32_t n_right_batches, uint32_t P, int threads, uint64_t* out_fk,
                         uint32_t* out_y, uint32_t* out_x) {
  if (P < 2 || (P & (P - 1)) != 0) return (uint64_t)-1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
  side_parts64_t L, R;
  memset(&L, 0, sizeof L);
  memset(&R, 0, sizeof R);
  uint64_t misses = 0;
  int err = 0;
  if (partition_side64(left_fk, left_y, left_rows, n_left_batches, P, &L) != 0 ||
      partition_side64(right_pk, right_x, right_rows, n_right_batches, P, &R) != 0)
    err = 1;
  const uint64_t total_left = err ? 0 : L.start[P];
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : misses)
  for (uint32_t p = 0; p < (err ? 0 : P); ++p) {
    const uint64_t r0 = R.start[p], rn = R.start[p + 1] - r0, l0 = L.start[p], ln = L.start[p + 1] - l0;
    if (ln == 0) continue;
    const uint32_t cap = next_pow2_u32((uint32_t)(rn * 2 < 64 ? 64 : rn * 2)), mask = cap - 1;
    uint64_t* tk = (uint64_t*)malloc((size_t)cap * sizeof(uint64_t));
    uint32_t* tv = (uint32_t*)malloc((size_t)cap * sizeof(uint32_t));
    uint32_t* used = (uint32_t*)calloc(cap < 32 ? 1 : cap >> 5, sizeof(uint32_t));
    if (!tk || !tv || !used) {
#pragma omp atomic write
      err = 1;
    } else {
      for (uint64_t i = 0; i < rn; ++i) { /* ht_put, hashtable.c:89-165: value = right payload (take fused) */
        const uint64_t key = R.key[r0 + i];
        uint32_t s = oracle_ht_hash_u64(key) & mask;
        while (((used[s >> 5] >> (s & 31)) & 1u) && tk[s] != key) s = (s + 1) & mask;
        used[s >> 5] |= 1u << (s & 31);
        tk[s] = key;
        tv[s] = R.payload[r0 + i];
      }
      for (uint64_t i = 0; i < ln; ++i) { /* ht_get, hashtable.c:167-192 */
        const uint64_t key = L.key[l0 + i];
        uint32_t s = oracle_ht_hash_u64(key) & mask, first = s;
        int hit = 0;
        do {
          if (!((used[s >> 5] >> (s & 31)) & 1u)) break;
          if (tk[s] == key) {
            hit = 1;
            break;
          }
          s = (s + 1) & mask;
        } while (s != first);
        if (!hit) {
          ++misses;
          continue;
        }
        out_fk[l0 + i] = key;
        out_y[l0 + i] = L.payload[l0 + i];
        out_x[l0 + i] = tv[s];
      }
    }
    free(tk);
    free(tv);
    free(used);
  }
  free(L.key); free(L.payload); free(L.start);
  free(R.key); free(R.payload); free(R.start);
  if (err || misses) return (uint64_t)-1;
  return total_left;
}
