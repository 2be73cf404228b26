// join.hip -- host side of the radix partition and the radix-partitioned PK-FK hash join
// (olap_partition_*, olap_join_*): plans the passes, sizes the arena, launches the kernels of
// radix_kernels.h / probe_kernels.h, handles the retries.
//
// Reference behaviour (upmem/dpu_olap): the shuffle host/partition/partitioner.cc:280-375,
// the join composition host/join/join_dpu.cc:168-400 (partition both sides, then per
// partition build / probe / take).
//
// Plan: (key, payload) tuples go through 1..4 MSD radix passes over wang_hash(key) until a
// partition's build side fits an LDS table -- the direct-address table (<= 14 hash bits left,
// 16 Ki rows per partition; two 9-bit passes at 2^32 rows) or the probing table (8192 slots).
// Pass 0 reads the Arrow columns, as a claiming scatter without histogram when the plan has
// two passes and the input is large; the last pass uses the capacity layout (no histogram)
// when every workgroup owns whole segments; otherwise hist + scan + scatter (exact layout).
// Overflowing regions or tables are detected on the device and answered by a rerun with
// exact passes / more radix bits.  Bound: HBM bandwidth (integer work, no MFMA).
#include "common.h"
#include "probe_kernels.h"
#include "radix_kernels.h"
#include "ring_scatter.h"
#include "ring_claim.h"
#include "pack6.h"
#include "radix64.h"
#include "ring64.h"

namespace olap {

// ------------------------------------------------------------------ host
struct PassPlan {
  int passes = 0;
  uint32_t bits[kMaxPasses] = {0, 0, 0, 0};
  uint32_t total_bits = 0;
};

static int ceil_log2_u64(uint64_t x) {
  int l = 0;
  while ((1ull << l) < x) ++l;
  return l;
}

static int make_plan(olap_ctx* ctx, uint32_t total_bits, PassPlan* plan) {
  int maxb = ctx->max_bits_per_pass > 0 ? ctx->max_bits_per_pass : kMaxFBits;
  if (maxb > kMaxFBits) maxb = kMaxFBits;
  if (total_bits < 1) total_bits = 1;
  const int passes = ((int)total_bits + maxb - 1) / maxb;
  if (passes > kMaxPasses)
    return ctx->fail(OLAP_ERR_INVALID, "%u radix bits need %d passes (max %d)", total_bits, passes, kMaxPasses);
  plan->passes = passes;
  plan->total_bits = total_bits;
  const uint32_t base = total_bits / passes, extra = total_bits % passes;
  for (int i = 0; i < passes; ++i) plan->bits[i] = base + ((uint32_t)i < extra ? 1u : 0u);
  // The histogram-free passes need one input segment per CU for the last pass (capacity_eligible): 14 bits split 8 + 6 instead
  // of 7 + 7 (SF=16: 1.28 -> 1.08 ms, SF=12: 1.03 -> 0.86 ms; a last pass of 5 bits overflows its regions: SF=8 stays 7 + 6)
  const uint32_t cu_bits = (uint32_t)ceil_log2_u64((uint64_t)(ctx->compute_units > 1 ? ctx->compute_units : 2));
  if (passes == 2 && plan->bits[0] < cu_bits && (int)cu_bits <= maxb && total_bits >= cu_bits + 6) {
    plan->bits[0] = cu_bits;
    plan->bits[1] = total_bits - cu_bits;
  }
  if (passes == 2 && ctx->plan_first_bits > 0 && (uint32_t)ctx->plan_first_bits < total_bits && ctx->plan_first_bits <= maxb &&
      (int)total_bits - ctx->plan_first_bits <= maxb) {  // (experiments: an uneven split)
    plan->bits[0] = (uint32_t)ctx->plan_first_bits;
    plan->bits[1] = total_bits - plan->bits[0];
  }
  return OLAP_OK;
}

struct SideBuffers {
  uint2* final_tuples = nullptr;    // rows + 2
  uint64_t* final_off = nullptr;    // [P + 1]
};

static uint64_t seg_blocks(const olap_ctx* ctx, uint64_t n_seg) {
  uint64_t want = (uint64_t)ctx->compute_units * 4;
  uint64_t b = (want + n_seg - 1) / n_seg;
  if (b < 1) b = 1;
  if (b > 512) b = 512;  // two workgroups per CU when a single segment feeds the whole chip
  return b;
}

// Tuples that arrive already grouped by their leading hash bits (below the skipped ones):
// segment j fixes `bits` = log2(n_seg) bits and is the union of `chunks` chunks
// [lo[j*chunks + c], hi[j*chunks + c]) of the tuple array -- one chunk per sender after
// the fused owner + sub-bucket exchange.  Host arrays.
// Bucket order of a single-pass split: hash fields [owner | group | low] emitted as [group | owner | low].
struct BucketOrder {
  uint32_t owner_bits = 0, group_bits = 0, low_bits = 0;
};

struct Presplit {
  uint32_t n_seg = 0;
  uint32_t chunks = 0;
  uint32_t bits = 0;
  const uint64_t* lo = nullptr;
  const uint64_t* hi = nullptr;
};

// workspace bytes one side needs besides its final tuple buffer and the shared temps
static uint64_t side_scratch_bytes(const olap_ctx* ctx, uint64_t rows, uint64_t n_batches, const PassPlan& plan,
                                   const Presplit* pre = nullptr) {
  uint64_t bytes = 0;
  uint64_t n_seg = pre ? pre->n_seg : 1;
  // upper bound of pass-0 tiles: the smaller tile geometry, every batch rounded up on its own
  const uint64_t tiles = rows / 4096 + n_batches + 1;
  if (pre) bytes += align_up((uint64_t)pre->n_seg * pre->chunks * 16, 256) + align_up((n_seg + 1) * 8, 256);
  for (int i = 0; i < plan.passes; ++i) {
    const uint64_t F = 1ull << plan.bits[i];
    uint64_t blocks;
    if (i == 0 && pre) {
      blocks = n_seg * pre->chunks;
      if (blocks < (uint64_t)ctx->compute_units * 4) blocks = (uint64_t)ctx->compute_units * 4;  // chunks read by several workgroups
    } else if (i == 0) {
      blocks = (uint64_t)ctx->compute_units * 2;
      if (blocks > tiles) blocks = tiles;
    } else {
      blocks = 2 * n_seg * seg_blocks(ctx, n_seg) + 4 * n_seg + 8;  // (sized segments: an upper bound of the workgroups handed out)
      bytes += align_up((n_seg + 1) * 4, 256);
    }
    bytes += align_up(blocks * F * 4, 256);          // hist matrix
    bytes += align_up((n_seg * F + 1) * 8, 256);     // segment table
    n_seg *= F;
  }
  bytes += 2 * align_up(n_seg * 4, 256) + align_up(n_seg * 8, 256);  // capacity-layout tables (+ the packed layout's hash bits)
  if (pre && plan.passes == 1) {  // ... per chunk when the pass runs chunk by chunk (presplit_sub_eligible)
    const uint64_t subs = (uint64_t)pre->n_seg * pre->chunks << plan.bits[0];
    bytes += 2 * align_up(subs * 4, 256) + align_up(subs * 8, 256) + align_up((uint64_t)pre->n_seg * pre->chunks * 8, 256);
  }
  bytes += align_up((1ull << plan.bits[0]) * 4, 256) + align_up((1ull << plan.bits[0]) * 8, 256);  // claim cursors, segment totals
  return bytes + 4096;
}

// Runs all passes of one side.  `cs` views the key/payload columns with unit_rows = kPTile.
struct SideOut {
  uint64_t* off = nullptr;   // exact layout: [P + 1]
  uint32_t cap = 0;          // capacity layout: tuples per region (0 = exact)
  uint32_t* cnt = nullptr;   // capacity layout: [P]
  uint64_t* dense = nullptr; // capacity layout: [P]
  bool packed = false;       // capacity layout of 6-byte tuples (pack6.h)
  uint32_t* top = nullptr;   // packed: [P] hash bits above the table bits
  uint32_t nsub = 0;         // packed, pre-split input taken chunk by chunk: sub-partitions per partition (PartView::nsub; 0: plain)
  uint32_t sub_fbits = 0;
};

// Region size of the capacity layout: the average partition plus the larger of 1/8 and six standard deviations of a
// partition whose keys repeat like a Poisson(1) foreign key (variance 2 * avg) -- the second term matters for small
// partitions (avg 512: 4 sigma = 128 was overrun by a few of 2^16 partitions on uniform data).
static uint32_t region_cap(uint64_t rows, uint64_t P) {
  const uint64_t avg = (rows + P - 1) / P;
  uint64_t sd = 1;
  while (sd * sd < 2 * avg) ++sd;
  const uint64_t slack = avg / 8 > 6 * sd ? avg / 8 : 6 * sd;
  return (uint32_t)align_up(avg + slack + 64, kLineAlign);
}

// ... of a side the last pass writes as 6-byte tuples (pack6.h): whole 128-byte lines per region
static uint32_t region_cap_side(uint64_t rows, uint64_t P, bool packed) {
  const uint32_t cap = region_cap(rows, P);
  return packed ? (uint32_t)align_up(cap, kPackAlign) : cap;
}

// Pre-split input (what a rank holds after the all-to-all) whose ONE remaining pass has too few segments for the capacity
// layout's one workgroup per segment: the pass runs chunk by chunk instead -- one workgroup per (segment, sender) chunk, every chunk
// a set of F packed sub-partitions of its own (PassSrc::out_by_block) -- and the probe builds a partition's table from the
// sub-partitions of all of its segment's chunks (k_join_probe_direct6_sub).  No histogram, 6-byte tuples, as in the
// single-GPU plan.  presplit_sub: 0 never | 1 (default) when the chunks fill the chip | 2 whenever the plan allows (tests).
static uint64_t max_chunk_rows(const Presplit* pre) {
  uint64_t m = 0;
  for (uint64_t k = 0; k < (uint64_t)pre->n_seg * pre->chunks; ++k) m = std::max(m, pre->hi[k] - pre->lo[k]);
  return m;
}
bool ring_ok(const olap_ctx* ctx, uint32_t fbits, int role);
static bool presplit_sub_plan_ok(const olap_ctx* ctx, const Presplit* lpre, const Presplit* rpre, const PassPlan& plan, uint32_t table_bits,
                                 bool direct) {
  if (!ctx->presplit_sub || !lpre || !rpre || plan.passes != 1 || !direct || !ctx->pack6 || !ctx->capacity_mode || ctx->scatter_variant != 0)
    return false;
  const uint32_t fbits = plan.bits[0];
  return fbits >= 4 && ring_ok(ctx, fbits, 2) && table_bits >= 1 && table_bits <= 16 && lpre->n_seg == rpre->n_seg;
}
static uint64_t nonempty_chunks(const Presplit* pre) {
  uint64_t n = 0;
  for (uint64_t k = 0; k < (uint64_t)pre->n_seg * pre->chunks; ++k) n += pre->hi[k] > pre->lo[k];
  return n;
}
// Too few chunks to give every CU one (a table that travelled in one slice, few senders): every chunk place is cut into k
// pieces -- chunk places are just tuple ranges.  k = the power of two that brings the non-empty places to one per CU (presplit_refine), while a
// piece still leaves its sub-partitions >= 128 tuples.
static uint32_t presplit_refine_factor(const olap_ctx* ctx, const Presplit* pre, uint32_t fbits) {
  const uint64_t ne = std::max<uint64_t>(nonempty_chunks(pre), 1), want = (uint64_t)ctx->compute_units * (uint64_t)std::max(1, ctx->presplit_refine), rows = max_chunk_rows(pre);
  uint32_t k = 1;
  while (ne * k < want && k < 16 && ((rows / (2ull * k)) >> fbits) >= 128) k *= 2;
  return k;
}
static void refine_presplit(const Presplit& src, uint32_t k, std::vector<uint64_t>* lo, std::vector<uint64_t>* hi, Presplit* out) {
  const uint64_t n = (uint64_t)src.n_seg * src.chunks;
  lo->resize(n * k);
  hi->resize(n * k);
  for (uint64_t c = 0; c < n; ++c) {
    const uint64_t L = src.lo[c], H = src.hi[c];
    const uint64_t piece = align_up((H - L + k - 1) / k, 2);  // (pieces start on even tuples: 16-byte loads)
    for (uint32_t j = 0; j < k; ++j) {
      const uint64_t a = std::min(L + j * piece, H);
      (*lo)[c * k + j] = a;
      (*hi)[c * k + j] = std::min(a + piece, H);
    }
  }
  *out = src;
  out->chunks = src.chunks * k;
  out->lo = lo->data();
  out->hi = hi->data();
}
// ... and whether the (refined) chunks are worth it: every CU gets one, sub-partitions of a useful size
static bool presplit_sub_sizes_ok(const olap_ctx* ctx, const Presplit* pre, uint32_t fbits) {
  if (pre->chunks > (uint32_t)(kSubBlocks * kWave)) return false;  // (the probe holds a partition's sub-partitions one per lane)
  if (ctx->presplit_sub >= 2) return true;
  return nonempty_chunks(pre) >= (uint64_t)ctx->compute_units && (max_chunk_rows(pre) >> fbits) >= 128;
}

// The last pass may skip its histogram (capacity layout) when every workgroup owns whole
// input segments and there are enough of them to fill the chip.
static bool capacity_eligible(const olap_ctx* ctx, uint64_t rows, const PassPlan& plan, uint32_t pre_bits = 0) {
  if (!ctx->capacity_mode || ctx->scatter_variant != 0 || plan.passes < 2) return false;
  const uint64_t P = 1ull << (plan.total_bits + pre_bits);
  const uint64_t n_seg_last = P >> plan.bits[plan.passes - 1];
  return n_seg_last >= (uint64_t)ctx->compute_units && rows / P >= 512;
}

// Region size of the claiming scatter: the average bucket, 8 sigma of slack (uniform hash),
// and what the workgroups may leave unused in their last allotments.  0: does not fit 32 bits.
bool ring_ok(const olap_ctx* ctx, uint32_t fbits, int role);
static uint32_t claim_cap(const olap_ctx* ctx, uint64_t rows, uint64_t F) {
  // allotments a pass-0 workgroup may leave unused per bucket, times the workgroups: one for the write-combining kernel
  // (two workgroups per CU), up to three for the ring scatter (the one in use, the reserve, a claim in flight; one per CU)
  const bool ring = ring_ok(ctx, (uint32_t)ceil_log2_u64(F), 1);
  // (ring: twice that, because ragged column input is scattered by TWO launches sharing the cursors -- whole aligned
  // tiles through the straight-line loader, the batches' tails through the generic one)
  const uint64_t kb0 = (uint64_t)ctx->compute_units * (ring ? 6 : 2);
  const uint64_t kb = kb0 > 256 ? kb0 : 256;
  const uint64_t avg = (rows + F - 1) / F;
  uint64_t sd = 1;
  while (sd * sd < avg) ++sd;
  // woven regions (ring scatter, radix_kernels.h woven_at): whole chunks
  const uint64_t cap = align_up(avg + avg / 64 + 8 * sd + (kb + 2) * kClaimChunk, ring && ctx->claim_weave ? (1ull << kWeaveShift) : kClaimChunk);
  return cap >= (1ull << 32) ? 0u : (uint32_t)cap;
}

// Pass 0 of a two-pass plan may run as a claiming scatter (no histogram) when the last pass
// uses the capacity layout; automatic when the worst-case padding stays below half of the rows (>= ~2^26 rows: SF=128 7.6 -> 6.4 ms, SF=32 2.1 -> 2.0 ms).
static bool claim_eligible(const olap_ctx* ctx, uint64_t rows, const PassPlan& plan, bool capm) {
  if (!ctx->claim_mode || !capm || plan.passes != 2 || ctx->scatter_variant != 0 || ctx->line16) return false;
  const uint64_t F = 1ull << plan.bits[0];
  if (!claim_cap(ctx, rows, F)) return false;
  if (ctx->claim_mode >= 2) return true;
  const uint64_t kb0 = (uint64_t)ctx->compute_units * (ring_ok(ctx, plan.bits[0], 1) ? 3 : 2);
  const uint64_t kb = kb0 > 256 ? kb0 : 256;
  return kb * kClaimChunk * F <= rows;  // (worst-case padding; what is actually written as fillers is a fraction of it.  rows / 2 until
                                        // round 3: SF=20 1.36 -> 1.27 ms, SF=12 and 16 with the 8 + 6 split, profiles/r3_small_split.txt)
}


static bool pass_uses_geometry2(const olap_ctx* ctx, uint32_t fbits) {
  return ctx->scatter_variant == 0 && ctx->wc_geometry == 2 && !ctx->line16 && fbits <= 9;
}

// The ring scatter (ring_scatter.h) serves the claiming pass 0 and the capacity-layout last pass of <= 512 buckets.
bool ring_ok(const olap_ctx* ctx, uint32_t fbits, int role = 3) {  // role: 1 claiming pass, 2 capacity pass
  return (ctx->ring_scatter & role) != 0 && ctx->scatter_variant == 0 && !ctx->line16 && (1u << fbits) <= (uint32_t)kRingFmax;
}
static uint32_t ring_tile(const olap_ctx* ctx) { return (uint32_t)(kRingProd * kWave * (ctx->ring_tpt == 4 ? 4 : 8)); }

// Tile size (rows per unit of the column view) pass 0 will use for `plan`; `claim`: pass 0 is a claiming scatter.
static uint32_t pass0_tile(const olap_ctx* ctx, const PassPlan& plan, bool claim = false) {
  if (claim && ring_ok(ctx, plan.bits[0], 1)) return ring_tile(ctx);
  return pass_uses_geometry2(ctx, plan.bits[0]) ? kW2Tile : kPTile;
}

// ---- kernel launches.  Geometry 2 = {512 threads, 4096-tuple tiles, two workgroups per CU}; geometry 1 =
// {1024 threads, 8192-tuple tiles}.  The scatters need more than the default 64 KiB of dynamic LDS: the limit
// is raised once per kernel and context.
static int allow_lds(olap_ctx* ctx, const void* kernel, uint32_t bytes) {
  if (ctx->lds_raised.insert(kernel).second)
    OLAP_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return OLAP_OK;
}

template <int SRC, bool CAPM, int LINE, int BLOCK, int TILE>
static int launch_wc(olap_ctx* ctx, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, const uint32_t* hist,
                     const uint64_t* seg_out, uint2* out, const CapOut& co) {
  constexpr uint32_t kLds = WcLds<LINE, BLOCK, TILE>::kBytes;
  OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(k_part_scatter_wc<SRC, CAPM, LINE, BLOCK, TILE>), kLds));
  char label[96];
  std::snprintf(label, sizeof label, "olap::k_part_scatter_wc<%d, %s, %d, %d, %d>", SRC, CAPM ? "true" : "false", LINE, BLOCK, TILE);
  ctx->phase_kernel[CAPM ? PH_PART_SCATTER_CAP : PH_PART_SCATTER] = label;
  hipLaunchKernelGGL((k_part_scatter_wc<SRC, CAPM, LINE, BLOCK, TILE>), dim3(blocks), dim3(BLOCK), kLds, ctx->stream, src, shift, fbits,
                     hist, seg_out, out, co);
  OLAP_HIP(ctx, hipGetLastError());
  return OLAP_OK;
}

// write-combining scatter, exact (hist + seg_out) or capacity layout (CAPM; tuple input only)
template <bool CAPM>
static int launch_scatter(olap_ctx* ctx, bool cols, bool g2, bool line16, unsigned blocks, const PassSrc& src, uint32_t shift,
                          uint32_t fbits, const uint32_t* hist, const uint64_t* seg_out, uint2* out, const CapOut& co) {
  if constexpr (!CAPM) {
    if (cols) {
      if (g2) return launch_wc<SRC_COLS, false, 8, kW2Block, kW2Tile>(ctx, blocks, src, shift, fbits, hist, seg_out, out, co);
      if (line16) return launch_wc<SRC_COLS, false, 16, kWBlock, kPTile>(ctx, blocks, src, shift, fbits, hist, seg_out, out, co);
      return launch_wc<SRC_COLS, false, 8, kWBlock, kPTile>(ctx, blocks, src, shift, fbits, hist, seg_out, out, co);
    }
  }
  if (g2) return launch_wc<SRC_TUPLES, CAPM, 8, kW2Block, kW2Tile>(ctx, blocks, src, shift, fbits, hist, seg_out, out, co);
  if (line16) return launch_wc<SRC_TUPLES, CAPM, 16, kWBlock, kPTile>(ctx, blocks, src, shift, fbits, hist, seg_out, out, co);
  return launch_wc<SRC_TUPLES, CAPM, 8, kWBlock, kPTile>(ctx, blocks, src, shift, fbits, hist, seg_out, out, co);
}

template <int SRC, int BLOCK, int TILE>
static int launch_claim_as(olap_ctx* ctx, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, uint2* out,
                           const ClaimOut& co) {
  constexpr uint32_t kLds = WcLds<8, BLOCK, TILE>::kBytes;
  OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(k_part_scatter_claim<SRC, BLOCK, TILE>), kLds));
  char label[96];
  std::snprintf(label, sizeof label, "olap::k_part_scatter_claim<%d, %d, %d>", SRC, BLOCK, TILE);
  ctx->phase_kernel[PH_PART_SCATTER_CLAIM] = label;
  hipLaunchKernelGGL((k_part_scatter_claim<SRC, BLOCK, TILE>), dim3(blocks), dim3(BLOCK), kLds, ctx->stream, src, shift, fbits, out, co);
  OLAP_HIP(ctx, hipGetLastError());
  return OLAP_OK;
}

static int launch_claim(olap_ctx* ctx, bool cols, bool g2, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, uint2* out,
                        const ClaimOut& co) {
  if (g2) return cols ? launch_claim_as<SRC_COLS, kW2Block, kW2Tile>(ctx, blocks, src, shift, fbits, out, co)
                      : launch_claim_as<SRC_TUPLES, kW2Block, kW2Tile>(ctx, blocks, src, shift, fbits, out, co);
  return cols ? launch_claim_as<SRC_COLS, kWBlock, kPTile>(ctx, blocks, src, shift, fbits, out, co)
              : launch_claim_as<SRC_TUPLES, kWBlock, kPTile>(ctx, blocks, src, shift, fbits, out, co);
}

// ring scatter: claiming pass (CLAIM) or capacity-layout pass, 1024 threads, one workgroup per CU
template <int SRC, bool CLAIM, int TPT, int FULL = 0>
static int launch_ring_as(olap_ctx* ctx, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, uint2* out, const CapOut& co,
                          const ClaimOut& cq) {
  constexpr uint32_t kLds = RingCfg<kRingFmax, kRingLine, kRingBlock, kRingProd>::kBytes;
  auto kernel = k_part_scatter_ring<SRC, CLAIM, kRingFmax, kRingLine, kRingBlock, kRingProd, TPT, FULL>;
  OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kernel), kLds));
  char label[96];
  std::snprintf(label, sizeof label, "olap::k_part_scatter_ring<%d, %s, %d, %d, %d, %d, %d, %d>", SRC, CLAIM ? "true" : "false", kRingFmax, kRingLine,
                kRingBlock, kRingProd, TPT, FULL);
  ctx->phase_kernel[CLAIM ? PH_PART_SCATTER_CLAIM : PH_PART_SCATTER_CAP] = label;
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kRingBlock), kLds, ctx->stream, src, shift, fbits, out, co, cq);
  OLAP_HIP(ctx, hipGetLastError());
  return OLAP_OK;
}

// ring scatter into the EXACT layout of a histogram + scan (4096-tuple tiles: the histogram must have counted the same tiles)
// (FULL as for the claiming pass: 1 / 2 = column input of whole, 16-byte aligned tiles with / without a payload column)
template <int SRC, int FULL = 0>
static int launch_ring_exact_as(olap_ctx* ctx, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, const uint32_t* hist,
                                const uint64_t* seg_out, uint2* out) {
  constexpr uint32_t kLds = RingCfg<kRingFmax, kRingLine, kRingBlock, kRingProd>::kBytes;
  auto kernel = k_part_scatter_ring<SRC, false, kRingFmax, kRingLine, kRingBlock, kRingProd, 8, FULL, true>;
  OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kernel), kLds));
  ctx->phase_kernel[PH_PART_SCATTER] = SRC != SRC_COLS ? "olap::k_part_scatter_ring<1, false, 512, 16, 1024, 8, 8, 0, true>"
                                       : FULL == 1     ? "olap::k_part_scatter_ring<0, false, 512, 16, 1024, 8, 8, 1, true>"
                                       : FULL == 2     ? "olap::k_part_scatter_ring<0, false, 512, 16, 1024, 8, 8, 2, true>"
                                                       : "olap::k_part_scatter_ring<0, false, 512, 16, 1024, 8, 8, 0, true>";
  CapOut co{};
  co.hist = hist;
  co.seg_off = seg_out;
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kRingBlock), kLds, ctx->stream, src, shift, fbits, out, co, ClaimOut{});
  OLAP_HIP(ctx, hipGetLastError());
  return OLAP_OK;
}

// the claiming ring scatter from columns rebuilt around its instruction count (ring_claim.h)
template <int FULL, bool STOREH>
static int launch_claim_ring_as(olap_ctx* ctx, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, uint2* out, const ClaimOut& cq) {
  constexpr uint32_t kLds = Claim8Cfg<kRingFmax, kRingBlock, kRingProd>::kBytes;
  auto kernel = k_part_claim_ring<kRingFmax, kRingBlock, kRingProd, 8, STOREH, FULL>;
  OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kernel), kLds));
  char label[96];
  std::snprintf(label, sizeof label, "olap::k_part_claim_ring<%d, %d, %d, 8, %s, %d>", kRingFmax, kRingBlock, kRingProd, STOREH ? "true" : "false", FULL);
  ctx->phase_kernel[PH_PART_SCATTER_CLAIM] = label;
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kRingBlock), kLds, ctx->stream, src, shift, fbits, out, cq);
  OLAP_HIP(ctx, hipGetLastError());
  return OLAP_OK;
}
static int launch_claim_ring(olap_ctx* ctx, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, uint2* out, const ClaimOut& cq, int full) {
  if (cq.store_hash) {
    if (full == 1) return launch_claim_ring_as<1, true>(ctx, blocks, src, shift, fbits, out, cq);
    if (full == 2) return launch_claim_ring_as<2, true>(ctx, blocks, src, shift, fbits, out, cq);
    return launch_claim_ring_as<0, true>(ctx, blocks, src, shift, fbits, out, cq);
  }
  if (full == 1) return launch_claim_ring_as<1, false>(ctx, blocks, src, shift, fbits, out, cq);
  if (full == 2) return launch_claim_ring_as<2, false>(ctx, blocks, src, shift, fbits, out, cq);
  return launch_claim_ring_as<0, false>(ctx, blocks, src, shift, fbits, out, cq);
}

// full: 0 | 1 | 2 -- column input of whole, 16-byte aligned tiles (load_tile's FULL), with / without a payload column
static int launch_ring(olap_ctx* ctx, bool cols, bool claim, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, uint2* out,
                       const CapOut& co, const ClaimOut& cq, int full = 0) {
  const bool t4 = ctx->ring_tpt == 4;
  if (claim && cols && !t4 && ctx->claim_lean && fbits >= 4 && !src.ord_group_bits) return launch_claim_ring(ctx, blocks, src, shift, fbits, out, cq, full);
  if (claim) {
    if (cols && !t4 && full == 1) return launch_ring_as<SRC_COLS, true, 8, 1>(ctx, blocks, src, shift, fbits, out, co, cq);
    if (cols && !t4 && full == 2) return launch_ring_as<SRC_COLS, true, 8, 2>(ctx, blocks, src, shift, fbits, out, co, cq);
    if (cols) return t4 ? launch_ring_as<SRC_COLS, true, 4>(ctx, blocks, src, shift, fbits, out, co, cq)
                        : launch_ring_as<SRC_COLS, true, 8>(ctx, blocks, src, shift, fbits, out, co, cq);
    return t4 ? launch_ring_as<SRC_TUPLES, true, 4>(ctx, blocks, src, shift, fbits, out, co, cq)
              : launch_ring_as<SRC_TUPLES, true, 8>(ctx, blocks, src, shift, fbits, out, co, cq);
  }
  return t4 ? launch_ring_as<SRC_TUPLES, false, 4>(ctx, blocks, src, shift, fbits, out, co, cq)
            : launch_ring_as<SRC_TUPLES, false, 8>(ctx, blocks, src, shift, fbits, out, co, cq);
}

// capacity-layout ring scatter with 6-byte output tuples (pack6.h).  check: the segments are not this plan's own pass-0 regions
template <int TPT, bool CHECK, bool PIPE>
static int launch_ring6_as(olap_ctx* ctx, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, void* out, const CapOut& co,
                           uint32_t* ptop, unsigned long long* foreign) {
  constexpr uint32_t kLds = Ring6Cfg<kRingFmax, kRingBlock, kRingProd>::kBytes;
  auto kernel = k_part_scatter_ring6<kRingFmax, kRingBlock, kRingProd, TPT, CHECK, PIPE>;
  OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kernel), kLds));
  char label[96];
  std::snprintf(label, sizeof label, "olap::k_part_scatter_ring6<%d, %d, %d, %d, %s, %s>", kRingFmax, kRingBlock, kRingProd, TPT, CHECK ? "true" : "false",
                PIPE ? "true" : "false");
  ctx->phase_kernel[PH_PART_SCATTER_CAP] = label;
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kRingBlock), kLds, ctx->stream, src, shift, fbits, (uint16_t*)out, co, ptop, foreign);
  OLAP_HIP(ctx, hipGetLastError());
  return OLAP_OK;
}
static int launch_ring6(olap_ctx* ctx, bool check, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, void* out, const CapOut& co,
                        uint32_t* ptop, unsigned long long* foreign) {
  // 2048-tuple tiles with the flush one tile behind the insert, or 4096-tuple tiles flushed at once (a ring holds 42 tuples)
  if (ctx->ring6_tpt == 8)
    return check ? launch_ring6_as<8, true, false>(ctx, blocks, src, shift, fbits, out, co, ptop, foreign)
                 : launch_ring6_as<8, false, false>(ctx, blocks, src, shift, fbits, out, co, ptop, foreign);
  return check ? launch_ring6_as<4, true, true>(ctx, blocks, src, shift, fbits, out, co, ptop, foreign)
               : launch_ring6_as<4, false, true>(ctx, blocks, src, shift, fbits, out, co, ptop, foreign);
}

// The last pass of a side may hand the probe 6-byte tuples: capacity layout through the ring scatter, direct-address table.
static bool pack6_ok(const olap_ctx* ctx, const PassPlan& plan, uint32_t table_bits, bool direct) {
  return ctx->pack6 && direct && plan.passes >= 2 && table_bits >= 1 && table_bits <= 16 && plan.bits[plan.passes - 1] >= 4 &&
         ring_ok(ctx, plan.bits[plan.passes - 1], 2);  // (>= 16 buckets: the dummy rings' share of the LDS)
}

// tile-staged scatter without residues (scatter_variant = 1, A/B)
static int launch_staged(olap_ctx* ctx, bool cols, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, const uint32_t* hist,
                         const uint64_t* seg_out, uint2* out) {
  if (cols) {
    OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(k_part_scatter<SRC_COLS>), kScatterLds));
    hipLaunchKernelGGL((k_part_scatter<SRC_COLS>), dim3(blocks), dim3(kPBlock), kScatterLds, ctx->stream, src, shift, fbits, hist, seg_out, out);
  } else {
    OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(k_part_scatter<SRC_TUPLES>), kScatterLds));
    hipLaunchKernelGGL((k_part_scatter<SRC_TUPLES>), dim3(blocks), dim3(kPBlock), kScatterLds, ctx->stream, src, shift, fbits, hist, seg_out, out);
  }
  OLAP_HIP(ctx, hipGetLastError());
  return OLAP_OK;
}

static void launch_hist(olap_ctx* ctx, bool cols, bool g2, unsigned blocks, const PassSrc& src, uint32_t shift, uint32_t fbits, uint32_t* hist,
                        bool wide = false) {
  // wide: geometry 2's tiles counted by 1024-thread workgroups -- the exact ring scatter behind it runs one workgroup per CU
  if (g2 && wide && cols) hipLaunchKernelGGL((k_part_hist<SRC_COLS, kWBlock, kW2Tile>), dim3(blocks), dim3(kWBlock), 0, ctx->stream, src, shift, fbits, hist);
  else if (g2 && wide) hipLaunchKernelGGL((k_part_hist<SRC_TUPLES, kWBlock, kW2Tile>), dim3(blocks), dim3(kWBlock), 0, ctx->stream, src, shift, fbits, hist);
  else if (g2 && cols) hipLaunchKernelGGL((k_part_hist<SRC_COLS, kW2Block, kW2Tile>), dim3(blocks), dim3(kW2Block), 0, ctx->stream, src, shift, fbits, hist);
  else if (g2) hipLaunchKernelGGL((k_part_hist<SRC_TUPLES, kW2Block, kW2Tile>), dim3(blocks), dim3(kW2Block), 0, ctx->stream, src, shift, fbits, hist);
  else if (cols) hipLaunchKernelGGL((k_part_hist<SRC_COLS, kWBlock>), dim3(blocks), dim3(kWBlock), 0, ctx->stream, src, shift, fbits, hist);
  else hipLaunchKernelGGL((k_part_hist<SRC_TUPLES, kWBlock>), dim3(blocks), dim3(kWBlock), 0, ctx->stream, src, shift, fbits, hist);
}

// Input of pass 0: the batched columns `cs`, or (tuple_in != nullptr) an array of tuple_n
// (key, payload) tuples already in HBM -- what a rank receives from the all-to-all.
static int partition_side(olap_ctx* ctx, const ColumnSet& cs, bool has_payload, const uint2* tuple_in,
                          uint64_t tuple_n, const PassPlan& plan, uint32_t skip_bits, uint2* tmp0, uint2* tmp1,
                          uint2* final_tuples, bool capm, unsigned long long* d_overflow, SideOut* so,
                          const Presplit* pre = nullptr, bool claim = false, const BucketOrder* order = nullptr, bool pack = false,
                          unsigned long long* d_foreign = nullptr, bool sub = false) {
  const uint64_t rows = tuple_in ? tuple_n : cs.view.total_rows;
  PassSrc src{};
  uint64_t n_seg = 1;
  const uint64_t* seg_in = nullptr;
  const uint2* in_tuples = nullptr;
  const uint64_t* pre_ranges = nullptr;
  if (tuple_in && pre) {
    // pre-split input: upload the chunk table and the dense base of every segment
    const uint64_t nchunk = (uint64_t)pre->n_seg * pre->chunks;
    uint64_t* d_rng = (uint64_t*)ws_alloc(ctx, nchunk * 16);
    uint64_t* d_base = (uint64_t*)ws_alloc(ctx, ((uint64_t)pre->n_seg + 1) * 8);
    uint64_t* h_rng = (uint64_t*)pinned_alloc(ctx, nchunk * 16);
    uint64_t* h_base = (uint64_t*)pinned_alloc(ctx, ((uint64_t)pre->n_seg + 1) * 8);
    if (!d_rng || !d_base || !h_rng || !h_base) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (chunk table)");
    uint64_t run = 0;
    for (uint32_t j = 0; j < pre->n_seg; ++j) {
      h_base[j] = run;
      for (uint32_t c = 0; c < pre->chunks; ++c) {
        const uint64_t k = (uint64_t)j * pre->chunks + c;
        h_rng[2 * k] = pre->lo[k];
        h_rng[2 * k + 1] = pre->hi[k];
        run += pre->hi[k] - pre->lo[k];
      }
    }
    h_base[pre->n_seg] = run;
    OLAP_HIP(ctx, hipMemcpyAsync(d_rng, h_rng, nchunk * 16, hipMemcpyHostToDevice, ctx->stream));
    OLAP_HIP(ctx, hipMemcpyAsync(d_base, h_base, ((uint64_t)pre->n_seg + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    pre_ranges = d_rng;
    seg_in = d_base;
    n_seg = pre->n_seg;
    in_tuples = tuple_in;
  } else if (tuple_in) {
    uint64_t* seg0 = (uint64_t*)ws_alloc(ctx, 16);
    uint64_t* h_seg0 = (uint64_t*)pinned_alloc(ctx, 16);
    if (!seg0 || !h_seg0) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (segment table)");
    h_seg0[0] = 0;
    h_seg0[1] = tuple_n;
    OLAP_HIP(ctx, hipMemcpyAsync(seg0, h_seg0, 16, hipMemcpyHostToDevice, ctx->stream));
    seg_in = seg0;
    in_tuples = tuple_in;
  } else {
    src.v = cs.view;
    src.key_ptrs = cs.dev_ptr_tables[0];
    src.pay_ptrs = has_payload ? cs.dev_ptr_tables[1] : nullptr;
    if (cs.contiguous && ctx->contig_columns) {  // (read by the straight-line loader only, which runs when full_units holds)
      src.contig_key = (const uint32_t*)cs.h_ptrs[0][0];
      src.contig_pay = has_payload ? (const uint32_t*)cs.h_ptrs[1][0] : nullptr;
    }
  }
  if (order && order->group_bits) {
    if (plan.passes != 1 || plan.bits[0] != order->owner_bits + order->group_bits + order->low_bits)
      return ctx->fail(OLAP_ERR_INVALID, "a grouped bucket order needs a single pass over all of its bits");
    src.ord_owner_bits = order->owner_bits;
    src.ord_group_bits = order->group_bits;
    src.ord_low_bits = order->low_bits;
  }
  uint32_t consumed = skip_bits + (pre ? pre->bits : 0u);
  bool claimed_input = false;  // the current input was written by the claiming scatter
  const bool wc = ctx->scatter_variant == 0;  // write-combining scatter
  for (int i = 0; i < plan.passes; ++i) {
    const uint32_t fbits = plan.bits[i];
    const uint64_t F = 1ull << fbits;
    consumed += fbits;
    const uint32_t shift = 32 - consumed;
    uint2* out = (i == plan.passes - 1) ? final_tuples : ((i & 1) ? tmp1 : tmp0);
    uint64_t blocks, bseg, sized_blocks = 0;
    const bool cols = i == 0 && !tuple_in;
    const bool g2 = wc && pass_uses_geometry2(ctx, fbits);  // 512 threads, 4096-tuple tiles, 2 workgroups per CU
    const bool claim_pass = claim && i == 0 && plan.passes == 2 && wc && !pre;  // claiming scatter: no histogram
    const bool ring_claim = claim_pass && ring_ok(ctx, fbits, 1);
    const bool ring_cap = capm && i == plan.passes - 1 && !(pre_ranges && i == 0) && ring_ok(ctx, fbits, 2);
    // the exact layout through the ring scatter (4096-tuple tiles: geometry 2's histogram counts the same tiles); not for chunk
    // shares of pre-split input (chunk_split), whose histogram runs on other tiles
    const bool ring_exact = ctx->ring_exact && g2 && !claim_pass && !(capm && i == plan.passes - 1 && !(pre_ranges && i == 0)) &&
                            ring_ok(ctx, fbits, 3) && fbits >= 1 && (ctx->ring_exact >= 2 || rows >= (1ull << 26));  // (2^25 rows: 0.53 -> 0.58 ms)
    if (cols) {
      const uint64_t tiles = cs.view.total_units;
      uint64_t nb = (uint64_t)ctx->compute_units * ((ring_claim || (ring_exact && ctx->ring_exact_blocks == 1) || (wc && !g2)) ? 1 : 2);
      if (nb > tiles) nb = tiles;
      if (nb == 0) nb = 1;
      src.tiles_per_block = (tiles + nb - 1) / nb;
      if (src.tiles_per_block == 0) src.tiles_per_block = 1;
      // -1 (default): tile by tile with the straight-line loader (as a plain copy is fastest with round-robin tiles:
      // profiles/r2_ubench_copy_rate.txt), one contiguous run per workgroup with the generic loader (which pays a batch
      // lookup whenever its tiles are not consecutive)
      const bool full_loader = cs.full_units && cs.view.unit_rows == ring_tile(ctx) && ctx->ring_full_tiles && ctx->ring_tpt != 4;
      src.interleave = !ring_claim ? 0u : ctx->ring_interleave >= 0 ? (uint32_t)ctx->ring_interleave : full_loader ? 1u : 0u;
      blocks = (tiles + src.tiles_per_block - 1) / src.tiles_per_block;
      if (blocks == 0) blocks = 1;
      bseg = blocks;
    } else {
      const bool ranged = pre_ranges && i == 0;  // pre-split input: every chunk is read by chunk_split workgroups
      uint32_t chunk_split = 1;
      if (ranged) {
        const uint64_t want = (uint64_t)ctx->compute_units * 2, have = n_seg * pre->chunks;
        while (have * chunk_split < want && chunk_split < 64) chunk_split *= 2;
      }
      src.chunk_split = chunk_split;
      bseg = ranged ? pre->chunks * chunk_split : (capm && i == plan.passes - 1) ? 1 : seg_blocks(ctx, n_seg);
      // exact passes over the segments of an earlier pass: workgroups per segment in proportion to the segment SIZES
      // (k_seg_blocks), so that a segment far larger than the others (a hot key) gets its share of the chip -- 2^27 rows with
      // one key on half of them: pass 1 took 10.4 ms with the same number of workgroups for every segment.  The grid is an
      // upper bound of what the table hands out.
      src.blk_first = nullptr;
      uint32_t* d_blk_first = nullptr;
      if (!ranged && !(capm && i == plan.passes - 1) && ctx->balance_segments && seg_in && !claim_pass) {
        const uint64_t tile = g2 ? (uint64_t)kW2Tile : (uint64_t)kPTile;
        const uint64_t share = std::max<uint64_t>(1, rows / (n_seg * bseg) / tile);  // tiles per workgroup when the segments are equal
        const uint64_t upper = (rows / tile + 2 * n_seg) / share + n_seg + 1;      // sum of ceil(tiles(seg) / share), tiles(seg) <= rows(seg) / tile + 2
        if (upper < 0x7FFFFFFFull && share <= 0xFFFFFFFFull) {
          d_blk_first = (uint32_t*)ws_alloc(ctx, (n_seg + 1) * 4);
          if (!d_blk_first) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (segment workgroups)");
          hipLaunchKernelGGL(k_seg_blocks, dim3(1), dim3(1024), 0, ctx->stream, seg_in, (uint32_t)n_seg, (uint32_t)tile, (uint32_t)share,
                             d_blk_first);
          OLAP_HIP(ctx, hipGetLastError());
          src.blk_first = d_blk_first;
          sized_blocks = upper;
        }
      }
      if (ring_claim) bseg = std::max<uint64_t>(1, ((uint64_t)ctx->compute_units + n_seg - 1) / n_seg);  // one workgroup per CU
      blocks = src.blk_first ? sized_blocks : n_seg * bseg;
      src.tuples = in_tuples;
      src.seg_off = seg_in;
      src.n_seg = (uint32_t)n_seg;
      src.blocks_per_seg = (uint32_t)bseg;
      src.blk_range = ranged ? pre_ranges : nullptr;
    }
    if (blocks > 0x7FFFFFFFull) return ctx->fail(OLAP_ERR_INVALID, "partition grid too large");
    if (sub && pre_ranges && i == 0 && plan.passes == 1) {  // chunk by chunk into packed sub-partitions (presplit_sub_eligible)
      const uint64_t nchunk = n_seg * pre->chunks;
      so->cap = region_cap_side(max_chunk_rows(pre), F, true);
      if (nchunk * F * ((uint64_t)so->cap * 3 / 64) >= (1ull << 32)) return ctx->fail(OLAP_ERR_INVALID, "packed sub-partitions: too many lines");
      so->packed = true;
      so->nsub = pre->chunks;
      so->sub_fbits = fbits;
      so->top = (uint32_t*)ws_alloc(ctx, nchunk * F * 4);
      so->cnt = (uint32_t*)ws_alloc(ctx, nchunk * F * 4);
      so->dense = (uint64_t*)ws_alloc(ctx, nchunk * F * 8);
      uint64_t* seg_total = (uint64_t*)ws_alloc(ctx, nchunk * 8);
      if (!so->top || !so->cnt || !so->dense || !seg_total || !d_foreign) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (sub-partition tables)");
      src.tuples = in_tuples;
      src.seg_off = seg_in;
      src.n_seg = (uint32_t)n_seg;
      src.blocks_per_seg = pre->chunks;
      src.blk_range = pre_ranges;
      src.chunk_split = 1;
      src.blk_first = nullptr;
      src.out_by_block = 1;
      PhaseScope ps(ctx, PH_PART_SCATTER_CAP);
      const CapOut co{so->cap, so->cnt, so->dense, d_overflow, seg_total};
      OLAP_TRY(launch_ring6(ctx, /*check=*/true, (unsigned)nchunk, src, shift, fbits, out, co, so->top, d_foreign));
      hipLaunchKernelGGL(k_dense_fix, dim3((unsigned)nchunk), dim3(1024), 0, ctx->stream, (const uint64_t*)seg_total, (uint32_t)F, so->dense);
      OLAP_HIP(ctx, hipGetLastError());
      return OLAP_OK;
    }
    if (claim_pass) {
      const uint32_t ccap = claim_cap(ctx, rows, F);
      const bool weave = ring_claim && ctx->claim_weave && ccap % (1u << kWeaveShift) == 0;
      uint32_t* cursor = (uint32_t*)ws_alloc(ctx, F * 4);
      if (!cursor || !ccap) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (claim cursors)");
      OLAP_HIP(ctx, hipMemsetAsync(cursor, 0, F * 4, ctx->stream));
      {
        PhaseScope ps(ctx, PH_PART_SCATTER_CLAIM);
        // the packed last pass (pack6.h) only needs hashes: the ring scatter then leaves the hash in place of the key
        const bool store_hash = ring_claim && cols && ctx->ring_tpt != 4 && ctx->claim_lean && fbits >= 4 && pack && ctx->pack6_hash && capm &&
                                ring_ok(ctx, plan.bits[plan.passes - 1], 2);
        const ClaimOut co{ccap, cursor, d_overflow, weave ? kWeaveShift : 0u, (uint32_t)(n_seg * F), store_hash ? 1u : 0u};
        src.hashed = store_hash ? 1u : 0u;  // (read by the next pass)

        const bool loader_ok = cols && cs.view.unit_rows == ring_tile(ctx) && ctx->ring_full_tiles && ctx->ring_tpt != 4;
        const int full = loader_ok && cs.full_units ? (has_payload ? 1 : 2) : 0;
        if (ring_claim && loader_ok && !cs.full_units && ctx->ring_split_ragged) {
          // ragged batches (or some unaligned buffers): the whole aligned tiles of every batch through the straight-line
          // loader, the rest (tails, unaligned batches) through the generic one -- two launches on the same cursors
          ColumnSet whole, rest;
          OLAP_TRY(split_column_set(ctx, cs, &whole, &rest));
          const ColumnSet* parts[2] = {&whole, &rest};
          for (int k = 0; k < 2; ++k) {
            const ColumnSet& pc = *parts[k];
            const uint64_t tiles = pc.view.total_units;
            if (!tiles) continue;
            PassSrc s2 = src;
            s2.contig_key = s2.contig_pay = nullptr;
            s2.v = pc.view;
            s2.key_ptrs = pc.dev_ptr_tables[0];
            s2.pay_ptrs = has_payload ? pc.dev_ptr_tables[1] : nullptr;
            const uint64_t nbk = std::min<uint64_t>((uint64_t)ctx->compute_units, tiles);
            s2.tiles_per_block = (tiles + nbk - 1) / nbk;
            // (round-robin tiles only when locating a tile is a division: with pieces of different sizes every tile that does
            // not follow its predecessor costs a binary search with dependent loads)
            s2.interleave = k == 0 && pc.view.units_per_batch ? (ctx->ring_interleave >= 0 ? (uint32_t)ctx->ring_interleave : 1u) : 0u;
            const uint64_t blocks2 = (tiles + s2.tiles_per_block - 1) / s2.tiles_per_block;
            OLAP_TRY(launch_ring(ctx, true, true, (unsigned)blocks2, s2, shift, fbits, out, CapOut{}, co, k == 0 ? (has_payload ? 1 : 2) : 0));
          }
        } else if (ring_claim) OLAP_TRY(launch_ring(ctx, cols, true, (unsigned)blocks, src, shift, fbits, out, CapOut{}, co, full));
        else OLAP_TRY(launch_claim(ctx, cols, g2, (unsigned)blocks, src, shift, fbits, out, co));
        OLAP_HIP(ctx, hipGetLastError());
      }
      src.seg_cnt = cursor;  // the next pass reads regions of `ccap` tuples and skips the fillers
      src.seg_cap = ccap;
      src.weave_shift = weave ? kWeaveShift : 0u;
      src.weave_n = (uint32_t)(n_seg * F);
      src.chk_shift = shift;
      src.chk_mask = (uint32_t)F - 1;
      claimed_input = true;
      in_tuples = out;
      seg_in = nullptr;
      n_seg *= F;
      continue;
    }
    if (capm && i == plan.passes - 1 && !src.blk_range) {  // capacity layout: scatter only
      const bool pack6 = pack && ring_cap;
      so->cap = region_cap_side(rows, n_seg * F, pack6);
      so->packed = pack6;
      if (pack6) {
        so->top = (uint32_t*)ws_alloc(ctx, n_seg * F * 4);
        if (!so->top || !d_foreign) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (partition hash bits)");
      }
      so->cnt = (uint32_t*)ws_alloc(ctx, n_seg * F * 4);
      so->dense = (uint64_t*)ws_alloc(ctx, n_seg * F * 8);
      if (!so->cnt || !so->dense) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (capacity tables)");
      PhaseScope ps(ctx, PH_PART_SCATTER_CAP);
      uint64_t* seg_total = nullptr;
      if (claimed_input) {
        seg_total = (uint64_t*)ws_alloc(ctx, n_seg * 8);
        if (!seg_total) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (segment totals)");
      }
      CapOut co{so->cap, so->cnt, so->dense, d_overflow, seg_total};
      // (the leading hash bits of a segment are its index only when the segments are this plan's own first-pass regions)
      const bool own_regions = claimed_input && skip_bits == 0 && !pre && plan.passes == 2 && !src.ord_group_bits;
      if (pack6) OLAP_TRY(launch_ring6(ctx, !own_regions, (unsigned)blocks, src, shift, fbits, out, co, so->top, d_foreign));
      else if (ring_cap) OLAP_TRY(launch_ring(ctx, /*cols=*/false, false, (unsigned)blocks, src, shift, fbits, out, co, ClaimOut{}));
      else OLAP_TRY((launch_scatter<true>(ctx, /*cols=*/false, g2, fbits <= 9 && ctx->line16, (unsigned)blocks, src, shift, fbits, nullptr,
                                          nullptr, out, co)));
      OLAP_HIP(ctx, hipGetLastError());
      if (claimed_input) {  // the segments' dense starts were unknown to the scatter: number the partitions now
        hipLaunchKernelGGL(k_dense_fix, dim3((unsigned)n_seg), dim3(1024), 0, ctx->stream, (const uint64_t*)seg_total, (uint32_t)F,
                           so->dense);
        OLAP_HIP(ctx, hipGetLastError());
      }
      return OLAP_OK;
    }
    if (claimed_input) return ctx->fail(OLAP_ERR_INVALID, "internal: claimed input needs a capacity-layout consumer");
    uint32_t* hist = (uint32_t*)ws_alloc(ctx, blocks * F * 4);
    uint64_t* seg_out = (uint64_t*)ws_alloc(ctx, (n_seg * F + 1) * 8);
    if (!hist || !seg_out) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (partition pass %d)", i);
    {
      PhaseScope ps(ctx, PH_PART_HIST);
      ctx->phase_kernel[PH_PART_HIST] = g2 ? (ring_exact && cols && ctx->ring_exact_blocks == 1 ? "olap::k_part_hist<0, 1024, 4096>" : cols ? "olap::k_part_hist<0, 512, 4096>" : "olap::k_part_hist<1, 512, 4096>")
                                           : (cols ? "olap::k_part_hist<0, 1024, 8192>" : "olap::k_part_hist<1, 1024, 8192>");
      launch_hist(ctx, cols, g2, (unsigned)blocks, src, shift, fbits, hist, ring_exact && cols && ctx->ring_exact_blocks == 1);
      OLAP_HIP(ctx, hipGetLastError());
    }
    {
      PhaseScope ps(ctx, PH_PART_SCAN);
      ctx->phase_kernel[PH_PART_SCAN] = "olap::k_part_scan";
      hipLaunchKernelGGL(k_part_scan, dim3((unsigned)n_seg), dim3(1024), 0, ctx->stream, hist, seg_in, rows,
                         (uint32_t)n_seg, (uint32_t)bseg, fbits, seg_out, src.blk_first);
      OLAP_HIP(ctx, hipGetLastError());
    }
    {
      PhaseScope ps(ctx, PH_PART_SCATTER);
      if (ring_exact) {
        const bool whole = cols && cs.full_units && cs.view.unit_rows == ring_tile(ctx) && ctx->ring_full_tiles && ctx->ring_tpt != 4;
        if (cols && whole && has_payload) OLAP_TRY((launch_ring_exact_as<SRC_COLS, 1>(ctx, (unsigned)blocks, src, shift, fbits, hist, seg_out, out)));
        else if (cols && whole) OLAP_TRY((launch_ring_exact_as<SRC_COLS, 2>(ctx, (unsigned)blocks, src, shift, fbits, hist, seg_out, out)));
        else if (cols) OLAP_TRY(launch_ring_exact_as<SRC_COLS>(ctx, (unsigned)blocks, src, shift, fbits, hist, seg_out, out));
        else OLAP_TRY(launch_ring_exact_as<SRC_TUPLES>(ctx, (unsigned)blocks, src, shift, fbits, hist, seg_out, out));
      } else if (wc) {
        OLAP_TRY((launch_scatter<false>(ctx, cols, g2, fbits <= 9 && ctx->line16, (unsigned)blocks, src, shift, fbits, hist, seg_out,
                                        out, CapOut{})));
      } else {  // tile-staged A/B variant
        OLAP_TRY(launch_staged(ctx, cols, (unsigned)blocks, src, shift, fbits, hist, seg_out, out));
      }
      OLAP_HIP(ctx, hipGetLastError());
    }
    in_tuples = out;
    seg_in = seg_out;
    n_seg *= F;
  }
  so->off = const_cast<uint64_t*>(seg_in);
  return OLAP_OK;
}

static int check_side(olap_ctx* ctx, const void* keys, const uint64_t* rows, uint64_t nb, uint64_t* total) {
  if (nb && (!keys || !rows)) return ctx->fail(OLAP_ERR_INVALID, "null batch table");
  uint64_t t = 0;
  for (uint64_t b = 0; b < nb; ++b) t += rows[b];
  if (t > (1ull << 32)) return ctx->fail(OLAP_ERR_INVALID, "more than 2^32 rows on one side");
  *total = t;
  return OLAP_OK;
}

// Join inputs: batched columns per side, or (device only) pre-zipped tuple arrays.
struct JoinIn {
  const uint32_t* const* lfk = nullptr;
  const uint32_t* const* ly = nullptr;
  const uint64_t* lrows = nullptr;
  uint64_t lnb = 0;
  const uint32_t* const* rpk = nullptr;
  const uint32_t* const* rx = nullptr;
  const uint64_t* rrows = nullptr;
  uint64_t rnb = 0;
  int in_mem = OLAP_MEM_DEVICE;
  const uint2* l_tuples = nullptr;
  const uint2* r_tuples = nullptr;
  const Presplit* l_pre = nullptr;  // both or neither; same n_seg
  const Presplit* r_pre = nullptr;
  bool broadcast = false;           // the right table fits one LDS table: no partitioning (k_join_broadcast)
};

// direct-table probe over packed partitions (pack6.h).  probe6_geometry: 0 = two workgroups of 512 per CU (256-tuple
// chunks per wave), 1 = one workgroup of 1024 per CU (512-tuple chunks).
template <int BLOCK, int CHUNK>
static int launch_probe6_as(olap_ctx* ctx, bool rpk, bool lpk, const PartView& rv, const PartView& lv, uint32_t P, uint32_t table_bits,
                            uint32_t* d_fk, uint32_t* d_y, uint32_t* d_x, uint32_t pair_ok, unsigned long long* d_cnt, uint32_t* miss_bits,
                            const ProbeSlices& slices) {
  constexpr uint32_t kLds = kDSlots * 4 + kDSlots / 8 + (BLOCK / kWave) * CHUNK * 6;
  uint64_t grid = (uint64_t)ctx->compute_units * (BLOCK == 512 ? 8 : 4);
  if (grid > P) grid = P;
  char label[96];
  std::snprintf(label, sizeof label, "olap::k_join_probe_direct6<%d, %d, %s, %s>", BLOCK, CHUNK, rpk ? "true" : "false", lpk ? "true" : "false");
  ctx->phase_kernel[PH_JOIN_PROBE] = label;
#define OLAP_PROBE6(RPK, LPK)                                                                                                      \
  do {                                                                                                                             \
    auto kernel = k_join_probe_direct6<BLOCK, CHUNK, RPK, LPK>;                                                                    \
    OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kernel), kLds));                                                         \
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(BLOCK), kLds, ctx->stream, rv, lv, P, table_bits, d_fk, d_y, d_x, pair_ok, \
                       d_cnt, miss_bits, slices);                                                                                  \
  } while (0)
  if (rpk && lpk) OLAP_PROBE6(true, true);
  else if (rpk) OLAP_PROBE6(true, false);
  else OLAP_PROBE6(false, true);
#undef OLAP_PROBE6
  OLAP_HIP(ctx, hipGetLastError());
  return OLAP_OK;
}

// ... both sides packed, loads ahead of the partition's phases (k_join_probe_direct6_pf)
static int launch_probe6_pf(olap_ctx* ctx, const PartView& rv, const PartView& lv, uint32_t P, uint32_t table_bits, uint32_t* d_fk, uint32_t* d_y,
                            uint32_t* d_x, uint32_t pair_ok, unsigned long long* d_cnt, uint32_t* miss_bits, const ProbeSlices& slices) {
  constexpr int BLOCK = 512, CHUNK = 256;
  constexpr uint32_t kLds = kDSlots * 4 + kDSlots / 8 + (BLOCK / kWave) * CHUNK * 6;
  uint64_t grid = (uint64_t)ctx->compute_units * 8;
  if (grid > P) grid = P;
  ctx->phase_kernel[PH_JOIN_PROBE] = "olap::k_join_probe_direct6_pf<512, 256>";
  auto kernel = k_join_probe_direct6_pf<BLOCK, CHUNK>;
  OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kernel), kLds));
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(BLOCK), kLds, ctx->stream, rv, lv, P, table_bits, d_fk, d_y, d_x, pair_ok, d_cnt, miss_bits,
                     slices);
  OLAP_HIP(ctx, hipGetLastError());
  return OLAP_OK;
}

static int launch_probe6(olap_ctx* ctx, bool rpk, bool lpk, const PartView& rv, const PartView& lv, uint32_t P, uint32_t table_bits,
                         uint32_t* d_fk, uint32_t* d_y, uint32_t* d_x, uint32_t pair_ok, unsigned long long* d_cnt, uint32_t* miss_bits,
                         const ProbeSlices& slices) {
  if (rpk && lpk && ctx->probe6_prefetch && ctx->probe6_geometry != 1)
    return launch_probe6_pf(ctx, rv, lv, P, table_bits, d_fk, d_y, d_x, pair_ok, d_cnt, miss_bits, slices);
  if (ctx->probe6_geometry == 1)
    return launch_probe6_as<1024, 512>(ctx, rpk, lpk, rv, lv, P, table_bits, d_fk, d_y, d_x, pair_ok, d_cnt, miss_bits, slices);
  return launch_probe6_as<512, 256>(ctx, rpk, lpk, rv, lv, P, table_bits, d_fk, d_y, d_x, pair_ok, d_cnt, miss_bits, slices);
}

// ... over sub-partitions (pre-split input taken chunk by chunk)
static int launch_probe6_sub(olap_ctx* ctx, const PartView& rv, const PartView& lv, uint32_t P, uint32_t table_bits, uint32_t* d_fk, uint32_t* d_y,
                             uint32_t* d_x, uint32_t pair_ok, unsigned long long* d_cnt, uint32_t* miss_bits) {
  constexpr int BLOCK = 512, CHUNK = 256;
  constexpr uint32_t kLds = kDSlots * 4 + kDSlots / 8 + (BLOCK / kWave) * CHUNK * 6;
  uint64_t grid = (uint64_t)ctx->compute_units * 8;
  if (grid > P) grid = P;
  ctx->phase_kernel[PH_JOIN_PROBE] = "olap::k_join_probe_direct6_sub<512, 256>";
  auto kernel = k_join_probe_direct6_sub<BLOCK, CHUNK>;
  OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kernel), kLds));
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(BLOCK), kLds, ctx->stream, rv, lv, P, table_bits, d_fk, d_y, d_x, pair_ok, d_cnt, miss_bits);
  OLAP_HIP(ctx, hipGetLastError());
  return OLAP_OK;
}

static int join_once(olap_ctx* ctx, const JoinIn& in, uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x, int out_mem,
                     uint64_t NL, uint64_t NR, uint32_t total_bits, bool allow_capacity, olap_join_stats* st,
                     unsigned long long counters[6], uint64_t* rows_out, hipEvent_t right_ready) {
  const uint32_t* const* lfk = in.lfk;
  const uint32_t* const* ly = in.ly;
  const uint32_t* const* rpk = in.rpk;
  const uint32_t* const* rx = in.rx;
  const uint64_t* lrows = in.lrows;
  const uint64_t* rrows = in.rrows;
  const uint64_t lnb = in.l_tuples ? 0 : in.lnb, rnb = in.r_tuples ? 0 : in.rnb;
  const int in_mem = in.in_mem;
  PassPlan plan;
  const uint32_t pre_bits = in.l_pre ? in.l_pre->bits : 0u;  // already fixed by the pre-split
  const bool bc = in.broadcast;
  if (total_bits <= pre_bits) total_bits = pre_bits + 1;
  if (bc) {
    plan = PassPlan{};  // no radix pass at all
  } else {
    OLAP_TRY(make_plan(ctx, total_bits - pre_bits, &plan));
  }
  if (pre_bits + plan.total_bits + (uint32_t)ctx->hash_skip_bits > 32)
    return ctx->fail(OLAP_ERR_INVALID, "radix bits (%u) + skipped hash bits (%d) exceed 32", pre_bits + plan.total_bits,
                     ctx->hash_skip_bits);
  const uint32_t P = 1u << (pre_bits + plan.total_bits);
  // direct-address LDS table when few enough hash bits remain below the partition bits
  const uint32_t table_bits = 32u - (uint32_t)ctx->hash_skip_bits - pre_bits - plan.total_bits;
  const bool direct = ctx->direct_table != 0 && table_bits <= (uint32_t)kDirectBits;
  const uint32_t eb[2] = {4, 4};
  const void* const* lcols[2] = {(const void* const*)lfk, (const void* const*)ly};
  const void* const* rcols[2] = {(const void* const*)rpk, (const void* const*)rx};
  const uint64_t maxN = NL > NR ? NL : NR;
  const bool lcap = allow_capacity && capacity_eligible(ctx, NL, plan, pre_bits);
  const bool rcap = allow_capacity && capacity_eligible(ctx, NR, plan, pre_bits);
  // 6-byte tuples from the last pass to the probe (pack6.h); a side that is redone exactly falls back to 8-byte tuples
  const bool pk_ok = !bc && pack6_ok(ctx, plan, table_bits, direct);
  const bool lpack = pk_ok && lcap, rpack = pk_ok && rcap;
  // pre-split input with one pass left: chunk by chunk into packed sub-partitions (both sides, or neither), the chunk places cut
  // finer when there are too few of them
  const Presplit *lpre = in.l_pre, *rpre = in.r_pre;
  Presplit lref, rref;
  std::vector<uint64_t> lref_lo, lref_hi, rref_lo, rref_hi;
  bool sub = !bc && allow_capacity && presplit_sub_plan_ok(ctx, lpre, rpre, plan, table_bits, direct);
  if (sub) {
    const uint32_t kl = presplit_refine_factor(ctx, lpre, plan.bits[0]), kr = presplit_refine_factor(ctx, rpre, plan.bits[0]);
    if (kl > 1) {
      refine_presplit(*in.l_pre, kl, &lref_lo, &lref_hi, &lref);
      lpre = &lref;
    }
    if (kr > 1) {
      refine_presplit(*in.r_pre, kr, &rref_lo, &rref_hi, &rref);
      rpre = &rref;
    }
    sub = presplit_sub_sizes_ok(ctx, lpre, plan.bits[0]) && presplit_sub_sizes_ok(ctx, rpre, plan.bits[0]);
    if (!sub) {
      lpre = in.l_pre;
      rpre = in.r_pre;
    }
  }
  auto sub_units = [&](const Presplit* pre, uint64_t rows) -> uint64_t {  // final buffer of a side in 8-byte units
    const uint64_t regions = ((uint64_t)pre->n_seg * pre->chunks << plan.bits[0]) * region_cap_side(max_chunk_rows(pre), 1ull << plan.bits[0], true);
    return std::max<uint64_t>((regions * 6 + 7) / 8 + 512, rows);
  };
  // final buffers in 8-byte units: the (possibly packed) capacity layout, and never less than an exact redo of the side needs
  auto final_units = [&](uint64_t rows, bool cap, bool pack) -> uint64_t {
    if (bc) return 0;
    if (!cap) return rows;
    const uint64_t regions = (uint64_t)P * region_cap_side(rows, P, pack);
    return pack ? std::max<uint64_t>((regions * 6 + 7) / 8 + 512, rows) : regions;  // (+ 4 KiB: the probe's chunks read whole vectors)
  };
  const uint64_t lt_tuples = sub ? sub_units(lpre, NL) : final_units(NL, lcap, lpack);
  const uint64_t rt_tuples = sub ? sub_units(rpre, NR) : final_units(NR, rcap, rpack);
  if (lpack && (uint64_t)P * region_cap_side(NL, P, true) * 3 / 64 >= (1ull << 32)) return ctx->fail(OLAP_ERR_INVALID, "packed layout: too many lines");
  if (rpack && (uint64_t)P * region_cap_side(NR, P, true) * 3 / 64 >= (1ull << 32)) return ctx->fail(OLAP_ERR_INVALID, "packed layout: too many lines");
  const bool lclaim = !in.l_pre && claim_eligible(ctx, NL, plan, lcap);
  const bool rclaim = !in.r_pre && claim_eligible(ctx, NR, plan, rcap);
  uint64_t tmp_tuples = maxN;  // pass-0 output: dense, or the claiming scatter's padded regions
  if (lclaim) tmp_tuples = std::max<uint64_t>(tmp_tuples, ((uint64_t)claim_cap(ctx, NL, 1ull << plan.bits[0])) << plan.bits[0]);
  if (rclaim) tmp_tuples = std::max<uint64_t>(tmp_tuples, ((uint64_t)claim_cap(ctx, NR, 1ull << plan.bits[0])) << plan.bits[0]);
  const bool l_pay = ly != nullptr, r_pay = rx != nullptr;  // null payload: the tuple carries its global row index
  uint64_t need = column_set_bytes(lrows, lnb, l_pay ? 2 : 1, eb, in_mem) + column_set_bytes(rrows, rnb, r_pay ? 2 : 1, eb, in_mem) +
                  side_scratch_bytes(ctx, NL, lnb, plan, lpre) + side_scratch_bytes(ctx, NR, rnb, plan, rpre) +
                  align_up((lt_tuples + 2) * 8, 256) + align_up((rt_tuples + 2) * 8, 256) + 8192;
  need += align_up((NL / kProbeSlice + 2) * 8, 256) + 256;  // probe slice list
  need += split_column_set_bytes(lnb, 2) + split_column_set_bytes(rnb, 2);  // piece views of ragged column input
  if (lcap || lclaim || sub) need += side_scratch_bytes(ctx, NL, lnb, plan, in.l_pre);  // an exact redo of a side whose regions overflowed
  if (rcap || rclaim || sub) need += side_scratch_bytes(ctx, NR, rnb, plan, in.r_pre);
  if (plan.passes >= 2) need += align_up((tmp_tuples + 2) * 8, 256);
  if (plan.passes >= 3) need += align_up((maxN + 2) * 8, 256);
  if (out_mem == OLAP_MEM_HOST) need += 3 * align_up(NL * 4 + 4, 256);
  const bool drop = ctx->join_drop_unmatched != 0;
  const uint64_t miss_words = (NL + 31) / 32 + 2;
  if (drop) {  // miss bitmap + the row compaction's scratch
    const uint64_t cseg = (NL + kCompactSeg - 1) / kCompactSeg;
    need += align_up(miss_words * 4, 256) + align_up(cseg * 4, 256) + align_up((cseg + 1) * 8, 256) + align_up(NL * 4 + 4, 256);
  }
  uint64_t pin = (lnb + rnb + 2) * 8 * 8 + 8192 + split_column_set_pinned(lnb, 2) + split_column_set_pinned(rnb, 2);
  if (lpre) pin += 2 * ((uint64_t)lpre->n_seg * lpre->chunks * 16 + ((uint64_t)lpre->n_seg + 1) * 8 + 512) +
                   2 * ((uint64_t)rpre->n_seg * rpre->chunks * 16 + ((uint64_t)rpre->n_seg + 1) * 8 + 512) +
                   2 * ((uint64_t)in.l_pre->n_seg * in.l_pre->chunks * 16 + ((uint64_t)in.l_pre->n_seg + 1) * 8 + 512);
  OLAP_TRY(pinned_begin(ctx, pin));
  OLAP_TRY(ws_begin(ctx, need));
  ColumnSet lcs, rcs;
  if (!in.l_tuples) OLAP_TRY(build_column_set(ctx, lcols, l_pay ? 2 : 1, eb, lrows, lnb, in_mem, bc ? (uint32_t)kBcTile : pass0_tile(ctx, plan, lclaim), &lcs));
  if (!in.r_tuples) OLAP_TRY(build_column_set(ctx, rcols, r_pay ? 2 : 1, eb, rrows, rnb, in_mem, bc ? (uint32_t)kBcTile : pass0_tile(ctx, plan, rclaim), &rcs));
  uint2* tmp0 = plan.passes >= 2 ? (uint2*)ws_alloc(ctx, (tmp_tuples + 2) * 8) : nullptr;
  uint2* tmp1 = plan.passes >= 3 ? (uint2*)ws_alloc(ctx, (maxN + 2) * 8) : nullptr;
  uint2* Lt = (uint2*)ws_alloc(ctx, (lt_tuples + 2) * 8);
  uint2* Rt = (uint2*)ws_alloc(ctx, (rt_tuples + 2) * 8);
  // [0..5] probe counters; [4] / [6]: region overflows left / right; [8] / [9]: tuples of the left / right side that the
  // packed last pass found outside their segment's leading hash bits (added to [5] on the host)
  unsigned long long* d_cnt = (unsigned long long*)ws_alloc(ctx, 128);
  uint32_t *d_fk = out_fk, *d_y = out_y, *d_x = out_x;
  if (out_mem == OLAP_MEM_HOST) {
    d_fk = (uint32_t*)ws_alloc(ctx, NL * 4 + 4);
    d_y = (uint32_t*)ws_alloc(ctx, NL * 4 + 4);
    d_x = (uint32_t*)ws_alloc(ctx, NL * 4 + 4);
  }
  if ((plan.passes >= 2 && !tmp0) || (plan.passes >= 3 && !tmp1) || !Lt || !Rt || !d_cnt || !d_fk || !d_y || !d_x)
    return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (join buffers)");
  OLAP_HIP(ctx, hipMemsetAsync(d_cnt, 0, 128, ctx->stream));
  uint32_t* miss_bits = nullptr;  // inner-join mode: one bit per output row, set where the probe missed
  if (drop) {
    miss_bits = (uint32_t*)ws_alloc(ctx, miss_words * 4);
    if (!miss_bits) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (miss bitmap)");
    OLAP_HIP(ctx, hipMemsetAsync(miss_bits, 0, miss_words * 4, ctx->stream));
  }
  SideOut lso, rso;
  if (bc) {
    if (right_ready) OLAP_HIP(ctx, hipStreamWaitEvent(ctx->stream, right_ready, 0));
    PassSrc ls{}, rs{};
    ls.v = lcs.view;
    ls.key_ptrs = lcs.dev_ptr_tables[0];
    ls.pay_ptrs = l_pay ? lcs.dev_ptr_tables[1] : nullptr;
    rs.v = rcs.view;
    rs.key_ptrs = rcs.dev_ptr_tables[0];
    rs.pay_ptrs = r_pay ? rcs.dev_ptr_tables[1] : nullptr;
    const uint64_t tiles = lcs.view.total_units;
    uint64_t nb = std::min<uint64_t>((uint64_t)ctx->compute_units * 8, std::max<uint64_t>(tiles, 1));
    ls.tiles_per_block = (tiles + nb - 1) / nb;
    if (ls.tiles_per_block == 0) ls.tiles_per_block = 1;
    const uint64_t blocks = std::max<uint64_t>(1, (tiles + ls.tiles_per_block - 1) / ls.tiles_per_block);
    PhaseScope ps(ctx, PH_JOIN_PROBE);
    ctx->phase_kernel[PH_JOIN_PROBE] = "olap::k_join_broadcast";
    hipLaunchKernelGGL(k_join_broadcast, dim3((unsigned)blocks), dim3(kJBlock), 0, ctx->stream, rs, ls, d_fk, d_y, d_x, d_cnt, miss_bits);
    OLAP_HIP(ctx, hipGetLastError());
  } else {
  OLAP_TRY(partition_side(ctx, lcs, l_pay, in.l_tuples, NL, plan, (uint32_t)ctx->hash_skip_bits, tmp0, tmp1, Lt, lcap,
                          d_cnt + 4, &lso, sub ? lpre : in.l_pre, lclaim, nullptr, lpack, d_cnt + 8, sub));
  if (right_ready) {  // the right table may still be arriving (olap_join_set_right_ready_event)
    OLAP_HIP(ctx, hipStreamWaitEvent(ctx->stream, right_ready, 0));
  }
  OLAP_TRY(partition_side(ctx, rcs, r_pay, in.r_tuples, NR, plan, (uint32_t)ctx->hash_skip_bits, tmp0, tmp1, Rt, rcap,
                          d_cnt + 6, &rso, sub ? rpre : in.r_pre, rclaim, nullptr, rpack, d_cnt + 9, sub));
  // The histogram-free layouts (claimed / fixed-capacity regions) overflow when keys repeat far more than the others: a
  // foreign key with a few hundred rows is enough for its partition's region.  Find out BEFORE the probe, and redo only the
  // side that overflowed, exactly (histogram + scan + scatter into the same buffer) -- the other side's partitions stay.
  if (lcap || rcap || lclaim || rclaim || sub) {
    unsigned long long* h_over = (unsigned long long*)pinned_alloc(ctx, 32);
    if (!h_over) return ctx->fail(OLAP_ERR_NOMEM, "pinned scratch exhausted");
    OLAP_HIP(ctx, hipMemcpyAsync(h_over, d_cnt + 4, 24, hipMemcpyDeviceToHost, ctx->stream));
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long over_l = h_over[0], over_r = h_over[2];
    if (sub && (over_l || over_r)) over_l = over_r = std::max<unsigned long long>(over_l + over_r, 1);  // (the sub-partition probe reads two packed sides: both are redone)
    if (over_l || over_r) {
      ctx->last_join_region_overflows += over_l + over_r;
      ++ctx->last_join_attempts;
      OLAP_HIP(ctx, hipMemsetAsync(d_cnt + 4, 0, 8, ctx->stream));
      OLAP_HIP(ctx, hipMemsetAsync(d_cnt + 6, 0, 8, ctx->stream));
      if (over_l) {
        lso = SideOut{};
        OLAP_HIP(ctx, hipMemsetAsync(d_cnt + 8, 0, 8, ctx->stream));  // (what an overflowing attempt read may be stale memory)
        OLAP_TRY(partition_side(ctx, lcs, l_pay, in.l_tuples, NL, plan, (uint32_t)ctx->hash_skip_bits, tmp0, tmp1, Lt, false, d_cnt + 4, &lso,
                                in.l_pre, false));
      }
      if (over_r) {
        rso = SideOut{};
        OLAP_HIP(ctx, hipMemsetAsync(d_cnt + 9, 0, 8, ctx->stream));
        OLAP_TRY(partition_side(ctx, rcs, r_pay, in.r_tuples, NR, plan, (uint32_t)ctx->hash_skip_bits, tmp0, tmp1, Rt, false, d_cnt + 6, &rso,
                                in.r_pre, false));
      }
    }
  }
  }
  if (!bc) {
    PhaseScope ps(ctx, PH_JOIN_PROBE);
    uint64_t grid = (uint64_t)ctx->compute_units * 8;
    if (grid > P) grid = P;
    const PartView rv{Rt, rso.off, rso.cnt, rso.dense, rso.cap, rso.top, rso.nsub, rso.sub_fbits};
    const PartView lv{Lt, lso.off, lso.cnt, lso.dense, lso.cap, lso.top, lso.nsub, lso.sub_fbits};
    const uintptr_t a_fk = (uintptr_t)d_fk & 7, a_y = (uintptr_t)d_y & 7, a_x = (uintptr_t)d_x & 7;
    const uint32_t pair_ok = (a_fk | a_y | a_x) == 0 ? 1u : (a_fk == 4 && a_y == 4 && a_x == 4) ? 2u : 0u;
    ctx->phase_kernel[PH_JOIN_PROBE] = direct ? "olap::k_join_probe_direct" : "olap::k_join_probe";
    // probe partitions longer than one slice (skew: a hot foreign key) become several work items
    ProbeSlices slices{nullptr, nullptr, 0};
    if (ctx->probe_slices && !lso.nsub) {
      slices.capacity = (uint32_t)std::min<uint64_t>(NL / kProbeSlice + 1, 0xFFFFFFFFull);
      slices.list = (uint2*)ws_alloc(ctx, (uint64_t)slices.capacity * 8);
      slices.count = (uint32_t*)ws_alloc(ctx, 16);
      if (!slices.list || !slices.count) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (probe slices)");
      OLAP_HIP(ctx, hipMemsetAsync(slices.count, 0, 16, ctx->stream));
      hipLaunchKernelGGL(k_probe_slices, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, lv, P, slices);
    }
    if (direct && rso.packed && lso.packed && (rso.nsub || lso.nsub)) {
      OLAP_TRY(launch_probe6_sub(ctx, rv, lv, P, table_bits, d_fk, d_y, d_x, pair_ok, d_cnt, miss_bits));
    } else if (direct && (rso.packed || lso.packed)) {
      OLAP_TRY(launch_probe6(ctx, rso.packed, lso.packed, rv, lv, P, table_bits, d_fk, d_y, d_x, pair_ok, d_cnt, miss_bits, slices));
    } else if (direct) {
      uint64_t dgrid = (uint64_t)ctx->compute_units * 8;
      if (dgrid > P) dgrid = P;
      const uint32_t run = (uint32_t)std::max(1, ctx->probe_run);
      hipLaunchKernelGGL(k_join_probe_direct, dim3((unsigned)dgrid), dim3(kDBlock), 0, ctx->stream, rv, lv, P, run, table_bits,
                         (uint32_t)ctx->hash_skip_bits, d_fk, d_y, d_x, pair_ok, d_cnt, miss_bits, slices);
    } else {
      // the small table when the build partitions are at the plan's target of <= 2048 rows on average (a partition above 3072
      // rows sends the call back here with the large table: join_run)
      const bool small = ctx->probe_small_table && !ctx->probe_big_once && NR / P <= 2048;
      ctx->last_probe_slots = small ? 4096u : (uint32_t)kSlots;
      if (small) {
        ctx->phase_kernel[PH_JOIN_PROBE] = "olap::k_join_probe<4096u>";
        uint64_t sgrid = std::min<uint64_t>((uint64_t)ctx->compute_units * 16, P);
        hipLaunchKernelGGL(k_join_probe<4096>, dim3((unsigned)sgrid), dim3(kJBlock), 0, ctx->stream, rv, lv, P, d_fk, d_y, d_x, pair_ok, d_cnt,
                           miss_bits, slices);
      } else {
        ctx->phase_kernel[PH_JOIN_PROBE] = "olap::k_join_probe<8192u>";
        hipLaunchKernelGGL(k_join_probe<kSlots>, dim3((unsigned)grid), dim3(kJBlock), 0, ctx->stream, rv, lv, P, d_fk, d_y, d_x, pair_ok, d_cnt,
                           miss_bits, slices);
      }
    }
    OLAP_HIP(ctx, hipGetLastError());
  }
  unsigned long long* h_foreign = (unsigned long long*)pinned_alloc(ctx, 16);
  if (!h_foreign) return ctx->fail(OLAP_ERR_NOMEM, "pinned scratch exhausted");
  OLAP_HIP(ctx, hipMemcpyAsync(counters, d_cnt, 48, hipMemcpyDeviceToHost, ctx->stream));
  OLAP_HIP(ctx, hipMemcpyAsync(h_foreign, d_cnt + 8, 16, hipMemcpyDeviceToHost, ctx->stream));
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  counters[5] += h_foreign[0] + h_foreign[1];
  if (counters[5] && counters[2] == 0 && counters[4] == 0)
    return ctx->fail(OLAP_ERR_INVALID, "%llu tuple(s) / thread(s) saw keys that do not share the %d leading hash bits this join was told are fixed "
                     "(olap_join_set_hash_skip_bits): the inputs were not split by owner", counters[5], ctx->hash_skip_bits);
  const bool sound = counters[2] == 0 && counters[4] == 0;  // no table / region overflow: the rows written are final
  uint64_t kept_rows = NL;
  if (drop && sound && counters[0]) {
    // inner join: squeeze the marked rows out of the three columns, order preserved (scratch from the arena)
    const uint64_t n_seg = (NL + kCompactSeg - 1) / kCompactSeg;
    uint32_t* d_kept = (uint32_t*)ws_alloc(ctx, n_seg * 4);
    uint64_t* d_base = (uint64_t*)ws_alloc(ctx, (n_seg + 1) * 8);
    uint32_t* d_tmp = (uint32_t*)ws_alloc(ctx, NL * 4 + 4);
    if (!d_kept || !d_base || !d_tmp) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (row compaction)");
    std::vector<uint32_t> kept(n_seg);
    std::vector<uint64_t> base(n_seg + 1, 0);
    hipLaunchKernelGGL(k_compact_count, dim3((unsigned)n_seg), dim3(256), 0, ctx->stream, (const uint32_t*)miss_bits, NL, d_kept);
    OLAP_HIP(ctx, hipMemcpyAsync(kept.data(), d_kept, n_seg * 4, hipMemcpyDeviceToHost, ctx->stream));
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint64_t k = 0; k < n_seg; ++k) base[k + 1] = base[k] + kept[k];
    kept_rows = base[n_seg];
    if (kept_rows + counters[0] != NL)
      return ctx->fail(OLAP_ERR_DEVICE, "internal: miss bitmap (%llu rows kept) disagrees with the miss counter (%llu of %llu)",
                       (unsigned long long)kept_rows, counters[0], (unsigned long long)NL);
    OLAP_HIP(ctx, hipMemcpyAsync(d_base, base.data(), (n_seg + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    for (uint32_t* col : {d_fk, d_y, d_x}) {
      hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)n_seg), dim3(256), 0, ctx->stream, (const uint32_t*)col, (const uint32_t*)miss_bits,
                         (const uint64_t*)d_base, NL, d_tmp);
      OLAP_HIP(ctx, hipMemcpyAsync(col, d_tmp, kept_rows * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (base lives on this frame)
  }
  if (rows_out) *rows_out = kept_rows;
  if (out_mem == OLAP_MEM_HOST && sound && (counters[0] == 0 || drop) && kept_rows) {
    PhaseScope ps(ctx, PH_COPY_FROM);
    OLAP_TRY(stage_d2h(ctx, {StageItem{d_fk, out_fk, kept_rows * 4}, StageItem{d_y, out_y, kept_rows * 4},
                             StageItem{d_x, out_x, kept_rows * 4}}));
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (st) {
    st->passes = plan.passes;
    for (int i = 0; i < 4; ++i) st->radix_bits[i] = i < plan.passes ? plan.bits[i] : 0;
    st->nr_partitions = P;
    st->table_slots = direct ? (1u << table_bits) : bc ? (uint32_t)kSlots : ctx->last_probe_slots;
    st->left_rows = NL;
    st->right_rows = NR;
    st->max_build_rows = counters[3];
    st->duplicate_keys = counters[1];
    st->missing_keys = counters[0];
  }
  return resolve_timers(ctx);
}

static int join_run(olap_ctx* ctx, const JoinIn& jin, uint64_t NL, uint64_t NR, uint32_t* out_fk, uint32_t* out_y,
                    uint32_t* out_x, uint64_t out_capacity, int out_mem, uint64_t* out_rows, olap_join_stats* stats) {
  if (stats) std::memset(stats, 0, sizeof *stats);
  *out_rows = 0;
  // one-shot: consumed here, so that EVERY way out of this call (empty sides, capacity errors, failures) drops it and a
  // later join never waits on a stale handle
  const hipEvent_t right_ready = ctx->right_ready;
  ctx->right_ready = nullptr;
  if (NL == 0) return OLAP_OK;
  if (NL > out_capacity) return ctx->fail(OLAP_ERR_CAPACITY, "output capacity %llu < %llu left rows",
                                          (unsigned long long)out_capacity, (unsigned long long)NL);
  if (!out_fk || !out_y || !out_x) return ctx->fail(OLAP_ERR_INVALID, "null output column");
  if (NR == 0) {
    if (stats) { stats->left_rows = NL; stats->missing_keys = NL; }
    if (ctx->join_drop_unmatched) return OLAP_OK;  // inner join with an empty build side: no rows
    return ctx->fail(OLAP_ERR_NO_MATCH, "right table is empty: %llu foreign keys without a match", (unsigned long long)NL);
  }
  uint32_t bits;
  if (ctx->radix_bits_total >= 0) {
    bits = ctx->radix_bits_total < 1 ? 1u : (uint32_t)ctx->radix_bits_total;
  } else {
    bits = (uint32_t)ceil_log2_u64((NR + kTargetBuildRows - 1) / kTargetBuildRows);
    if (bits < 1) bits = 1;
    // Large build sides: stop at the direct-address table's 16 Ki-row partitions (two passes
    // of <= 512 buckets at 2^32 rows) instead of 4 Ki-row partitions for the probing table.
    const int dbits = 32 - kDirectBits - ctx->hash_skip_bits;
    // (one bit MORE than the probing table would need is still the better plan: the same two passes, and a probe that is
    // twice as fast per row -- SF=256: 15.5 -> 10.x ms per join)
    if (ctx->direct_table && dbits >= 1 && dbits <= (int)bits + 1 && (NR >> dbits) >= 2048) bits = (uint32_t)dbits;
    // Probing table: one bit more (2048 instead of 4096 build rows per 8192-slot table: shorter probe chains) whenever it
    // does not cost another pass -- SF=64: probe 1.63 -> 1.32 ms, SF=16: 0.46 -> 0.37 ms.
    if (!(ctx->direct_table && dbits >= 1 && bits == (uint32_t)dbits)) {
      const int maxb = std::min(ctx->max_bits_per_pass > 0 ? ctx->max_bits_per_pass : kMaxFBits, kMaxFBits);
      if ((int)(bits + 1 + maxb - 1) / maxb == (int)(bits + maxb - 1) / maxb && (NR >> (bits + 1)) >= 64) ++bits;
    }
    // A small build side must not leave the probe to a handful of workgroups (one partition = one workgroup at a time;
    // 2^28 x 2^16 rows took 52 ms with the 16 partitions the table size asks for): at least 1024 partitions -- still
    // one pass -- when the probe side is large.
    // ... and 512 where the exact ring scatter takes the probe side (<= 512 buckets, >= 2^26 rows: 0.9 instead of 1.35 ms per 2^28
    // tuples; probe slices keep the chip busy with 512 partitions): 2^28 x 2^16 rows 3.30 -> 2.54 ms (profiles/r3_asym_bits.txt)
    const uint32_t floor_bits = ctx->ring_exact && NL >= (1ull << 26) && ctx->probe_slices ? 9u : 10u;
    if (bits < floor_bits && NL >= (1ull << 22)) bits = floor_bits;
  }
  // ... and a build side that fits ONE LDS table is not partitioned at all (k_join_broadcast)
  JoinIn jb = jin;
  jb.broadcast = ctx->broadcast_join && ctx->radix_bits_total < 0 && !jin.l_tuples && !jin.r_tuples && !jin.l_pre &&
                 ctx->hash_skip_bits == 0 && NR <= (uint64_t)kTableCap;
  if (jin.l_pre && bits <= jin.l_pre->bits) bits = jin.l_pre->bits + 1;  // at least one pass gathers the chunks
  unsigned long long counters[6] = {0, 0, 0, 0, 0};
  uint64_t kept_rows = NL;
  bool allow_capacity = true;
  ctx->last_join_attempts = 0;
  ctx->last_join_region_overflows = 0;
  ctx->probe_big_once = false;
  ctx->last_probe_slots = 0;
  for (int attempt = 0; attempt < 5; ++attempt) {
    ++ctx->last_join_attempts;
    OLAP_TRY(join_once(ctx, jb, out_fk, out_y, out_x, out_mem, NL, NR, bits, allow_capacity, stats, counters, &kept_rows,
                       attempt == 0 ? right_ready : nullptr));
    if (attempt == 0) ctx->last_join_region_overflows += counters[4];  // (join_once handles overflows itself and records them)
    if (counters[4]) {  // a fixed-capacity region overflowed (skew): redo with exact partitioning
      allow_capacity = false;
      continue;
    }
    if (counters[2] == 0) break;
    if (ctx->last_probe_slots == 4096u && !jb.broadcast) {  // ... the small table: the same plan with the large one first
      ctx->probe_big_once = true;
      continue;
    }
    bits += 2;  // a build partition overflowed the LDS table: split finer and retry
    if (bits + (uint32_t)ctx->hash_skip_bits > 28) break;
  }
  if (counters[2])
    return ctx->fail(OLAP_ERR_OVERFLOW, "%llu partition(s) exceed the LDS hash table (%u rows); largest build partition %llu rows",
                     counters[2], kTableCap, counters[3]);
  if (counters[0] && !ctx->join_drop_unmatched)
    return ctx->fail(OLAP_ERR_NO_MATCH, "%llu foreign key(s) without a matching primary key", counters[0]);
  *out_rows = kept_rows;
  return OLAP_OK;
}

}  // namespace olap

using namespace olap;

extern "C" {

int olap_set_tuning(olap_ctx* ctx, const char* key, int value) {
  if (!ctx || !key) return OLAP_ERR_INVALID;
  if (!std::strcmp(key, "scatter_variant")) {
    if (value < 0 || value > 1) return ctx->fail(OLAP_ERR_INVALID, "scatter_variant must be 0 (write-combining) or 1 (tile-staged)");
    ctx->scatter_variant = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "direct_table")) {
    ctx->direct_table = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "wc_geometry")) {
    if (value != 1 && value != 2) return ctx->fail(OLAP_ERR_INVALID, "wc_geometry must be 1 or 2");
    ctx->wc_geometry = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "line16")) {
    ctx->line16 = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "capacity_mode")) {
    ctx->capacity_mode = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "take_variant")) {
    ctx->take_variant = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "ring_scatter")) {
    if (value < 0 || value > 3) return ctx->fail(OLAP_ERR_INVALID, "ring_scatter must be 0 (off), 1 (claiming pass), 2 (capacity pass) or 3 (both)");
    ctx->ring_scatter = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "filter_whole_batches")) {
    ctx->filter_whole_batches = value < 0 ? 0 : value;  // 0 off, N: from N long batches per CU on
    return OLAP_OK;
  }
  if (!std::strcmp(key, "balance_segments")) {
    ctx->balance_segments = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "probe_slices")) {
    ctx->probe_slices = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "broadcast_join")) {
    ctx->broadcast_join = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "probe_run")) {
    if (value < 1 || value > 65536) return ctx->fail(OLAP_ERR_INVALID, "probe_run must be in [1, 65536]");
    ctx->probe_run = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "ring_split_ragged")) {
    ctx->ring_split_ragged = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "ring_full_tiles")) {
    ctx->ring_full_tiles = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "claim_weave")) {
    ctx->claim_weave = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "ring_interleave")) {
    if (value < -1 || value > 4096) return ctx->fail(OLAP_ERR_INVALID, "ring_interleave must be in [-1, 4096]");
    ctx->ring_interleave = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "pack6")) {
    ctx->pack6 = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "contig_columns")) {
    ctx->contig_columns = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "claim_lean")) {
    ctx->claim_lean = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "pack6_hash")) {
    ctx->pack6_hash = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "probe6_prefetch")) {
    ctx->probe6_prefetch = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "ring_exact_blocks")) {
    if (value != 1 && value != 2) return ctx->fail(OLAP_ERR_INVALID, "ring_exact_blocks must be 1 or 2");
    ctx->ring_exact_blocks = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "ring_exact")) {
    if (value < 0 || value > 2) return ctx->fail(OLAP_ERR_INVALID, "ring_exact must be 0, 1 or 2");
    ctx->ring_exact = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "filter_stores")) {
    if (value < 0 || value > 2) return ctx->fail(OLAP_ERR_INVALID, "filter_stores must be 0, 1 or 2");
    ctx->filter_stores = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "presplit_refine")) {
    if (value < 1 || value > 8) return ctx->fail(OLAP_ERR_INVALID, "presplit_refine must be in [1, 8]");
    ctx->presplit_refine = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "presplit_sub")) {
    if (value < 0 || value > 2) return ctx->fail(OLAP_ERR_INVALID, "presplit_sub must be 0, 1 or 2");
    ctx->presplit_sub = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "join64_ring")) {
    if (value < 0 || value > 2) return ctx->fail(OLAP_ERR_INVALID, "join64_ring must be 0, 1 or 2");
    ctx->join64_ring = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "join64_tpt")) {
    if (value != 4 && value != 8) return ctx->fail(OLAP_ERR_INVALID, "join64_tpt must be 4 or 8");
    ctx->join64_tpt = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "plan_first_bits")) {
    if (value < 0 || value > kMaxFBits) return ctx->fail(OLAP_ERR_INVALID, "plan_first_bits must be in [0, %d]", kMaxFBits);
    ctx->plan_first_bits = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "probe_small_table")) {
    if (value < 0 || value > 1) return ctx->fail(OLAP_ERR_INVALID, "probe_small_table must be 0 or 1");
    ctx->probe_small_table = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "probe64_variant")) {
    if (value != 1 && value != 2 && value != 5 && value != 6) return ctx->fail(OLAP_ERR_INVALID, "probe64_variant must be W + 4 * DH with W in 1..2");
    ctx->probe64_variant = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "probe64_grid")) {
    if (value < 1 || value > 64) return ctx->fail(OLAP_ERR_INVALID, "probe64_grid must be in [1, 64]");
    ctx->probe64_grid = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "ring6_tpt")) {
    if (value != 4 && value != 8) return ctx->fail(OLAP_ERR_INVALID, "ring6_tpt must be 4 or 8");
    ctx->ring6_tpt = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "probe6_geometry")) {
    if (value < 0 || value > 1) return ctx->fail(OLAP_ERR_INVALID, "probe6_geometry must be 0 (2 x 512 threads per CU) or 1 (1 x 1024)");
    ctx->probe6_geometry = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "ring_tpt")) {
    if (value != 4 && value != 8) return ctx->fail(OLAP_ERR_INVALID, "ring_tpt must be 4 or 8");
    ctx->ring_tpt = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "claim_mode")) {
    if (value < 0 || value > 2) return ctx->fail(OLAP_ERR_INVALID, "claim_mode must be 0 (off), 1 (auto) or 2 (always)");
    ctx->claim_mode = value;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "host_stage")) {
    ctx->host_stage = value != 0;
    return OLAP_OK;
  }
  if (!std::strcmp(key, "host_copy_threads")) {
    if (value < 0 || value > 256) return ctx->fail(OLAP_ERR_INVALID, "host_copy_threads must be in [0, 256]");
    if (value != ctx->host_copy_threads) stager_destroy(ctx);  // rebuilt with the new pool on next use
    ctx->host_copy_threads = value;
    return OLAP_OK;
  }
  return ctx->fail(OLAP_ERR_INVALID, "unknown tuning key '%s'", key);
}

int olap_get_counter(olap_ctx* ctx, const char* name, uint64_t* out_value) {
  if (!ctx || !name || !out_value) return OLAP_ERR_INVALID;
  if (!std::strcmp(name, "join_attempts")) *out_value = ctx->last_join_attempts;
  else if (!std::strcmp(name, "join_region_overflows")) *out_value = ctx->last_join_region_overflows;
  else if (!std::strcmp(name, "arena_base")) *out_value = (uint64_t)(uintptr_t)ctx->ws;
  else if (!std::strcmp(name, "arena_bytes")) *out_value = ctx->ws_bytes;
  else return ctx->fail(OLAP_ERR_INVALID, "unknown counter '%s'", name);
  return OLAP_OK;
}

int olap_join_set_hash_skip_bits(olap_ctx* ctx, int bits) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (bits < 0 || bits > 16) return ctx->fail(OLAP_ERR_INVALID, "hash skip bits must be in [0, 16]");
  ctx->hash_skip_bits = bits;
  return OLAP_OK;
}

int olap_join_set_right_ready_event(olap_ctx* ctx, void* hip_event) {
  if (!ctx) return OLAP_ERR_INVALID;
  ctx->right_ready = static_cast<hipEvent_t>(hip_event);
  return OLAP_OK;
}

int olap_join_set_unmatched(olap_ctx* ctx, int mode) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (mode != OLAP_UNMATCHED_ERROR && mode != OLAP_UNMATCHED_DROP) return ctx->fail(OLAP_ERR_INVALID, "unknown unmatched-row mode %d", mode);
  ctx->join_drop_unmatched = mode == OLAP_UNMATCHED_DROP;
  return OLAP_OK;
}

int olap_join_configure(olap_ctx* ctx, int radix_bits_total, int max_bits_per_pass) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (radix_bits_total > 30 || max_bits_per_pass > kMaxFBits || max_bits_per_pass < 0)
    return ctx->fail(OLAP_ERR_INVALID, "bad join configuration");
  ctx->radix_bits_total = radix_bits_total;
  ctx->max_bits_per_pass = max_bits_per_pass;
  return OLAP_OK;
}

int olap_join_u32(olap_ctx* ctx, const uint32_t* const* left_fk_ptrs, const uint32_t* const* left_y_ptrs,
                  const uint64_t* left_rows, uint64_t n_left_batches, const uint32_t* const* right_pk_ptrs,
                  const uint32_t* const* right_x_ptrs, const uint64_t* right_rows, uint64_t n_right_batches,
                  int in_mem, uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x, uint64_t out_capacity,
                  int out_mem, uint64_t* out_rows, olap_join_stats* stats) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
  if ((in_mem != OLAP_MEM_HOST && in_mem != OLAP_MEM_DEVICE) || (out_mem != OLAP_MEM_HOST && out_mem != OLAP_MEM_DEVICE))
    return ctx->fail(OLAP_ERR_INVALID, "bad memory space");
  if (!out_rows) return ctx->fail(OLAP_ERR_INVALID, "out_rows is null");
  uint64_t NL = 0, NR = 0;
  OLAP_TRY(check_side(ctx, left_fk_ptrs, left_rows, n_left_batches, &NL));
  OLAP_TRY(check_side(ctx, right_pk_ptrs, right_rows, n_right_batches, &NR));
  JoinIn jin;
  jin.lfk = left_fk_ptrs;
  jin.ly = left_y_ptrs;
  jin.lrows = left_rows;
  jin.lnb = n_left_batches;
  jin.rpk = right_pk_ptrs;
  jin.rx = right_x_ptrs;
  jin.rrows = right_rows;
  jin.rnb = n_right_batches;
  jin.in_mem = in_mem;
  return join_run(ctx, jin, NL, NR, out_fk, out_y, out_x, out_capacity, out_mem, out_rows, stats);
}

int olap_join_tuples_u32(olap_ctx* ctx, const uint64_t* left_tuples, uint64_t n_left, const uint64_t* right_tuples,
                         uint64_t n_right, uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x, uint64_t out_capacity,
                         uint64_t* out_rows, olap_join_stats* stats) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
  if (!out_rows) return ctx->fail(OLAP_ERR_INVALID, "out_rows is null");
  if ((n_left && !left_tuples) || (n_right && !right_tuples)) return ctx->fail(OLAP_ERR_INVALID, "null tuple array");
  if ((((uintptr_t)left_tuples) | ((uintptr_t)right_tuples)) & 15)
    return ctx->fail(OLAP_ERR_INVALID, "tuple arrays must be 16-byte aligned");
  if (n_left > (1ull << 32) || n_right > (1ull << 32)) return ctx->fail(OLAP_ERR_INVALID, "more than 2^32 rows on one side");
  JoinIn jin;
  jin.l_tuples = reinterpret_cast<const uint2*>(left_tuples);
  jin.r_tuples = reinterpret_cast<const uint2*>(right_tuples);
  return join_run(ctx, jin, n_left, n_right, out_fk, out_y, out_x, out_capacity, OLAP_MEM_DEVICE, out_rows, stats);
}

static int check_presplit(olap_ctx* ctx, const uint64_t* lo, const uint64_t* hi, uint64_t n_chunks, uint64_t n_tuples,
                          uint64_t* rows) {
  uint64_t t = 0;
  for (uint64_t k = 0; k < n_chunks; ++k) {
    if (lo[k] > hi[k] || hi[k] > n_tuples) return ctx->fail(OLAP_ERR_INVALID, "chunk %llu lies outside the tuple array", (unsigned long long)k);
    t += hi[k] - lo[k];
  }
  if (t > (1ull << 32)) return ctx->fail(OLAP_ERR_INVALID, "more than 2^32 rows on one side");
  *rows = t;
  return OLAP_OK;
}

int olap_join_presplit_u32(olap_ctx* ctx, const uint64_t* left_tuples, uint64_t n_left, const uint64_t* left_chunk_lo,
                           const uint64_t* left_chunk_hi, const uint64_t* right_tuples, uint64_t n_right,
                           const uint64_t* right_chunk_lo, const uint64_t* right_chunk_hi, uint32_t n_segments,
                           uint32_t chunks_per_segment, uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x,
                           uint64_t out_capacity, uint64_t* out_rows, olap_join_stats* stats) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
  if (!out_rows) return ctx->fail(OLAP_ERR_INVALID, "out_rows is null");
  if (n_segments < 1 || (n_segments & (n_segments - 1)) || n_segments > (1u << 16))
    return ctx->fail(OLAP_ERR_INVALID, "n_segments must be a power of two in [1, 2^16]");
  if (chunks_per_segment < 1 || chunks_per_segment > 4096) return ctx->fail(OLAP_ERR_INVALID, "chunks_per_segment must be in [1, 4096]");
  if (!left_chunk_lo || !left_chunk_hi || !right_chunk_lo || !right_chunk_hi) return ctx->fail(OLAP_ERR_INVALID, "null chunk table");
  if ((n_left && !left_tuples) || (n_right && !right_tuples)) return ctx->fail(OLAP_ERR_INVALID, "null tuple array");
  if ((((uintptr_t)left_tuples) | ((uintptr_t)right_tuples)) & 15)
    return ctx->fail(OLAP_ERR_INVALID, "tuple arrays must be 16-byte aligned");
  const uint64_t nchunk = (uint64_t)n_segments * chunks_per_segment;
  uint64_t NL = 0, NR = 0;
  OLAP_TRY(check_presplit(ctx, left_chunk_lo, left_chunk_hi, nchunk, n_left, &NL));
  OLAP_TRY(check_presplit(ctx, right_chunk_lo, right_chunk_hi, nchunk, n_right, &NR));
  Presplit lp, rp;
  lp.n_seg = rp.n_seg = n_segments;
  lp.chunks = rp.chunks = chunks_per_segment;
  lp.bits = rp.bits = (uint32_t)ceil_log2_u64(n_segments);
  lp.lo = left_chunk_lo;
  lp.hi = left_chunk_hi;
  rp.lo = right_chunk_lo;
  rp.hi = right_chunk_hi;
  JoinIn jin;
  jin.l_tuples = reinterpret_cast<const uint2*>(left_tuples);
  jin.r_tuples = reinterpret_cast<const uint2*>(right_tuples);
  jin.l_pre = &lp;
  jin.r_pre = &rp;
  return join_run(ctx, jin, NL, NR, out_fk, out_y, out_x, out_capacity, OLAP_MEM_DEVICE, out_rows, stats);
}

// Shared body of olap_partition_u32 (keys/payload columns out) and
// olap_partition_tuples_u32 (zipped tuples straight into the caller's device buffer).
static int partition_run(olap_ctx* ctx, const uint32_t* const* key_ptrs, const uint32_t* const* payload_ptrs,
                         const uint64_t* rows, uint64_t n_batches, int in_mem, uint32_t nr_partitions,
                         uint32_t* out_keys, uint32_t* out_payload, uint2* out_tuples, uint64_t out_capacity,
                         int out_mem, uint64_t* out_offsets, const BucketOrder* order = nullptr) {
  if (!ctx) return OLAP_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
  if ((in_mem != OLAP_MEM_HOST && in_mem != OLAP_MEM_DEVICE) || (out_mem != OLAP_MEM_HOST && out_mem != OLAP_MEM_DEVICE))
    return ctx->fail(OLAP_ERR_INVALID, "bad memory space");
  // dpu/join/main.c:87-89: power of two, at least 2
  if (nr_partitions < 2 || (nr_partitions & (nr_partitions - 1)) || nr_partitions > (1u << 20))
    return ctx->fail(OLAP_ERR_INVALID, "nr_partitions must be a power of two in [2, 2^20]");
  if (!out_offsets) return ctx->fail(OLAP_ERR_INVALID, "out_offsets is null");
  uint64_t N = 0;
  OLAP_TRY(check_side(ctx, key_ptrs, rows, n_batches, &N));
  if (N > out_capacity) return ctx->fail(OLAP_ERR_CAPACITY, "output capacity too small");
  const bool zipped = out_tuples != nullptr;
  if (N && !zipped && (!out_keys || !out_payload)) return ctx->fail(OLAP_ERR_INVALID, "null output column");
  if (zipped && (((uintptr_t)out_tuples) & 15)) return ctx->fail(OLAP_ERR_INVALID, "out_tuples must be 16-byte aligned");
  PassPlan plan;
  OLAP_TRY(make_plan(ctx, (uint32_t)ceil_log2_u64(nr_partitions), &plan));
  const uint32_t eb[2] = {4, 4};
  const int ncols = payload_ptrs ? 2 : 1;
  const void* const* cols[2] = {(const void* const*)key_ptrs, (const void* const*)payload_ptrs};
  uint64_t need = column_set_bytes(rows, n_batches, ncols, eb, in_mem) + side_scratch_bytes(ctx, N, n_batches, plan) +
                  align_up((N + 2) * 8, 256) * (uint64_t)(plan.passes >= 3 ? 3 : plan.passes) + 8192;
  if (out_mem == OLAP_MEM_HOST) need += 2 * align_up(N * 4 + 4, 256);
  OLAP_TRY(pinned_begin(ctx, (n_batches + 2) * 8 * 4 + ((uint64_t)nr_partitions + 1) * 8 + 8192));
  OLAP_TRY(ws_begin(ctx, need));
  ColumnSet cs;
  OLAP_TRY(build_column_set(ctx, cols, ncols, eb, rows, n_batches, in_mem, pass0_tile(ctx, plan), &cs));
  uint2* tmp0 = plan.passes >= 2 ? (uint2*)ws_alloc(ctx, (N + 2) * 8) : nullptr;
  uint2* tmp1 = plan.passes >= 3 ? (uint2*)ws_alloc(ctx, (N + 2) * 8) : nullptr;
  uint2* fin = zipped ? out_tuples : (uint2*)ws_alloc(ctx, (N + 2) * 8);
  uint32_t *d_k = out_keys, *d_p = out_payload;
  if (!zipped && out_mem == OLAP_MEM_HOST) {
    d_k = (uint32_t*)ws_alloc(ctx, N * 4 + 4);
    d_p = (uint32_t*)ws_alloc(ctx, N * 4 + 4);
  }
  if ((plan.passes >= 2 && !tmp0) || (plan.passes >= 3 && !tmp1) || (N && !fin) || (N && !zipped && (!d_k || !d_p)))
    return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (partition buffers)");
  SideOut pso;
  OLAP_TRY(partition_side(ctx, cs, payload_ptrs != nullptr, nullptr, 0, plan, (uint32_t)ctx->hash_skip_bits, tmp0, tmp1,
                          fin, false, nullptr, &pso, nullptr, false, order));
  if (N && !zipped) {
    PhaseScope ps(ctx, PH_PART_SCATTER);
    uint64_t grid = (N + 255) / 256;
    const uint64_t cap = (uint64_t)ctx->compute_units * 16;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k_unzip, dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const uint2*)fin, N, d_k, d_p);
    OLAP_HIP(ctx, hipGetLastError());
  }
  OLAP_HIP(ctx, hipMemcpyAsync(out_offsets, pso.off, ((uint64_t)nr_partitions + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (!zipped && out_mem == OLAP_MEM_HOST && N) {
    PhaseScope ps(ctx, PH_COPY_FROM);
    OLAP_TRY(stage_d2h(ctx, {StageItem{d_k, out_keys, N * 4}, StageItem{d_p, out_payload, N * 4}}));
  }
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return resolve_timers(ctx);
}

int olap_partition_u32(olap_ctx* ctx, const uint32_t* const* key_ptrs, const uint32_t* const* payload_ptrs,
                       const uint64_t* rows, uint64_t n_batches, int in_mem, uint32_t nr_partitions,
                       uint32_t* out_keys, uint32_t* out_payload, uint64_t out_capacity, int out_mem,
                       uint64_t* out_offsets) {
  return partition_run(ctx, key_ptrs, payload_ptrs, rows, n_batches, in_mem, nr_partitions, out_keys, out_payload,
                       nullptr, out_capacity, out_mem, out_offsets);
}

int olap_partition_tuples_u32(olap_ctx* ctx, const uint32_t* const* key_ptrs, const uint32_t* const* payload_ptrs,
                              const uint64_t* rows, uint64_t n_batches, int in_mem, uint32_t nr_partitions,
                              uint64_t* out_tuples, uint64_t out_capacity, uint64_t* out_offsets) {
  if (ctx && !out_tuples) {
    uint64_t n = 0;
    for (uint64_t b = 0; rows && b < n_batches; ++b) n += rows[b];
    if (n) return ctx->fail(OLAP_ERR_INVALID, "out_tuples is null");
    static uint64_t dummy[2] __attribute__((aligned(16)));
    out_tuples = dummy;  // nothing is written for an empty input
  }
  return partition_run(ctx, key_ptrs, payload_ptrs, rows, n_batches, in_mem, nr_partitions, nullptr, nullptr,
                       reinterpret_cast<uint2*>(out_tuples), out_capacity, OLAP_MEM_DEVICE, out_offsets);
}

int olap_split_tuples_u32(olap_ctx* ctx, const uint32_t* const* key_ptrs, const uint32_t* const* payload_ptrs,
                          const uint64_t* rows, uint64_t n_batches, int in_mem, uint32_t owner_bits, uint32_t group_bits,
                          uint32_t low_bits, uint64_t* out_tuples, uint64_t out_capacity, uint64_t* out_offsets) {
  if (!ctx) return OLAP_ERR_INVALID;
  const uint32_t bits = owner_bits + group_bits + low_bits;
  if (bits < 1 || bits > (uint32_t)kMaxFBits || owner_bits > 10 || group_bits > 10 || low_bits > 10)
    return ctx->fail(OLAP_ERR_INVALID, "owner + group + low bits must be in [1, %d] (one radix pass)", kMaxFBits);
  if (ctx->max_bits_per_pass > 0 && bits > (uint32_t)ctx->max_bits_per_pass)
    return ctx->fail(OLAP_ERR_INVALID, "the split needs one pass of %u bits but max_bits_per_pass is %d", bits, ctx->max_bits_per_pass);
  if (!out_tuples) {
    uint64_t n = 0;
    for (uint64_t b = 0; rows && b < n_batches; ++b) n += rows[b];
    if (n) return ctx->fail(OLAP_ERR_INVALID, "out_tuples is null");
    static uint64_t dummy[2] __attribute__((aligned(16)));
    out_tuples = dummy;  // nothing is written for an empty input
  }
  BucketOrder order;
  order.owner_bits = owner_bits;
  order.group_bits = group_bits;
  order.low_bits = low_bits;
  return partition_run(ctx, key_ptrs, payload_ptrs, rows, n_batches, in_mem, 1u << bits, nullptr, nullptr,
                       reinterpret_cast<uint2*>(out_tuples), out_capacity, OLAP_MEM_DEVICE, out_offsets, &order);
}

}  // extern "C"

#include "join64.h"

#ifdef OLAP_EXP_PHASE_CLOCKS
// experiment builds only (not part of include/olap_gpu.h): reads and clears the ring scatter's phase clocks
extern "C" int olap_debug_ring_clocks(unsigned long long* out32) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(olap::g_ring_clk), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  unsigned long long zero[32] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(olap::g_ring_clk), zero, sizeof zero) != hipSuccess) return -1;
  return 0;
}
#endif
