// join64.h -- host side of olap_join_u64 (included at the end of join.hip): the radix-partitioned PK-FK join on
// 64-bit keys with 32-bit payloads, exact layout (histogram + scan + tile-staged scatter per pass), probing LDS
// table with 64-bit keys.  Device code: radix64.h.  Reference: HT_64BIT_KEYS, dpu/shared/hashtable/hashtable.h:13-18,
// hashtable.c:23-26; composition as host/join/join_dpu.cc:168-400.
#pragma once

namespace olap {

static int make_plan64(olap_ctx* ctx, uint32_t total_bits, PassPlan* plan, int min_passes = 1) {
  if (total_bits < 1) total_bits = 1;
  const int passes = std::max(min_passes, ((int)total_bits + k64MaxFBits - 1) / k64MaxFBits);
  if (passes > kMaxPasses) return ctx->fail(OLAP_ERR_INVALID, "%u radix bits need %d passes (max %d)", total_bits, passes, kMaxPasses);
  plan->passes = passes;
  plan->total_bits = total_bits;
  const uint32_t base = total_bits / passes, extra = total_bits % passes;
  for (int i = 0; i < passes; ++i) plan->bits[i] = base + ((uint32_t)i < extra ? 1u : 0u);
  return OLAP_OK;
}

static uint64_t side_scratch_bytes64(const olap_ctx* ctx, uint64_t rows, uint64_t n_batches, const PassPlan& plan) {
  uint64_t bytes = 0, n_seg = 1;
  const uint64_t tiles = rows / k64Tile + n_batches + 1;
  for (int i = 0; i < plan.passes; ++i) {
    const uint64_t F = 1ull << plan.bits[i];
    uint64_t blocks = i == 0 ? std::min<uint64_t>((uint64_t)ctx->compute_units * 2, tiles) : n_seg * seg_blocks(ctx, n_seg);
    if (blocks == 0) blocks = 1;
    bytes += align_up(blocks * F * 4, 256) + align_up((n_seg * F + 1) * 8, 256);
    n_seg *= F;
  }
  return bytes + 4096;
}

// all passes of one side; *off_out = [P + 1] partition starts in `fin`
static int partition_side64(olap_ctx* ctx, const ColumnSet& cs, bool has_payload, const PassPlan& plan, uint4* tmp0, uint4* tmp1,
                            uint4* fin, const uint64_t** off_out) {
  const uint64_t rows = cs.view.total_rows;
  PassSrc src{};
  src.v = cs.view;
  src.key_ptrs = cs.dev_ptr_tables[0];
  src.pay_ptrs = has_payload ? cs.dev_ptr_tables[1] : nullptr;
  uint64_t n_seg = 1;
  const uint64_t* seg_in = nullptr;
  const uint4* in_tuples = nullptr;
  uint32_t consumed = 0;
  for (int i = 0; i < plan.passes; ++i) {
    const uint32_t fbits = plan.bits[i];
    const uint64_t F = 1ull << fbits;
    consumed += fbits;
    const uint32_t shift = 32 - consumed;
    uint4* out = (i == plan.passes - 1) ? fin : ((i & 1) ? tmp1 : tmp0);
    const bool cols = i == 0;
    uint64_t blocks, bseg;
    if (cols) {
      const uint64_t tiles = cs.view.total_units;
      uint64_t nb = std::min<uint64_t>((uint64_t)ctx->compute_units * 2, tiles);
      if (nb == 0) nb = 1;
      src.tiles_per_block = std::max<uint64_t>(1, (tiles + nb - 1) / nb);
      blocks = std::max<uint64_t>(1, (tiles + src.tiles_per_block - 1) / src.tiles_per_block);
      bseg = blocks;
    } else {
      bseg = seg_blocks(ctx, n_seg);
      blocks = n_seg * bseg;
      src.tuples = reinterpret_cast<const uint2*>(in_tuples);
      src.seg_off = seg_in;
      src.n_seg = (uint32_t)n_seg;
      src.blocks_per_seg = (uint32_t)bseg;
    }
    if (blocks > 0x7FFFFFFFull) return ctx->fail(OLAP_ERR_INVALID, "partition grid too large");
    uint32_t* hist = (uint32_t*)ws_alloc(ctx, blocks * F * 4);
    uint64_t* seg_out = (uint64_t*)ws_alloc(ctx, (n_seg * F + 1) * 8);
    if (!hist || !seg_out) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (64-bit partition pass %d)", i);
    {
      PhaseScope ps(ctx, PH_PART_HIST);
      if (cols) hipLaunchKernelGGL(k64_hist<SRC_COLS>, dim3((unsigned)blocks), dim3(k64Block), 0, ctx->stream, src, shift, fbits, hist);
      else hipLaunchKernelGGL(k64_hist<SRC_TUPLES>, dim3((unsigned)blocks), dim3(k64Block), 0, ctx->stream, src, shift, fbits, hist);
      OLAP_HIP(ctx, hipGetLastError());
    }
    {
      PhaseScope ps(ctx, PH_PART_SCAN);
      hipLaunchKernelGGL(k_part_scan, dim3((unsigned)n_seg), dim3(1024), 0, ctx->stream, hist, seg_in, rows, (uint32_t)n_seg, (uint32_t)bseg,
                         fbits, seg_out);
      OLAP_HIP(ctx, hipGetLastError());
    }
    {
      PhaseScope ps(ctx, PH_PART_SCATTER);
      if (cols) {
        OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(k64_scatter<SRC_COLS>), k64ScatterLds));
        ctx->phase_kernel[PH_PART_SCATTER] = "olap::k64_scatter<0>";
        hipLaunchKernelGGL(k64_scatter<SRC_COLS>, dim3((unsigned)blocks), dim3(k64Block), k64ScatterLds, ctx->stream, src, shift, fbits, hist,
                           seg_out, out);
      } else {
        OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(k64_scatter<SRC_TUPLES>), k64ScatterLds));
        ctx->phase_kernel[PH_PART_SCATTER] = "olap::k64_scatter<1>";
        hipLaunchKernelGGL(k64_scatter<SRC_TUPLES>, dim3((unsigned)blocks), dim3(k64Block), k64ScatterLds, ctx->stream, src, shift, fbits,
                           hist, seg_out, out);
      }
      OLAP_HIP(ctx, hipGetLastError());
    }
    in_tuples = out;
    seg_in = seg_out;
    n_seg *= F;
  }
  *off_out = seg_in;
  return OLAP_OK;
}

// ---- the histogram-free plan (ring64.h): two ring-scatter passes on 12-byte tuples, capacity-layout probe
using K64RingCfg = Ring64Cfg<kRingFmax, kRingBlock, kRingProd>;
// Tuples per producer thread and tile: 4 (2048-tuple tiles) at 512 buckets -- a 256-byte ring holds 21 tuples, such a tile brings
// 4 on average -- and 8 (4096-tuple tiles) up to 256 buckets (rings of >= 42 tuples, 16 per tile).  join64_tpt = 4 / 8 overrides.
static int ring64_tpt(const olap_ctx* ctx, uint32_t fbits) {
  if (fbits > 8) return 4;
  return ctx->join64_tpt == 4 ? 4 : 8;
}

// Region size of pass 0 (tuples, a multiple of the allotment): the average bucket, 8 sigma (uniform hash) and the allotments the
// workgroups may leave unused -- per workgroup and bucket the one in use, the reserve and a claim in flight.  0: does not fit.
static uint32_t claim_cap64(const olap_ctx* ctx, uint64_t rows, uint64_t F) {
  const uint64_t kb = (uint64_t)ctx->compute_units * 3;
  const uint64_t avg = (rows + F - 1) / F;
  uint64_t sd = 1;
  while (sd * sd < avg) ++sd;
  const uint64_t cap = align_up(avg + avg / 64 + 8 * sd + (kb + 2) * k64Allot, k64Allot);
  return (cap >= (1ull << 32) || cap / 32 * 3 * F >= (1ull << 32)) ? 0u : (uint32_t)cap;  // (line indices are 32-bit)
}
static uint32_t region_cap64(uint64_t rows, uint64_t P) { return (uint32_t)align_up(region_cap(rows, P), k64CapAlign); }

// join64_ring tuning: 0 never, 1 (default) two-pass plans whose worst-case padding stays below half of the rows, 2 every two-pass plan
static bool ring64_eligible(const olap_ctx* ctx, uint64_t NL, uint64_t NR, const PassPlan& plan) {
  if (!ctx->join64_ring || plan.passes != 2) return false;
  for (int i = 0; i < 2; ++i)
    if (plan.bits[i] < 4 || (1u << plan.bits[i]) > (uint32_t)kRingFmax) return false;
  const uint64_t F0 = 1ull << plan.bits[0], P = 1ull << plan.total_bits;
  if (!claim_cap64(ctx, NL, F0) || !claim_cap64(ctx, NR, F0)) return false;
  if (region_cap64(NL, P) / 32ull * 3 * P >= (1ull << 32) || region_cap64(NR, P) / 32ull * 3 * P >= (1ull << 32)) return false;
  if (ctx->join64_ring >= 2) return true;
  const uint64_t pad = (uint64_t)ctx->compute_units * 3 * k64Allot * F0;  // worst case; what is written as fillers is about a third of it
  return pad <= NL / 2 && pad <= NR / 2;
}

struct Side12 {
  uint32_t cap = 0;
  uint32_t* cnt = nullptr;
  uint64_t* dense = nullptr;
  const Tuple12* t = nullptr;
};

static uint64_t side_scratch_bytes64_ring(uint64_t F0, uint64_t P) {
  return align_up(F0 * 4, 256) + align_up(F0 * 8, 256) + align_up(P * 4, 256) + align_up(P * 8, 256) + 1024;
}

// both passes of one side: columns -> claimed regions in `tmp` -> capacity layout in `fin`
static int partition_side64_ring(olap_ctx* ctx, const ColumnSet& cs, bool has_payload, const PassPlan& plan, uint32_t* tmp, uint32_t* fin,
                                 unsigned long long* d_overflow, Side12* so) {
  const uint64_t rows = cs.view.total_rows;
  const uint32_t b0 = plan.bits[0], b1 = plan.bits[1];
  const uint64_t F0 = 1ull << b0, F1 = 1ull << b1, P = F0 * F1;
  const uint32_t cap0 = claim_cap64(ctx, rows, F0), cap1 = region_cap64(rows, P);
  uint32_t* cursor = (uint32_t*)ws_alloc(ctx, F0 * 4);
  uint64_t* seg_total = (uint64_t*)ws_alloc(ctx, F0 * 8);
  so->cnt = (uint32_t*)ws_alloc(ctx, P * 4);
  so->dense = (uint64_t*)ws_alloc(ctx, P * 8);
  so->cap = cap1;
  so->t = (const Tuple12*)fin;
  if (!cursor || !seg_total || !so->cnt || !so->dense || !cap0) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (64-bit ring passes)");
  OLAP_HIP(ctx, hipMemsetAsync(cursor, 0, F0 * 4, ctx->stream));
  // (the claiming pass gains from the larger tile -- 2^27 tuples 0.89 -> 0.72 ms -- the capacity pass loses: 0.92 -> 1.12 ms)
  const bool claim8 = ring64_tpt(ctx, b0) == 8, cap8 = false;
  auto claim_kernel = claim8 ? k64_ring<true, kRingFmax, kRingBlock, kRingProd, 8> : k64_ring<true, kRingFmax, kRingBlock, kRingProd, 4>;
  auto cap_kernel = cap8 ? k64_ring<false, kRingFmax, kRingBlock, kRingProd, 8> : k64_ring<false, kRingFmax, kRingBlock, kRingProd, 4>;
  OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(claim_kernel), K64RingCfg::kBytes));
  OLAP_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cap_kernel), K64RingCfg::kBytes));
  {  // pass 0: one workgroup per CU, a contiguous run of tiles each
    PassSrc src{};
    src.v = cs.view;
    src.key_ptrs = cs.dev_ptr_tables[0];
    src.pay_ptrs = has_payload ? cs.dev_ptr_tables[1] : nullptr;
    const uint64_t tiles = cs.view.total_units;
    const uint64_t nb = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ctx->compute_units, tiles));
    src.tiles_per_block = std::max<uint64_t>(1, (tiles + nb - 1) / nb);
    const uint64_t blocks = std::max<uint64_t>(1, (tiles + src.tiles_per_block - 1) / src.tiles_per_block);
    const ClaimOut cq{cap0, cursor, d_overflow, 0u, (uint32_t)F0, 0u};
    PhaseScope ps(ctx, PH_PART_SCATTER_CLAIM);
    ctx->phase_kernel[PH_PART_SCATTER_CLAIM] = claim8 ? "olap::k64_ring<true, 512, 1024, 8, 8>" : "olap::k64_ring<true, 512, 1024, 8, 4>";
    hipLaunchKernelGGL(claim_kernel, dim3((unsigned)blocks), dim3(kRingBlock), K64RingCfg::kBytes, ctx->stream, src, 32u - b0, b0, tmp, CapOut{}, cq);
    OLAP_HIP(ctx, hipGetLastError());
  }
  {  // pass 1: one workgroup per pass-0 bucket
    PassSrc src{};
    src.tuples = reinterpret_cast<const uint2*>(tmp);
    src.seg_cnt = cursor;
    src.seg_cap = cap0;
    src.chk_shift = 32u - b0;
    src.chk_mask = (uint32_t)F0 - 1u;
    src.n_seg = (uint32_t)F0;
    src.blocks_per_seg = 1;
    const CapOut co{cap1, so->cnt, so->dense, d_overflow, seg_total};
    PhaseScope ps(ctx, PH_PART_SCATTER_CAP);
    ctx->phase_kernel[PH_PART_SCATTER_CAP] = cap8 ? "olap::k64_ring<false, 512, 1024, 8, 8>" : "olap::k64_ring<false, 512, 1024, 8, 4>";
    hipLaunchKernelGGL(cap_kernel, dim3((unsigned)F0), dim3(kRingBlock), K64RingCfg::kBytes, ctx->stream, src, 32u - b0 - b1, b1, fin, co, ClaimOut{});
    OLAP_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_dense_fix, dim3((unsigned)F0), dim3(1024), 0, ctx->stream, (const uint64_t*)seg_total, (uint32_t)F1, so->dense);
    OLAP_HIP(ctx, hipGetLastError());
  }
  return OLAP_OK;
}

static int join64_once(olap_ctx* ctx, const uint64_t* const* lfk, const uint32_t* const* ly, const uint64_t* lrows, uint64_t lnb,
                       const uint64_t* const* rpk, const uint32_t* const* rx, const uint64_t* rrows, uint64_t rnb, int in_mem, uint64_t NL,
                       uint64_t NR, uint32_t total_bits, uint64_t* out_fk, uint32_t* out_y, uint32_t* out_x, int out_mem,
                       olap_join_stats* st, unsigned long long counters[4], uint64_t* rows_out, bool ring, bool* ring_overflow) {
  PassPlan plan;
  // (join64_ring = 2: 8 or 9 bits are split over two passes as well, so that the ring plan can be exercised at small sizes)
  OLAP_TRY(make_plan64(ctx, total_bits, &plan, ring && ctx->join64_ring >= 2 && total_bits >= 8 && total_bits <= (uint32_t)k64MaxFBits ? 2 : 1));
  const uint32_t P = 1u << plan.total_bits;
  ring = ring && ring64_eligible(ctx, NL, NR, plan);
  if (ring && plan.total_bits == 17) {  // 8 + 9, not 9 + 8: pass 0 keeps 256 buckets (4096-tuple tiles) and still feeds the last pass 256 segments
    plan.bits[0] = 8;
    plan.bits[1] = 9;
  }
  *ring_overflow = false;
  const uint32_t eb[2] = {8, 4};
  const bool l_pay = ly != nullptr, r_pay = rx != nullptr;
  const void* const* lcols[2] = {(const void* const*)lfk, (const void* const*)ly};
  const void* const* rcols[2] = {(const void* const*)rpk, (const void* const*)rx};
  const uint64_t maxN = std::max(NL, NR);
  const bool drop = ctx->join_drop_unmatched != 0;
  const uint64_t miss_words = (NL + 31) / 32 + 2, cseg = (NL + kCompactSeg - 1) / kCompactSeg;
  uint64_t need = column_set_bytes(lrows, lnb, l_pay ? 2 : 1, eb, in_mem) + column_set_bytes(rrows, rnb, r_pay ? 2 : 1, eb, in_mem) + 8192;
  need += align_up((NL / kProbeSlice + 2) * 8, 256) + 256;  // probe slice list
  const uint64_t F0 = 1ull << plan.bits[0];
  uint64_t tmp_bytes = 0, lfin_bytes = 0, rfin_bytes = 0;  // ring plan: claimed regions (one side at a time), capacity regions per side
  if (ring) {
    tmp_bytes = align_up((uint64_t)std::max(claim_cap64(ctx, NL, F0), claim_cap64(ctx, NR, F0)) * F0 * k64TupleBytes, 256);
    lfin_bytes = align_up((uint64_t)region_cap64(NL, P) * P * k64TupleBytes, 256) + 65536;  // (+ slack: whole 16-byte vectors / tiles are read)
    rfin_bytes = align_up((uint64_t)region_cap64(NR, P) * P * k64TupleBytes, 256) + 65536;
    tmp_bytes += 65536;
    need += tmp_bytes + lfin_bytes + rfin_bytes + 2 * side_scratch_bytes64_ring(F0, P);
  } else {
    need += side_scratch_bytes64(ctx, NL, lnb, plan) + side_scratch_bytes64(ctx, NR, rnb, plan) + align_up((NL + 1) * 16, 256) +
            align_up((NR + 1) * 16, 256);
    if (plan.passes >= 2) need += align_up((maxN + 1) * 16, 256);
    if (plan.passes >= 3) need += align_up((maxN + 1) * 16, 256);
  }
  if (out_mem == OLAP_MEM_HOST) need += align_up(NL * 8 + 8, 256) + 2 * align_up(NL * 4 + 4, 256);
  if (drop) need += align_up(miss_words * 4, 256) + align_up(cseg * 4, 256) + align_up((cseg + 1) * 8, 256) + align_up(NL * 8 + 8, 256);
  OLAP_TRY(pinned_begin(ctx, (lnb + rnb + 2) * 8 * 8 + (cseg + 1) * 12 + 8192));
  OLAP_TRY(ws_begin(ctx, need));
  ColumnSet lcs, rcs;
  const uint32_t unit_rows = ring ? (uint32_t)(kRingProd * kWave * ring64_tpt(ctx, plan.bits[0])) : (uint32_t)k64Tile;
  OLAP_TRY(build_column_set(ctx, lcols, l_pay ? 2 : 1, eb, lrows, lnb, in_mem, unit_rows, &lcs));
  OLAP_TRY(build_column_set(ctx, rcols, r_pay ? 2 : 1, eb, rrows, rnb, in_mem, unit_rows, &rcs));
  uint4* tmp0 = ring ? (uint4*)ws_alloc(ctx, tmp_bytes) : plan.passes >= 2 ? (uint4*)ws_alloc(ctx, (maxN + 1) * 16) : nullptr;
  uint4* tmp1 = !ring && plan.passes >= 3 ? (uint4*)ws_alloc(ctx, (maxN + 1) * 16) : nullptr;
  uint4* Lt = (uint4*)ws_alloc(ctx, ring ? lfin_bytes : (NL + 1) * 16);
  uint4* Rt = (uint4*)ws_alloc(ctx, ring ? rfin_bytes : (NR + 1) * 16);
  unsigned long long* d_cnt = (unsigned long long*)ws_alloc(ctx, 64);  // [0..3] the probe's counters, [4] region overflows of the ring passes
  uint64_t* d_fk = out_fk;
  uint32_t *d_y = out_y, *d_x = out_x;
  if (out_mem == OLAP_MEM_HOST) {
    d_fk = (uint64_t*)ws_alloc(ctx, NL * 8 + 8);
    d_y = (uint32_t*)ws_alloc(ctx, NL * 4 + 4);
    d_x = (uint32_t*)ws_alloc(ctx, NL * 4 + 4);
  }
  if ((plan.passes >= 2 && !tmp0) || (!ring && plan.passes >= 3 && !tmp1) || !Lt || !Rt || !d_cnt || !d_fk || !d_y || !d_x)
    return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (64-bit join buffers)");
  OLAP_HIP(ctx, hipMemsetAsync(d_cnt, 0, 64, ctx->stream));
  uint32_t* miss_bits = nullptr;
  if (drop) {
    miss_bits = (uint32_t*)ws_alloc(ctx, miss_words * 4);
    if (!miss_bits) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (miss bitmap)");
    OLAP_HIP(ctx, hipMemsetAsync(miss_bits, 0, miss_words * 4, ctx->stream));
  }
  const uint64_t *loff = nullptr, *roff = nullptr;
  Side12 ls, rs;
  if (ring) {
    OLAP_TRY(partition_side64_ring(ctx, lcs, l_pay, plan, (uint32_t*)tmp0, (uint32_t*)Lt, d_cnt + 4, &ls));
    OLAP_TRY(partition_side64_ring(ctx, rcs, r_pay, plan, (uint32_t*)tmp0, (uint32_t*)Rt, d_cnt + 4, &rs));
    // a region overflowed (skew): nothing below is trusted, the caller redoes the join with the exact layout
    unsigned long long h_over = 0;
    OLAP_HIP(ctx, hipMemcpyAsync(&h_over, d_cnt + 4, 8, hipMemcpyDeviceToHost, ctx->stream));
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->last_join_region_overflows = h_over;
    if (h_over) {
      *ring_overflow = true;
      return resolve_timers(ctx);
    }
  } else {
    OLAP_TRY(partition_side64(ctx, lcs, l_pay, plan, tmp0, tmp1, Lt, &loff));
    OLAP_TRY(partition_side64(ctx, rcs, r_pay, plan, tmp0, tmp1, Rt, &roff));
  }
  {
    PhaseScope ps(ctx, PH_JOIN_PROBE);
    ctx->phase_kernel[PH_JOIN_PROBE] = ring ? "olap::k64_probe12" : "olap::k64_probe";
    const uint64_t grid = std::min<uint64_t>((uint64_t)ctx->compute_units * (ring ? ctx->probe64_grid : 8), P);
    ProbeSlices slices{nullptr, nullptr, 0};  // probe partitions longer than one slice (a hot foreign key) become several work items
    if (ctx->probe_slices) {
      slices.capacity = (uint32_t)std::min<uint64_t>(NL / kProbeSlice + 1, 0xFFFFFFFFull);
      slices.list = (uint2*)ws_alloc(ctx, (uint64_t)slices.capacity * 8);
      slices.count = (uint32_t*)ws_alloc(ctx, 16);
      if (!slices.list || !slices.count) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (probe slices)");
      OLAP_HIP(ctx, hipMemsetAsync(slices.count, 0, 16, ctx->stream));
      const PartView lv{nullptr, loff, ls.cnt, ls.dense, ls.cap, nullptr};
      hipLaunchKernelGGL(k_probe_slices, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, lv, P, slices);
    }
    if (ring) {
      auto kern = k64_probe12<1, false>;
      switch (ctx->probe64_variant) {  // W + 4 * DH
        case 2: kern = k64_probe12<2, false>; break;
        case 5: kern = k64_probe12<1, true>; break;
        case 6: kern = k64_probe12<2, true>; break;
        default: break;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(k64Block), 0, ctx->stream, Part12{rs.t, rs.cnt, rs.dense, rs.cap},
                         Part12{ls.t, ls.cnt, ls.dense, ls.cap}, P, d_fk, d_y, d_x, d_cnt, miss_bits, slices);
    } else
      hipLaunchKernelGGL(k64_probe, dim3((unsigned)grid), dim3(k64Block), 0, ctx->stream, (const uint4*)Rt, roff, (const uint4*)Lt, loff, P, d_fk,
                         d_y, d_x, d_cnt, miss_bits, slices);
    OLAP_HIP(ctx, hipGetLastError());
  }
  OLAP_HIP(ctx, hipMemcpyAsync(counters, d_cnt, 32, hipMemcpyDeviceToHost, ctx->stream));
  OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const bool sound = counters[2] == 0;
  uint64_t kept_rows = NL;
  if (drop && sound && counters[0]) {  // inner join: squeeze the marked rows out, order preserved (scratch from the arena)
    uint32_t* d_kept = (uint32_t*)ws_alloc(ctx, cseg * 4);
    uint64_t* d_base = (uint64_t*)ws_alloc(ctx, (cseg + 1) * 8);
    uint64_t* d_tmp = (uint64_t*)ws_alloc(ctx, NL * 8 + 8);
    uint32_t* h_kept = (uint32_t*)pinned_alloc(ctx, cseg * 4);
    uint64_t* h_base = (uint64_t*)pinned_alloc(ctx, (cseg + 1) * 8);
    if (!d_kept || !d_base || !d_tmp || !h_kept || !h_base) return ctx->fail(OLAP_ERR_NOMEM, "workspace exhausted (row compaction)");
    hipLaunchKernelGGL(k_compact_count, dim3((unsigned)cseg), dim3(256), 0, ctx->stream, (const uint32_t*)miss_bits, NL, d_kept);
    OLAP_HIP(ctx, hipMemcpyAsync(h_kept, d_kept, cseg * 4, hipMemcpyDeviceToHost, ctx->stream));
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h_base[0] = 0;
    for (uint64_t k = 0; k < cseg; ++k) h_base[k + 1] = h_base[k] + h_kept[k];
    kept_rows = h_base[cseg];
    if (kept_rows + counters[0] != NL) return ctx->fail(OLAP_ERR_DEVICE, "internal: miss bitmap disagrees with the miss counter");
    OLAP_HIP(ctx, hipMemcpyAsync(d_base, h_base, (cseg + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k64_compact_rows, dim3((unsigned)cseg), dim3(256), 0, ctx->stream, (const uint64_t*)d_fk, (const uint32_t*)miss_bits,
                       (const uint64_t*)d_base, NL, d_tmp);
    OLAP_HIP(ctx, hipMemcpyAsync(d_fk, d_tmp, kept_rows * 8, hipMemcpyDeviceToDevice, ctx->stream));
    for (uint32_t* col : {d_y, d_x}) {
      hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)cseg), dim3(256), 0, ctx->stream, (const uint32_t*)col, (const uint32_t*)miss_bits,
                         (const uint64_t*)d_base, NL, (uint32_t*)d_tmp);
      OLAP_HIP(ctx, hipMemcpyAsync(col, d_tmp, kept_rows * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  *rows_out = kept_rows;
  if (out_mem == OLAP_MEM_HOST && sound && (counters[0] == 0 || drop) && kept_rows) {
    PhaseScope ps(ctx, PH_COPY_FROM);
    OLAP_TRY(stage_d2h(ctx, {StageItem{d_fk, out_fk, kept_rows * 8}, StageItem{d_y, out_y, kept_rows * 4}, StageItem{d_x, out_x, kept_rows * 4}}));
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (st) {
    st->passes = plan.passes;
    for (int i = 0; i < 4; ++i) st->radix_bits[i] = i < plan.passes ? plan.bits[i] : 0;
    st->nr_partitions = P;
    st->table_slots = k64Slots;
    st->left_rows = NL;
    st->right_rows = NR;
    st->max_build_rows = counters[3];
    st->duplicate_keys = counters[1];
    st->missing_keys = counters[0];
  }
  return resolve_timers(ctx);
}

}  // namespace olap

extern "C" int olap_join_u64(olap_ctx* ctx, const uint64_t* const* left_fk_ptrs, const uint32_t* const* left_y_ptrs, const uint64_t* left_rows,
                             uint64_t n_left_batches, const uint64_t* const* right_pk_ptrs, const uint32_t* const* right_x_ptrs,
                             const uint64_t* right_rows, uint64_t n_right_batches, int in_mem, uint64_t* out_fk, uint32_t* out_y,
                             uint32_t* out_x, uint64_t out_capacity, int out_mem, uint64_t* out_rows, olap_join_stats* stats) {
  using namespace olap;
  if (!ctx) return OLAP_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
  if ((in_mem != OLAP_MEM_HOST && in_mem != OLAP_MEM_DEVICE) || (out_mem != OLAP_MEM_HOST && out_mem != OLAP_MEM_DEVICE))
    return ctx->fail(OLAP_ERR_INVALID, "bad memory space");
  if (!out_rows) return ctx->fail(OLAP_ERR_INVALID, "out_rows is null");
  uint64_t NL = 0, NR = 0;
  OLAP_TRY(check_side(ctx, left_fk_ptrs, left_rows, n_left_batches, &NL));
  OLAP_TRY(check_side(ctx, right_pk_ptrs, right_rows, n_right_batches, &NR));
  if (stats) std::memset(stats, 0, sizeof *stats);
  *out_rows = 0;
  if (NL == 0) return OLAP_OK;
  if (NL > out_capacity) return ctx->fail(OLAP_ERR_CAPACITY, "output capacity %llu < %llu left rows", (unsigned long long)out_capacity, (unsigned long long)NL);
  if (!out_fk || !out_y || !out_x) return ctx->fail(OLAP_ERR_INVALID, "null output column");
  if (NR == 0) {
    if (stats) { stats->left_rows = NL; stats->missing_keys = NL; }
    if (ctx->join_drop_unmatched) return OLAP_OK;
    return ctx->fail(OLAP_ERR_NO_MATCH, "right table is empty: %llu foreign keys without a match", (unsigned long long)NL);
  }
  uint32_t bits = ctx->radix_bits_total >= 0 ? (uint32_t)std::max(1, ctx->radix_bits_total)
                                             : (uint32_t)std::max(1, ceil_log2_u64((NR + k64TargetBuildRows - 1) / k64TargetBuildRows));
  // a small build side must not leave the probe to a handful of workgroups (as in olap_join_u32): >= 512 partitions, one pass
  if (ctx->radix_bits_total < 0 && bits < (uint32_t)k64MaxFBits && NL >= (1ull << 22)) bits = k64MaxFBits;
  // The histogram-free passes cost the same for one bit more as long as pass 0 keeps 256 buckets (<= 17 bits in all), and a table
  // at load 0.25 instead of 0.5 shortens the probing chains the probe kernel spends its time in (2^27 rows: 5.38 -> 5.08 ms).
  if (ctx->radix_bits_total < 0 && ctx->join64_ring && bits >= 8 && bits < 17) {
    PassPlan finer;
    if (make_plan64(ctx, bits + 1, &finer, 2) == OLAP_OK && ring64_eligible(ctx, NL, NR, finer)) ++bits;
  }
  unsigned long long counters[4] = {0, 0, 0, 0};
  uint64_t kept_rows = NL;
  bool ring = true;  // the histogram-free plan where it applies; a region overflow (skew) falls back to the exact layout
  ctx->last_join_attempts = 0;
  for (int attempt = 0; attempt < 6; ++attempt) {
    bool ring_overflow = false;
    ++ctx->last_join_attempts;
    OLAP_TRY(join64_once(ctx, left_fk_ptrs, left_y_ptrs, left_rows, n_left_batches, right_pk_ptrs, right_x_ptrs, right_rows, n_right_batches,
                         in_mem, NL, NR, bits, out_fk, out_y, out_x, out_mem, stats, counters, &kept_rows, ring, &ring_overflow));
    if (ring_overflow) {
      ring = false;
      continue;
    }
    if (counters[2] == 0) break;
    bits += 2;  // a build partition overflowed the LDS table: split finer and retry
    if (bits > 30) break;
  }
  if (counters[2])
    return ctx->fail(OLAP_ERR_OVERFLOW, "%llu partition(s) exceed the LDS hash table (%u rows); largest build partition %llu rows", counters[2],
                     k64TableCap, counters[3]);
  if (counters[0] && !ctx->join_drop_unmatched)
    return ctx->fail(OLAP_ERR_NO_MATCH, "%llu foreign key(s) without a matching primary key", counters[0]);
  *out_rows = kept_rows;
  return OLAP_OK;
}
