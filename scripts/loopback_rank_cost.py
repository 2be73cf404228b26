#!/usr/bin/env python3
"""Per-rank device cost of the multi-GPU join, measured on ONE GPU (no exchange): for
G in {2,4,8} one logical rank's share of the SF=2048 problem goes through the owner split
(olap_partition_tuples_u32 x2) and the local join on a received-size tuple set
(olap_join_tuples_u32, hash skip bits = log2 G).  Prints the phase times; the exchange time
must be added (xGMI) to project a G-GPU step.  Projection aid only -- labelled as such."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpu_olap_amd import capi, datagen as dg
from dpu_olap_amd.dist_join import GpuLocalOps

ROWS = 2 << 20
ctx = capi.Context(0)
dev = torch.device("cuda", 0)
ops = GpuLocalOps(ctx, dev)
for G in (2, 4, 8):
    nb = 2048 // G
    n = nb * ROWS
    col = lambda: torch.empty(n, dtype=torch.int32, device=dev)
    fk, y, pk, x = col(), col(), col(), col()
    ctx.gen_fk_u32(fk.data_ptr(), n, dg.SEED, dg.STREAM_FK, 0, ROWS, ROWS)
    ctx.gen_uniform_u32(y.data_ptr(), n, dg.SEED, dg.STREAM_Y, 0)
    ctx.gen_index_u32(pk.data_ptr(), n, 0)
    ctx.gen_uniform_u32(x.data_ptr(), n, dg.SEED, dg.STREAM_X, 0)
    sl = lambda t: [t[b * ROWS:(b + 1) * ROWS] for b in range(nb)]
    res = {}
    for rep in range(2):
        ctx.set_profiling(True); ctx.reset_timers()
        t0 = time.perf_counter()
        lt, loff = ops.partition_tuples(sl(fk), sl(y), G)
        rt, roff = ops.partition_tuples(sl(pk), sl(x), G)
        ctx.synchronize(); t1 = time.perf_counter()
        split_t = ctx.timers()
        # received set of one rank ~ n tuples that all share the owner bits: use region 0 of a G-times larger split
        # emulate by taking this rank's own region 0 G times is wrong (duplicates); instead join the whole local set
        # restricted to owner 0 from all G regions of an equal-size different slice: here we simply time the local
        # join on n tuples with skip bits by re-partitioning region sizes: take owner-0 tuples only and scale.
        ctx.reset_timers()
        l0, r0 = lt[loff[0]:loff[1]].contiguous(), rt[roff[0]:roff[1]].contiguous()
        t2 = time.perf_counter()
        ofk, oy, ox, st = ops.join_tuples(l0, r0, G.bit_length() - 1)
        ctx.synchronize(); t3 = time.perf_counter()
        join_t = ctx.timers()
        res = {"G": G, "rows_per_rank_per_side": n, "owner_split_ms": round((t1 - t0) * 1e3, 2),
               "owner_split_phases": {k: round(v["ms"], 2) for k, v in split_t.items()},
               "join_on_1_over_G_of_rank_rows": l0.numel(), "join_ms": round((t3 - t2) * 1e3, 2),
               "join_ms_scaled_to_full_rank_share": round((t3 - t2) * 1e3 * G, 2),
               "join_phases": {k: round(v["ms"], 2) for k, v in join_t.items()}, "radix": st["radix_bits"]}
    print(json.dumps(res), flush=True)
    del fk, y, pk, x, lt, rt, l0, r0, ofk, oy, ox
    torch.cuda.empty_cache()
