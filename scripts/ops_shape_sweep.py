#!/usr/bin/env python3
"""Shapes outside the benchmark registrations: does any batch geometry fall off a cliff?  filter / sum / take on 2^30 u32 rows
cut into batches of 2^k rows, take with few / many indices; device-resident; prints kernel ms and GB/s of algorithmic bytes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from dpu_olap_amd import capi, datagen as dg  # noqa: E402

ctx = capi.Context(0)
for kv in os.environ.get("OLAP_TUNE", "").split(","):   # e.g. OLAP_TUNE=filter_stores=1
    if "=" in kv:
        ctx.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
ONLY = os.environ.get("ONLY", "")
N = 1 << 30


def timed(fn, phases, steps=3):
    fn()
    ctx.set_profiling(True)
    ctx.reset_timers()
    for _ in range(steps):
        r = fn()
    ctx.synchronize()
    tm = ctx.timers()
    return r, sum(v["ms"] for k, v in tm.items() if k in phases) / steps


data = ctx.malloc(N * 4)
out = ctx.malloc(N * 4)
ctx.gen_uniform_u32(data, N, dg.SEED, dg.STREAM_V, 0)
for k in (30, 26, 21, 19, 18, 16, 12, 10):
    rows = 1 << k
    nb = N // rows
    if nb > (1 << 20):
        nb = 1 << 20       # cap the table size: 2^20 batches
    n = nb * rows
    pt = capi.ptr_table([data + 4 * b * rows for b in range(nb)])
    ot = capi.ptr_table([out + 4 * b * rows for b in range(nb)])
    rt = capi.u64_table([rows] * nb)
    _, ms = timed(lambda: ctx.sum(capi.U32, pt, rt, capi.MEM_DEVICE), ("sum",))
    print(json.dumps({"op": "sum", "batches": nb, "batch_rows": rows, "ms": round(ms, 3), "GBps": round(n * 4 / ms / 1e6, 1)}), flush=True)
    for thr, sel in ((1 << 30, 0.25), (0, 0.0), (0xFFFFFFFF, 1.0)):
        (_, total), ms = timed(lambda: ctx.filter_lt(capi.U32, pt, rt, capi.MEM_DEVICE, thr, ot, capi.MEM_DEVICE), ("filter-count", "filter-compact"))
        print(json.dumps({"op": "filter", "batches": nb, "batch_rows": rows, "selectivity": sel, "ms": round(ms, 3),
                          "GBps": round(n * 4 * (1 + sel) / ms / 1e6, 1), "rows_out": int(total)}), flush=True)
# take: one big data batch, index batches of various sizes; many small data batches with few indices each
idx = ctx.malloc((1 << 28) * 4)
for (nb, rows, m) in ((1, 1 << 30, 1 << 28), (16, 1 << 26, 1 << 24), (1 << 10, 1 << 20, 1 << 18), (1 << 16, 1 << 14, 1 << 12), (1 << 18, 1 << 12, 16), (1 << 18, 1 << 12, 1 << 10)):
    ni = nb * m
    ctx.gen_fk_u32(idx, ni, dg.SEED, dg.STREAM_IDX, 0, 1 << 62, rows)
    dpt = capi.ptr_table([data + 4 * b * rows for b in range(nb)])
    ipt = capi.ptr_table([idx + 4 * b * m for b in range(nb)])
    opt = capi.ptr_table([out + 4 * b * m for b in range(nb)])
    _, ms = timed(lambda: ctx.take(capi.U32, dpt, capi.u64_table([rows] * nb), ipt, capi.u64_table([m] * nb), capi.MEM_DEVICE, opt, capi.MEM_DEVICE), ("take",))
    print(json.dumps({"op": "take", "batches": nb, "data_rows": rows, "indices": m, "ms": round(ms, 3), "G_indices_s": round(ni / ms / 1e6, 2),
                      "GBps_12B": round(ni * 12 / ms / 1e6, 1)}), flush=True)
