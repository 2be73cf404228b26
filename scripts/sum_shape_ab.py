#!/usr/bin/env python3
"""k_sum on device-resident columns of several batch geometries, in a chosen order (A/B of library builds through OLAP_GPU_LIB):
usage: sum_shape_ab.py ORDER   with ORDER a string over a = u32 2048 x 2Mi rows, b = u32 65536 x 64Ki, c = i64 2048 x 2Mi, d = u32 1024 x 4Mi, e = i64 4096 x 1Mi"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpu_olap_amd import capi, datagen as dg

SH = {"a": (2048, 2 << 20, 4), "b": (65536, 64 << 10, 4), "c": (2048, 2 << 20, 8), "d": (1024, 4 << 20, 4), "e": (4096, 1 << 20, 8)}
ctx = capi.Context(0)
if os.environ.get("PRE_ALLOC_GIB"):  # mimic bench_ops: two large buffers allocated, written and freed in front of the first case
    g = int(os.environ["PRE_ALLOC_GIB"]) << 30
    p1, p2 = ctx.malloc(g), ctx.malloc(g)
    ctx.gen_uniform_u32(p1, g // 4, dg.SEED, dg.STREAM_V, 0)
    ctx.gen_uniform_u32(p2, g // 4, dg.SEED, dg.STREAM_V, 0)
    ctx.synchronize()
    ctx.free(p1)
    ctx.free(p2)
for ch in (sys.argv[1] if len(sys.argv) > 1 else "abc"):
    nb, rows, eb = SH[ch]
    n = nb * rows
    col = ctx.malloc(n * eb)
    (ctx.gen_uniform_u32 if eb == 4 else ctx.gen_uniform_u64)(col, n, dg.SEED, dg.STREAM_V, 0)
    ptrs, rws = capi.ptr_table([col + eb * b * rows for b in range(nb)]), capi.u64_table([rows] * nb)
    code = capi.U32 if eb == 4 else capi.I64
    for _ in range(2):
        ctx.sum(code, ptrs, rws, capi.MEM_DEVICE)
    ctx.set_profiling(True)
    ctx.reset_timers()
    ctx.synchronize()
    t0 = time.perf_counter()
    steps = []
    for _ in range(5):
        t1 = time.perf_counter()
        ctx.sum(code, ptrs, rws, capi.MEM_DEVICE)
        steps.append(round((time.perf_counter() - t1) * 1e3, 2))
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(ch, nb, rows, eb, "col=%#x" % col, "steps", steps, "wall %.3f ms" % ms, "kernel %.3f ms" % (ctx.timers()["sum"]["ms"] / 5), "%.2f TB/s" % (n * eb / ms / 1e9), flush=True)
    ctx.free(col)
