#!/usr/bin/env python3
"""Asymmetric joins (large probe side, small build side: the fact x dimension shape) on one MI355X, device-resident:
left = 2^LOG2_LEFT rows, right = 2^k rows for several k; fk uniform over the right keys.  usage: join_asym_rate.py [log2_left]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpu_olap_amd import capi, datagen as dg  # noqa: E402

log2_left = int(sys.argv[1]) if len(sys.argv) > 1 else 28
nl = 1 << log2_left
ctx = capi.Context(0)
if os.environ.get("ASYM_BITS"):      # A/B: total radix bits instead of the automatic plan
    ctx.join_configure(int(os.environ["ASYM_BITS"]), 0)
br = 1 << 21
for k in ([int(v) for v in os.environ["ASYM_K"].split(",")] if os.environ.get("ASYM_K") else (8, 12, 13, 16, 20, 24, log2_left)):
    nr = 1 << k
    fk, y, pk, x = ctx.malloc(nl * 4), ctx.malloc(nl * 4), ctx.malloc(nr * 4), ctx.malloc(nr * 4)
    out = [ctx.malloc(nl * 4) for _ in range(3)]
    ctx.gen_fk_u32(fk, nl, dg.SEED, dg.STREAM_FK, 0, 1 << 62, nr)       # uniform in [0, nr)
    ctx.gen_uniform_u32(y, nl, dg.SEED, dg.STREAM_Y, 0)
    ctx.gen_index_u32(pk, nr, 0)
    ctx.gen_uniform_u32(x, nr, dg.SEED, dg.STREAM_X, 0)
    exp = ctx.malloc(nl * 4)
    ctx.gen_uniform_u32_at(exp, fk, nl, dg.SEED, dg.STREAM_X)
    want = ctx.checksum_rows3_u32(fk, y, exp, nl)
    ctx.free(exp)
    lb, rb = max(1, nl // br), max(1, nr // br)
    lrows, rrows = [nl // lb] * lb, [nr // rb] * rb
    tbl = lambda p, rows: [p + 4 * sum(rows[:i]) for i in range(len(rows))]
    run = lambda: ctx.join_u32(tbl(fk, lrows), tbl(y, lrows), lrows, tbl(pk, rrows), tbl(x, rrows), rrows, capi.MEM_DEVICE, out[0], out[1], out[2],
                               nl, capi.MEM_DEVICE)
    run()
    ctx.set_profiling(True)
    ctx.reset_timers()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        rows, st = run()
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    ok = rows == nl and ctx.checksum_rows3_u32(out[0], out[1], out[2], nl) == want
    print(json.dumps({"left_rows": nl, "right_rows": nr, "ms": round(ms, 3), "left_Grows_s": round(nl / ms / 1e6, 2), "radix_bits": st["radix_bits"],
                      "phases_ms": {k2: round(v["ms"] / 3, 3) for k2, v in ctx.timers().items()}, "verified": bool(ok)}), flush=True)
    for p in [fk, y, pk, x] + out:
        ctx.free(p)
