#!/usr/bin/env python3
"""Joins with a skewed probe side on one MI355X, device-resident: left = 2^27 rows of which a fraction h carries ONE hot
foreign key (the rest uniform over the right keys).  usage: join_skew_rate.py [log2 left rows = 27] [log2 right rows = 24]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpu_olap_amd import capi, datagen as dg  # noqa: E402

nl, nr = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 27), 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 24)
ctx = capi.Context(0)
br = 1 << 21
rng = np.random.default_rng(5)
for hot in (0.0, 0.01, 0.1, 0.5):
    fk_h = rng.integers(0, nr, size=nl, dtype=np.uint32)
    fk_h[rng.random(nl) < hot] = 12345
    fk, y, pk, x = ctx.malloc(nl * 4), ctx.malloc(nl * 4), ctx.malloc(nr * 4), ctx.malloc(nr * 4)
    out = [ctx.malloc(nl * 4) for _ in range(3)]
    ctx.h2d(fk, fk_h.ctypes.data, fk_h.nbytes)
    ctx.gen_uniform_u32(y, nl, dg.SEED, dg.STREAM_Y, 0)
    ctx.gen_index_u32(pk, nr, 0)
    ctx.gen_uniform_u32(x, nr, dg.SEED, dg.STREAM_X, 0)
    exp = ctx.malloc(nl * 4)
    ctx.gen_uniform_u32_at(exp, fk, nl, dg.SEED, dg.STREAM_X)
    want = ctx.checksum_rows3_u32(fk, y, exp, nl)
    ctx.free(exp)
    lrows, rrows = [br] * (nl // br), [br] * (nr // br)
    tbl = lambda p, rows: [p + 4 * br * i for i in range(len(rows))]
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
    print(json.dumps({"hot_fraction": hot, "ms": round(ms, 3), "left_Grows_s": round(nl / ms / 1e6, 2), "radix_bits": st["radix_bits"],
                      "attempts": ctx.get_counter("join_attempts"), "phases_ms": {k2: round(v["ms"] / 3, 3) for k2, v in ctx.timers().items()},
                      "verified": bool(ok)}), flush=True)
    for p in [fk, y, pk, x] + out:
        ctx.free(p)
