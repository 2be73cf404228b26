#!/usr/bin/env python3
"""Pass 0 over RAGGED batches (rows not a multiple of the 4096-row tile) on one MI355X, device-resident: 1024 batches of
2 Mi - 1234 rows per side; ring_split_ragged 1 (whole tiles through the straight-line loader + tails) vs 0 (generic loader)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpu_olap_amd import capi, datagen as dg  # noqa: E402

nb, rows = 1024, (2 << 20) - 1234
n = nb * rows
ctx = capi.Context(0)
# default: every batch starts 16-byte aligned (its own Arrow buffer), its length is ragged; "unaligned": the batches follow each other
# without padding (slices of one buffer): every other one starts 8 bytes off a 16-byte boundary
stride = rows * 4 if "unaligned" in sys.argv else (rows * 4 + 15) // 16 * 16
fk, y, pk, x = (ctx.malloc(nb * stride + 64) for _ in range(4))
out = [ctx.malloc(n * 4) for _ in range(3)]
m = nb * stride // 4                       # the padding words between batches are generated too and simply not part of any batch
ctx.gen_fk_u32(fk, m, dg.SEED, dg.STREAM_FK, 0, 1 << 62, n)
ctx.gen_uniform_u32(y, m, dg.SEED, dg.STREAM_Y, 0)
ctx.gen_index_u32(pk, m, 0)                # keys 0 .. m-1 (unique); foreign keys are drawn from [0, n) <= m: all present except those
ctx.gen_uniform_u32(x, m, dg.SEED, dg.STREAM_X, 0)   # that fall on a padding word of pk -- so join with drop_unmatched and compare the modes
tbl = lambda p: [p + b * stride for b in range(nb)]
lens = [rows] * nb
want = None
ctx.join_set_unmatched(True)
for knob in (1, 0, 1, 0):
    ctx.set_tuning("ring_split_ragged", knob)
    run = lambda: ctx.join_u32(tbl(fk), tbl(y), lens, tbl(pk), tbl(x), lens, capi.MEM_DEVICE, out[0], out[1], out[2], n, capi.MEM_DEVICE)
    run()
    ctx.set_profiling(True)
    ctx.reset_timers()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        got, st = run()
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    digest = (got, ctx.checksum_rows3_u32(out[0], out[1], out[2], got))
    want = want or digest
    ok = digest == want                    # both loaders give the same rows
    print(json.dumps({"ring_split_ragged": knob, "rows_per_side": n, "ms": round(ms, 3), "phases_ms": {k: round(v["ms"] / 3, 3) for k, v in ctx.timers().items()},
                      "claim_kernel": ctx.phase_kernel("partition-scatter-claim"), "verified": bool(ok)}), flush=True)
