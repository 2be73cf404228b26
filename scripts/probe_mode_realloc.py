#!/usr/bin/env python3
"""Does the packed probe's slow mode (18.0 vs 20.3 ms at SF=2048) belong to the PROCESS or to the ALLOCATION?
One process creates a context, generates the SF=2048 columns, joins a few times, destroys the context -- several times over --
and prints the probe time of every cycle with the addresses of its buffers.  usage: probe_mode_realloc.py [CYCLES] [SF]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpu_olap_amd import capi, datagen as dg

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sf = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
rows = 2 << 20
n = sf * rows
for c in range(cycles):
    ctx = capi.Context(0)
    cols = {k: ctx.malloc(n * 4) for k in ("fk", "y", "pk", "x", "o0", "o1", "o2")}
    ctx.gen_fk_u32(cols["fk"], n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
    ctx.gen_uniform_u32(cols["y"], n, dg.SEED, dg.STREAM_Y, 0)
    ctx.gen_index_u32(cols["pk"], n, 0)
    ctx.gen_uniform_u32(cols["x"], n, dg.SEED, dg.STREAM_X, 0)
    bp = lambda k: [cols[k] + 4 * b * rows for b in range(sf)]
    run = lambda: ctx.join_u32(bp("fk"), bp("y"), [rows] * sf, bp("pk"), bp("x"), [rows] * sf, capi.MEM_DEVICE, cols["o0"], cols["o1"], cols["o2"], n,
                               capi.MEM_DEVICE)
    run()
    ctx.set_profiling(True)
    ctx.reset_timers()
    for _ in range(3):
        got, st = run()
    ctx.synchronize()
    tm = {k: round(v["ms"] / 3, 2) for k, v in ctx.timers().items()}
    print({"cycle": c, "rows": got, "missing": st["missing_keys"], "phase_ms": tm, "out0": hex(cols["o0"]), "fk": hex(cols["fk"])}, flush=True)
    for p in cols.values():
        ctx.free(p)
    ctx.close()
