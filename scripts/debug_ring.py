#!/usr/bin/env python3
"""Debug: uniform 2^25-row join with an 8+8 plan under each ring_scatter setting; prints attempts / overflow counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dpu_olap_amd as o
from dpu_olap_amd import datagen as dg
dev = o.Device(0)
rows, nb = 1 << 20, 32
bits = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,8").split(",")]
n = rows * nb
cols = {k: dev.empty(n, "uint32") for k in ("fk", "y", "pk", "x", "exp")}
dev.gen_fk_u32(cols["fk"].ptr, n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
dev.gen_uniform_u32(cols["y"].ptr, n, dg.SEED, dg.STREAM_Y, 0)
dev.gen_index_u32(cols["pk"].ptr, n, 0)
dev.gen_uniform_u32(cols["x"].ptr, n, dg.SEED, dg.STREAM_X, 0)
dev.gen_uniform_u32_at(cols["exp"].ptr, cols["fk"].ptr, n, dg.SEED, dg.STREAM_X)
expect = dev.checksum_rows3_u32(cols["fk"].ptr, cols["y"].ptr, cols["exp"].ptr, n)
batch = lambda c, b: o.DeviceArray(cols[c].ptr + 4 * b * rows, rows, "uint32")
dev.join_configure(*bits)
dev.set_tuning("claim_mode", 2)
for ring in (0, 1, 2, 3):
    for tpt in (8, 4):
        dev.set_tuning("ring_scatter", ring); dev.set_tuning("ring_tpt", tpt)
        j = o.JoinGpu(dev, [(batch("fk", b), batch("y", b)) for b in range(nb)], [(batch("pk", b), batch("x", b)) for b in range(nb)]).Prepare()
        out = j.Run()
        ok = dev.checksum_rows3_u32(out["fk"].ptr, out["y"].ptr, out["x"].ptr, n) == expect
        print("ring", ring, "tpt", tpt, "ok", ok, "attempts", dev.get_counter("join_attempts"), "overflows", dev.get_counter("join_region_overflows"),
              {k: round(v, 3) for k, v in j.Timers().items()}, j.stats["radix_bits"], flush=True)
        for c in out.values(): dev.release(o.DeviceArray(c.ptr, 0))
