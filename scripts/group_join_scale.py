#!/usr/bin/env python3
"""The C-ABI group join (olap_group_join_u32: split, exchange, per-rank join inside the library) at scale on ONE GPU: G logical
ranks on device 0 (loopback, peer copies), device-resident generator columns, digest verified.  usage: group_join_scale.py [SF] [G]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpu_olap_amd import Device, capi, datagen as dg

sf, ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = Device(0)
rows = 2 << 20
n = rows * sf
cols = {k: dev.empty(n, "uint32") for k in ("fk", "y", "pk", "x", "exp")}
dev.gen_fk_u32(cols["fk"].ptr, n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
dev.gen_uniform_u32(cols["y"].ptr, n, dg.SEED, dg.STREAM_Y, 0)
dev.gen_index_u32(cols["pk"].ptr, n, 0)
dev.gen_uniform_u32(cols["x"].ptr, n, dg.SEED, dg.STREAM_X, 0)
dev.gen_uniform_u32_at(cols["exp"].ptr, cols["fk"].ptr, n, dg.SEED, dg.STREAM_X)
expect = dev.checksum_rows3_u32(cols["fk"].ptr, cols["y"].ptr, cols["exp"].ptr, n)
dev.synchronize()
g = capi.Group([0] * ranks)
out = [np.empty(n, dtype=np.uint32) for _ in range(3)]
bp = lambda c: [cols[c].ptr + 4 * b * rows for b in range(sf)]
for rep in range(2):
    t0 = time.perf_counter()
    got_rows, per_rank, st = g.join_u32(bp("fk"), bp("y"), [rows] * sf, bp("pk"), bp("x"), [rows] * sf, capi.MEM_DEVICE, *(a.ctypes.data for a in out), n)
    ms = (time.perf_counter() - t0) * 1e3
d = [dev.to_device(a) for a in out]
ok = dev.checksum_rows3_u32(d[0].ptr, d[1].ptr, d[2].ptr, n) == expect and got_rows == n and st["missing_keys"] == 0
print({"sf": sf, "ranks": ranks, "ms_incl_copy_out": round(ms, 1), "phase_ms": {k: round(v, 2) for k, v in g.last_phase_ms().items()},
       "verified": bool(ok), "stats": {k: st[k] for k in ("passes", "radix_bits", "nr_partitions", "table_slots")}, "transport": g.last_transport() if hasattr(g, "last_transport") else None})
g.close()
