#!/usr/bin/env python3
"""Experiment: scatter time per pass for 2^32 tuples with 9+9 radix bits (512 buckets per
pass), 128-byte vs 64-byte write-combining lines, and 10+10 bits (1024 buckets, 64-byte)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpu_olap_amd import capi, datagen as dg
ROWS = 2 << 20
sf = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = capi.Context(0); dev = torch.device("cuda", 0)
n = sf * ROWS
fk = torch.empty(n, dtype=torch.int32, device=dev); y = torch.empty(n, dtype=torch.int32, device=dev)
out = torch.empty(n + 2, dtype=torch.int64, device=dev)
ctx.gen_fk_u32(fk.data_ptr(), n, dg.SEED, dg.STREAM_FK, 0, ROWS, ROWS)
ctx.gen_uniform_u32(y.data_ptr(), n, dg.SEED, dg.STREAM_Y, 0)
kp = capi.ptr_table([fk.data_ptr() + 4 * b * ROWS for b in range(sf)])
pp = capi.ptr_table([y.data_ptr() + 4 * b * ROWS for b in range(sf)])
rw = capi.u64_table([ROWS] * sf)
torch.cuda.synchronize()
for bits, l16, geo in ((18, 0, 1), (18, 0, 2), (20, 0, 1), (16, 0, 2), (14, 0, 2)):
    ctx.set_tuning("line16", l16)
    ctx.set_tuning("wc_geometry", geo)
    for rep in range(2):
        ctx.set_profiling(True); ctx.reset_timers()
        offs = ctx.partition_tuples_u32(kp, pp, rw, capi.MEM_DEVICE, 1 << bits, out.data_ptr(), n)
        t = ctx.timers()
    sizes = [offs[i + 1] - offs[i] for i in range(0, 1 << bits, max(1, (1 << bits) // 64))]
    print(json.dumps({"bits": bits, "line16": l16, "geometry": geo, "phases": {k: [round(v["ms"], 2), v["launches"]] for k, v in t.items()},
                      "total_rows": offs[-1], "max_sampled_partition": max(sizes)}), flush=True)
