#!/usr/bin/env python3
"""Experiment: is pass 0 of the join slower because it reads two column streams (SRC_COLS)
or because its output spreads over the whole buffer?  Times the join at SF=2048 from
columns and from pre-zipped tuples (olap_join_tuples_u32)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpu_olap_amd import capi, datagen as dg
from dpu_olap_amd.dist_join import GpuLocalOps
ROWS = 2 << 20
sf = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = capi.Context(0); dev = torch.device("cuda", 0); ops = GpuLocalOps(ctx, dev)
n = sf * ROWS
col = lambda: torch.empty(n, dtype=torch.int32, device=dev)
fk, y, pk, x = col(), col(), col(), col()
ctx.gen_fk_u32(fk.data_ptr(), n, dg.SEED, dg.STREAM_FK, 0, ROWS, ROWS)
ctx.gen_uniform_u32(y.data_ptr(), n, dg.SEED, dg.STREAM_Y, 0)
ctx.gen_index_u32(pk.data_ptr(), n, 0)
ctx.gen_uniform_u32(x.data_ptr(), n, dg.SEED, dg.STREAM_X, 0)
# zip without reordering much: interleave via torch (stack) -> int64 view
lt = torch.stack([fk, y], dim=1).contiguous().view(torch.int64).view(-1)
rt = torch.stack([pk, x], dim=1).contiguous().view(torch.int64).view(-1)
del fk, y, pk, x
torch.cuda.empty_cache()
for rep in range(3):
    ctx.set_profiling(True); ctx.reset_timers()
    ofk, oy, ox, st = ops.join_tuples(lt, rt, 0)
    ctx.synchronize()
    t = ctx.timers()
    print(json.dumps({"src": "tuples", "rep": rep, "phases": {k: [round(v["ms"], 2), v["launches"]] for k, v in t.items()}, "radix": st["radix_bits"]}), flush=True)
    del ofk, oy, ox
