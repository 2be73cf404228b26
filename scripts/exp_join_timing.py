#!/usr/bin/env python3
"""Timing-only runs of the SF-scaled BM_Join shape for EXPERIMENT builds whose results are wrong on purpose
(scripts/build_variant.sh with -DOLAP_EXP_*: e.g. the ring scatter without its line stores): unmatched rows are
dropped instead of failing, nothing is verified, only the per-phase HIP-event timers are printed.

  OLAP_GPU_LIB=$PWD/ab_libs/NAME.so python scripts/exp_join_timing.py [--sf 2048] [--steps 3] [--tune k=v ...]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--tune", action="append", default=[])
    ap.add_argument("--clocks", action="store_true", help="-DOLAP_EXP_PHASE_CLOCKS builds: print the ring scatter's phase clocks per tile")
    args = ap.parse_args()
    import torch
    from dpu_olap_amd import capi, datagen as dg
    from dpu_olap_amd.dist_join import GpuLocalOps
    device = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ops = GpuLocalOps(ctx, device)
    for kv in args.tune:
        k, _, v = kv.partition("=")
        ctx.set_tuning(k, int(v))
    rows = 2 << 20
    n = args.sf * rows
    fk, y, pk, x = ops.empty_columns(n, 4)
    ctx.gen_fk_u32(fk.data_ptr(), n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
    ctx.gen_uniform_u32(y.data_ptr(), n, dg.SEED, dg.STREAM_Y, 0)
    ctx.gen_index_u32(pk.data_ptr(), n, 0)
    ctx.gen_uniform_u32(x.data_ptr(), n, dg.SEED, dg.STREAM_X, 0)
    out = ops.empty_columns(n, 3)
    torch.cuda.synchronize()
    ptrs = lambda t: [t.data_ptr() + 4 * b * rows for b in range(args.sf)]
    ctx.join_set_unmatched(True)
    ctx.set_profiling(True)
    res = None
    for step in range(args.steps + 1):
        if step == 1:
            ctx.reset_timers()
        try:
            res = ctx.join_u32(ptrs(fk), ptrs(y), [rows] * args.sf, ptrs(pk), ptrs(x), [rows] * args.sf, capi.MEM_DEVICE,
                               out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), n, capi.MEM_DEVICE)
        except capi.OlapError as e:   # e.g. a checked promise that the experiment breaks: the timers still count
            res = (None, {"error": str(e)[:80]})
    t = ctx.timers()
    if args.clocks:
        import ctypes
        buf = (ctypes.c_ulonglong * 32)()
        assert ctx.lib.olap_debug_ring_clocks(buf) == 0
        names = ["P insert", "P wait b1", "P hash+load", "P wait b2", "F wait b1", "F flush", "F wait b2", "tiles",
                 "P wait tile+1", "P hash | hot tiles (ring6)", "P claims | F emit (ring6)", "P load issue | F lines (ring6)", "hot tiles (claim)", "F list (claim)", "F emit (claim)", "F lines (claim)"]
        for role, label in ((0, "claim pass"), (1, "capacity pass")):
            v = list(buf[role * 16:role * 16 + 16])
            if v[7]:
                print(label, {n: round(x / v[7], 1) for n, x in zip(names, v) if n != "tiles"}, "cycles per tile;", v[7], "tiles sampled")
    print(json.dumps({"phase_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in t.items() if v["launches"]},
                      "attempts": ctx.get_counter("join_attempts"), "rows_out": res[0]}))
    ctx.close()


if __name__ == "__main__":
    main()
