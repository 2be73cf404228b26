#!/usr/bin/env python3
"""Rate of olap_join_u64 (64-bit keys, HT_64BIT_KEYS in the reference) on one MI355X with device-resident operands.
Keys: pk[i] = i * ODD mod 2^64 (a bijection, so unique), fk = pk[random rows], payload x[i] = i, y = row index of the left
table; checked by rows == n and sum(out_x) == sum(chosen rows).  usage: join64_rate.py [--log2-rows 27] [--steps 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpu_olap_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-rows", type=int, default=27)
    ap.add_argument("--log2-batch", type=int, default=21)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--ring", type=int, default=1, help="join64_ring tuning: 0 exact-layout pipeline, 1 automatic, 2 forced")
    ap.add_argument("--tpt", type=int, default=8)
    ap.add_argument("--bits", type=int, default=-1)
    ap.add_argument("--probe", default="", help="A/B of the probe: comma list of variant:grid pairs (probe64_variant, probe64_grid), e.g. 0:8,1:3")
    a = ap.parse_args()
    n, br = 1 << a.log2_rows, 1 << a.log2_batch
    nb = n // br
    ctx = capi.Context(0)
    ctx.set_tuning("join64_ring", a.ring)
    ctx.set_tuning("join64_tpt", a.tpt)
    ctx.join_configure(a.bits, 0)
    rng = np.random.default_rng(3)
    odd = np.uint64(0x9E3779B97F4A7C15)
    rows = np.arange(n, dtype=np.uint64)
    pk = rows * odd
    pick = rng.integers(0, n, size=n, dtype=np.int64)
    fk = pk[pick]
    want_sum = int(pick.sum())
    cols = {}
    for name, arr in (("fk", fk), ("y", np.arange(n, dtype=np.uint32)), ("pk", pk), ("x", np.arange(n, dtype=np.uint32))):
        p = ctx.malloc(arr.nbytes)
        ctx.h2d(p, arr.ctypes.data, arr.nbytes)
        cols[name] = (p, arr.itemsize)
    out_fk, out_y, out_x = ctx.malloc(n * 8), ctx.malloc(n * 4), ctx.malloc(n * 4)
    tbl = lambda name: [cols[name][0] + b * br * cols[name][1] for b in range(nb)]
    run = lambda: ctx.join_u64(tbl("fk"), tbl("y"), [br] * nb, tbl("pk"), tbl("x"), [br] * nb, capi.MEM_DEVICE, out_fk, out_y, out_x,
                               n, capi.MEM_DEVICE)
    for cfg in (a.probe.split(",") if a.probe else [""]):
        if cfg:
            var, grid = (int(v) for v in cfg.split(":"))
            ctx.set_tuning("probe64_variant", var)
            ctx.set_tuning("probe64_grid", grid)
        try:
            run()
        except capi.OlapError as e:
            if not cfg:
                raise
            print(json.dumps({"probe": cfg, "error": str(e)[:120]}))  # (the diagnostic variants produce wrong rows)
        ctx.set_profiling(True)
        ctx.reset_timers()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            got, stats = run()
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        hx = np.empty(n, dtype=np.uint32)
        ctx.d2h(hx.ctypes.data, out_x, hx.nbytes)
        ok = got == n and int(hx.astype(np.int64).sum()) == want_sum
        tm = {k: round(v["ms"] / a.steps, 3) for k, v in ctx.timers().items()}
        # compulsory bytes per row pair: 12 + 12 read, 16 written
        print(json.dumps({"probe": cfg, "metric": "join_u64 rows/s (left rows per second)", "rows_per_side": n, "ms_per_join": round(ms, 3),
                          "value_Grows_s": round(n / ms / 1e6, 2), "algorithmic_GBps": round(n * 40 / ms / 1e6, 1), "verified": bool(ok),
                          "moved_bytes_per_pair": 136 if "probe12" in ctx.phase_kernel("join-probe") else 224,
                          "attempts": ctx.get_counter("join_attempts"), "probe_kernel": ctx.phase_kernel("join-probe"), "stats": stats, "phase_ms": tm}))


if __name__ == "__main__":
    main()
