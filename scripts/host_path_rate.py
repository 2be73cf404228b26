#!/usr/bin/env python3
"""PCIe-inclusive rates of the OLAP_MEM_HOST path (pageable numpy buffers in, host results
out): filter / sum / take / join at a few GiB, with the pinned staging ring (default, and a
sweep of copy threads) and with the plain per-buffer hipMemcpyAsync (host_stage=0).
Never the bench.py `value` -- DESIGN.md section 5 quotes it as the PCIe-inclusive figure."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpu_olap_amd import Device, FilterGpu, SumGpu, TakeGpu, JoinGpu

dev = Device(0)
rng = np.random.default_rng(1)
NB, ROWS = int(os.environ.get("NB", "8192")), 65536          # filter/sum: BM_Filter shape, SF = NB/128
flt = [rng.integers(0, 1 << 32, size=ROWS, dtype=np.uint32) for _ in range(NB)]
JB, JROWS = int(os.environ.get("JB", "64")), 2 << 20            # join: SF = JB
pk = [np.arange(b * JROWS, (b + 1) * JROWS, dtype=np.uint32) for b in range(JB)]
x = [rng.integers(0, 1 << 32, size=JROWS, dtype=np.uint32) for _ in range(JB)]
fk = [rng.integers(b * JROWS, (b + 1) * JROWS, size=JROWS, dtype=np.uint32) for b in range(JB)]
y = [rng.integers(0, 1 << 32, size=JROWS, dtype=np.uint32) for _ in range(JB)]
tk_data = [rng.integers(0, 1 << 32, size=4 << 20, dtype=np.uint32) for _ in range(32)]
tk_idx = [rng.integers(0, 4 << 20, size=512 << 10, dtype=np.uint32) for _ in range(32)]


def timed(fn, reps=2):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best


for stage, threads in ((0, 0), (1, 1), (1, 2), (1, 4), (1, 8), (1, 16)):
    dev.set_tuning("host_stage", stage)
    dev.set_tuning("host_copy_threads", threads)
    out = {"host_stage": stage, "copy_threads": threads}
    t = timed(lambda: FilterGpu(dev, flt).Prepare().GetResult())
    out["filter_GBps_in"] = round(NB * ROWS * 4 / t / 1e9, 2)
    out["filter_Mrows_s"] = round(NB * ROWS / t / 1e6, 1)
    t = timed(lambda: SumGpu(dev, flt).Prepare().Run())
    out["sum_GBps_in"] = round(NB * ROWS * 4 / t / 1e9, 2)
    t = timed(lambda: TakeGpu(dev, tk_data, tk_idx).Prepare().Run())
    out["take_GBps_in"] = round(32 * ((4 << 20) + (512 << 10)) * 4 / t / 1e9, 2)
    t = timed(lambda: JoinGpu(dev, list(zip(fk, y)), list(zip(pk, x))).Prepare().Run())
    out["join_Mrows_s"] = round(JB * JROWS / t / 1e6, 1)
    out["join_GBps_moved"] = round(JB * JROWS * 28 / t / 1e9, 2)   # 16 B in + 12 B out per row pair
    print(json.dumps(out), flush=True)
