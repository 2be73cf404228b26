#!/usr/bin/env python3
"""bench_ops.py -- BM_Filter / BM_Sum / BM_Take equivalents on one MI355X (BASELINE.json
configs[1..2]); companion of bench.py (which measures the headline BM_Join metric).

Shapes follow the reference's benchmark registrations at scale factor SF (default 2048):
  BM_Filter {Batches=SF<<7, Batch-Size=64Ki}            host/filter/filter_benchmark.cc:153
  BM_Sum    {SF, 2Mi} and {SF<<5, 64Ki}                 host/aggr/aggr_benchmark.cc:148-150
  BM_Take   {SF, 4Mi, 512Ki} and {SF<<6, 64Ki, 8Ki}     host/take/take_benchmark.cc:157-159
u32 columns (the reference's element type) plus an int64 run of sum and take.  Inputs are
generated in HBM; items/bytes accounting as in the reference (take counts DATA rows);
roofline = algorithmic bytes / kernel time / 8 TB/s.  Prints one JSON line
per benchmark.  Results are verified by order-independent digests / the generator.

Multi-GPU (these operators shard by batch with no communication): launched as
`python -m torch.distributed.run --nproc-per-node N bench_ops.py`, every rank takes a
contiguous 1/N of the batches on its own GPU; the ranks only meet for the barrier, the
max-over-ranks time and (sum) the final 8-byte add -- over gloo, no data-path collective."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--only", default="")
    ap.add_argument("--cpu", action="store_true", help="also time Arrow compute (CPU) on a bounded sample")
    ap.add_argument("--tune", default="", help="comma-separated olap_set_tuning pairs for A/B runs, e.g. take_variant=5")
    args = ap.parse_args()
    import numpy as np
    from dpu_olap_amd import capi
    from dpu_olap_amd import datagen as dg

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev_index = 0 if os.environ.get("OLAP_BENCH_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")   # scalars only: barrier, max time, sum of results
    ctx = capi.Context(dev_index)
    info = ctx.device_info()
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        ctx.set_tuning(k, int(v))
    sf = args.sf
    if sf % world:
        raise SystemExit(f"SF={sf} must be a multiple of the number of GPUs ({world})")

    def gather(x):   # every rank's value of a small Python object
        if world == 1:
            return [x]
        out = [None] * world
        dist.all_gather_object(out, x)
        return out

    def timed(fn, phase):
        for _ in range(args.warmup):
            fn()
        ctx.set_profiling(True)
        ctx.reset_timers()
        ctx.synchronize()
        # (Wall time per step is the mean over `steps`: ONE step in a few dozen takes 7-18 ms longer on these boxes -- GPU idle, kernel
        # times unchanged, either library build, profiles/r3_sum_wall_diag.txt -- which shows in a 3 ms step and not in a 67 ms one;
        # phase_ms carries the kernel time next to it.)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = fn()
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / args.steps
        tm = ctx.timers()
        kern = sum(v["ms"] for k, v in tm.items() if k in phase) / args.steps * 1e-3
        wall, kern = max(gather(wall)), max(gather(kern))   # the slowest rank
        return r, wall, kern, {k: round(v["ms"] / args.steps, 4) for k, v in tm.items()}

    def emit(name, rows, elem, alg_bytes, wall, kern, extra):
        # rows / alg_bytes: this rank's share; the job moved world times that in the slowest rank's time
        extra = dict(extra, verified=all(gather(bool(extra.get("verified", True)))))
        if rank:
            return
        name = name.replace("GPUs:1", f"GPUs:{world}")
        line = {
            "metric": name, "value": round(world * rows / wall / 1e6, 1), "unit": "Mrows/s", "n_gpus": world, "scaling": "strong",
            "ms_per_step": round(wall * 1e3, 3), "kernel_ms": round(kern * 1e3, 3),
            "bytes_processed_GBps": round(world * rows * elem / wall / 1e9, 1),  # reference accounting: rows x elem size
            "roofline": {"bound": "hbm", "achieved": round(alg_bytes / kern / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(alg_bytes / kern / 1e9 / HBM_PEAK_GBPS, 4), "algorithmic_bytes": alg_bytes,
                         "note": "per GPU: this rank's algorithmic bytes over the slowest rank's kernel time"},
            "sf": sf, "device": info["name"].strip(), "data": "synthetic",
        }
        line.update(extra)
        print(json.dumps(line), flush=True)

    want = lambda n: not args.only or args.only in n

    # ------------------------------------------------------------------ filter
    if want("filter"):
        nb, rows = (sf << 7) // world, 64 << 10          # this rank's batches: [rank * nb, (rank + 1) * nb)
        n = nb * rows
        first = rank * n
        col = ctx.malloc(n * 4)
        out = ctx.malloc(n * 4)  # worst case per batch, like the reference's per-DPU output buffer
        ctx.gen_uniform_u32(col, n, dg.SEED, dg.STREAM_V, first)
        outs_l = [out + 4 * b * rows for b in range(nb)]
        ptrs = capi.ptr_table([col + 4 * b * rows for b in range(nb)])
        outs = capi.ptr_table(outs_l)
        rws = capi.u64_table([rows] * nb)
        # uint32 (the reference's element type: 25 % pass) and the same bits read as Arrow int32 (signed compare: 75 % pass)
        for code, npdt, sel in ((capi.U32, np.uint32, 0.25), (capi.I32, np.int32, 0.75)):
            (counts, total), wall, kern, tm = timed(lambda: ctx.filter_lt(code, ptrs, rws, capi.MEM_DEVICE, 1 << 30, outs, capi.MEM_DEVICE),
                                                   ("filter-count", "filter-compact"))
            # verify a sample of batches against the generator mirror + global count plausibility
            ok = abs(total / n - sel) < 1e-3
            for b in (0, nb // 2, nb - 1):
                h = np.empty(counts[b], dtype=npdt)
                ctx.d2h(h.ctypes.data, outs_l[b], h.nbytes)
                src = dg.uniform_u32(rows, dg.SEED, dg.STREAM_V, first + b * rows).view(npdt)
                ok = ok and np.array_equal(h, src[src < (1 << 30)])
            emit(f"BM_FilterGpu/Batches:{nb * world}/Batch-Size:{rows}/{np.dtype(npdt).name}/GPUs:1", n, 4, 4 * n + 4 * total, wall, kern,
                 {"results": sum(gather(total)), "selectivity": round(total / n, 5), "verified": bool(ok), "phase_ms": tm,
                  "note": "wall includes the host loop over the batch tables (262144 batches at SF=2048)"})
        ctx.free(col)
        ctx.free(out)

    # --------------------------------------------------------------------- sum
    if want("sum"):
        for (nb, rows, dt, code, gen) in ((sf, 2 << 20, np.uint32, capi.U32, ctx.gen_uniform_u32),
                                          (sf << 5, 64 << 10, np.uint32, capi.U32, ctx.gen_uniform_u32),
                                          (sf, 2 << 20, np.int64, capi.I64, ctx.gen_uniform_u64)):
            eb = np.dtype(dt).itemsize
            nb //= world
            n = nb * rows
            col = ctx.malloc(n * eb)
            gen(col, n, dg.SEED, dg.STREAM_V, rank * n)
            ptrs = capi.ptr_table([col + eb * b * rows for b in range(nb)])
            rws = capi.u64_table([rows] * nb)
            total, wall, kern, tm = timed(lambda: ctx.sum(code, ptrs, rws, capi.MEM_DEVICE), ("sum",))
            s, _, _ = ctx.checksum(col, n, eb)  # independent reduction kernel (digest): sum mod 2^64
            if code == capi.I64 and s >= 1 << 63:
                s -= 1 << 64
            whole = sum(gather(total))          # the final add of the per-device partials (host/aggr/aggr_dpu.cc:82-84)
            if code == capi.I64:
                whole = (whole + (1 << 63)) % (1 << 64) - (1 << 63)
            emit(f"BM_SumGpu/Batches:{nb * world}/Batch-Size:{rows}/{np.dtype(dt).name}/GPUs:1", n, eb, eb * n, wall, kern,
                 {"result": whole, "verified": bool(total == s), "phase_ms": tm})
            ctx.free(col)

    # -------------------------------------------------------------------- take
    if want("take"):
        for (nb, rows, m, dt, code) in ((sf, 4 << 20, 512 << 10, np.uint32, capi.U32),
                                        (sf << 6, 64 << 10, 8 << 10, np.uint32, capi.U32),
                                        (sf, 4 << 20, 512 << 10, np.int64, capi.I64)):
            eb = np.dtype(dt).itemsize
            nb //= world
            n, ni = nb * rows, nb * m
            data, idx, out = ctx.malloc(n * eb), ctx.malloc(ni * 4), ctx.malloc(ni * eb)
            (ctx.gen_uniform_u32 if eb == 4 else ctx.gen_uniform_u64)(data, n, dg.SEED, dg.STREAM_V, rank * n)
            # indices uniform in [0, rows): fk generator with one "left batch" spanning everything
            ctx.gen_fk_u32(idx, ni, dg.SEED, dg.STREAM_IDX, rank * ni, 1 << 62, rows)
            ip = [idx + 4 * b * m for b in range(nb)]
            op = [out + eb * b * m for b in range(nb)]
            dpt, ipt, opt = capi.ptr_table([data + eb * b * rows for b in range(nb)]), capi.ptr_table(ip), capi.ptr_table(op)
            drt, irt = capi.u64_table([rows] * nb), capi.u64_table([m] * nb)
            _, wall, kern, tm = timed(lambda: ctx.take(code, dpt, drt, ipt, irt, capi.MEM_DEVICE, opt, capi.MEM_DEVICE), ("take",))
            ok = True
            for b in (0, nb - 1):
                hi = np.empty(m, dtype=np.uint32)
                ho = np.empty(m, dtype=dt)
                ctx.d2h(hi.ctypes.data, ip[b], hi.nbytes)
                ctx.d2h(ho.ctypes.data, op[b], ho.nbytes)
                src = (dg.uniform_u32 if eb == 4 else dg.uniform_u64)(rows, dg.SEED, dg.STREAM_V, rank * n + b * rows)
                ok = ok and np.array_equal(ho, src[hi].astype(dt) if eb == 4 else src[hi].view(np.int64))
            emit(f"BM_TakeGpu/Batches:{nb * world}/Batch-Size:{rows}/Indices-Batch-Size:{m}/{np.dtype(dt).name}/GPUs:1", n, eb,
                 ni * (4 + 2 * eb), wall, kern, {"indices": ni, "verified": bool(ok), "phase_ms": tm,
                                                 "note": "items = DATA rows as in take_benchmark.cc:54-56; roofline on 4+2w bytes per index"})
            for p in (data, idx, out):
                ctx.free(p)

    # ------------------------------------------------------- CPU side by side
    if args.cpu and rank == 0:
        # the Arrow CPU engine (what the reference's *Native operators delegate to) on a bounded sample
        import pyarrow as pa
        import pyarrow.compute as pc
        nb, rows = 256, 64 << 10
        arrs = [pa.array(dg.uniform_u32(rows, dg.SEED, dg.STREAM_V, b * rows)) for b in range(nb)]
        t0 = time.perf_counter()
        tot = sum(len(pc.filter(a, pc.less(a, pa.scalar(1 << 30, pa.uint32())))) for a in arrs)
        dt = time.perf_counter() - t0
        print(json.dumps({"metric": f"cpu_baseline filter (Arrow {pa.__version__} compute, 1 thread)", "value": round(nb * rows / dt / 1e6, 1),
                          "unit": "Mrows/s", "sample": f"{nb} x 64Ki rows", "results": tot}))
        t0 = time.perf_counter()
        tot = sum(pc.sum(a).as_py() for a in arrs)
        dt = time.perf_counter() - t0
        print(json.dumps({"metric": f"cpu_baseline sum (Arrow {pa.__version__} compute, 1 thread)", "value": round(nb * rows / dt / 1e6, 1),
                          "unit": "Mrows/s", "sample": f"{nb} x 64Ki rows", "result": tot}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
