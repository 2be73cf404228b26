#!/usr/bin/env python3
"""bench.py -- BM_Join-equivalent (radix-partitioned PK-FK hash join, u32 columns) on
N MI355X GPUs, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W                      (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W      (N GPUs, RCCL)

Workload (BASELINE.json configs[3], reference host/join/join_benchmark.cc:145-171):
BM_Join {Batches=SF, L-Batch-Size=2Mi, R-Batch-Size=2Mi}, SF=2048 -> 2^32 rows per
side; right {pk = global row index, x uniform u32}, left {fk uniform in the matching
right batch's key range, y uniform u32}; synthetic, generated in HBM by the counter-
based generator (dpu_olap_amd/csrc/gen.hip).  One "step" = one full join (partition of
both tables + build + probe + materialise {fk, y, x}) of the resident inputs, Prepare()
inside the timed region like the reference.  value = (left rows + right rows) / time.
At N > 1 the SAME total problem (SF=2048) is split over the ranks (strong scaling:
SF=16384 of BASELINE configs[4] needs 2^35 unique keys, which uint32 cannot hold --
SURVEY.md gotcha 5) and the ranks exchange tuples once (all-to-all over xGMI).

Prints ONE JSON line on rank 0.  The oracle is used only for the cpu_baseline leg.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH_ROWS = 2 << 20           # join_benchmark.cc:170-171
HBM_PEAK_GBPS = 8000.0
MEASURED_READ_CEILING_GBPS = 6200.0   # streaming read measured on MI355X (our k_sum: 5.98-6.1 TB/s, k_part_hist: 6.2 TB/s)         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
JOIN_ALG_BYTES_PER_PAIR = 28   # read 2 cols x 2 tables (16 B) + write 3 cols (12 B)
SCATTER_ALG_BYTES_PER_TUPLE = 16  # dominant kernel: one (key,payload) tuple read + written once


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sf", type=int, default=0, help="scale factor (default 2048, or the largest that fits)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--radix-bits", type=int, default=-1, help="tuning: total radix bits (default auto)")
    ap.add_argument("--max-bits-per-pass", type=int, default=0, help="tuning: fan-out cap per pass (default 10)")
    ap.add_argument("--scatter-variant", type=int, default=0, help="tuning: 0 write-combining, 1 tile-staged")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT", help="olap_set_tuning knob for A/B runs (e.g. claim_mode=0)")
    ap.add_argument("--cpu-sf", type=int, default=0, help="scale factor of the bounded CPU sample (default: auto)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N=1 only: a process group of one rank over RCCL, and the join takes the multi-GPU route (fused split -> "
                         "all_to_all_single -> pre-split join) -- rehearses the exchange code on a one-GPU box")
    ap.add_argument("--exchange-order", default="", help="multi-GPU: issue order of the exchanges, 'table' or 'group' (default: dist_join's)")
    return ap.parse_args()


def cpu_baseline(cores_hint: int, cpu_sf: int):
    """CPU baselines on this box's host cores, on a bounded sample of the workload shape (never the thing measured):
      * "port": the oracle (CPU restatement of the reference's partition / build / probe / take pipeline,
        oracle/olap_oracle.c, OpenMP over batches and partitions), 3 runs, best and median;
      * "arrow": the reference's own CPU path, BM_JoinNative (host/join/join_benchmark.cc:114-125,161-162: Arrow's Acero
        hash join, Partitioned = 1 and 0, {SF<<5 batches x 64Ki rows per side}) run from the C++ harness
        (dpu_olap_amd/host/build/olap-query-host, Arrow 25 from the pyarrow wheel), 3 repetitions, thread count from
        the harness' own context.
    Both use every hardware thread of the box; the counts are printed."""
    import subprocess
    import tempfile
    import numpy as np
    import oracle
    from dpu_olap_amd import datagen as dg

    threads = oracle.max_threads()
    sf = cpu_sf or max(2, min(32, 1 << (max(threads, 1).bit_length())))
    rows = BATCH_ROWS
    sp = lambda fn: [fn(b) for b in range(sf)]
    pk = sp(lambda b: dg.index_u32(rows, b * rows))
    x = sp(lambda b: dg.uniform_u32(rows, dg.SEED, dg.STREAM_X, b * rows))
    fk = sp(lambda b: dg.fk_u32(rows, dg.SEED, dg.STREAM_FK, b * rows, rows, rows))
    y = sp(lambda b: dg.uniform_u32(rows, dg.SEED, dg.STREAM_Y, b * rows))
    n = sf * rows
    P = max(2, 1 << (max(sf, 2) - 1).bit_length())  # reference: #partitions == #batches (join_dpu.cc:168)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        ofk, oy, ox = oracle.join(fk, y, pk, x, nr_partitions=P, threads=threads)
        times.append(time.perf_counter() - t0)
        assert ofk.size == n
        del ofk, oy, ox
    times.sort()
    out = {
        "value": round(2 * n / times[0] / 1e6, 2), "unit": "Mrows/s", "cores": threads, "kind": "port",
        "median": round(2 * n / times[1] / 1e6, 2),
        "sample": f"BM_Join shape at SF={sf} ({sf} batches x 2Mi rows per side, {P} partitions), oracle/olap_oracle.c, "
                  f"best of 3 runs ({times[0]:.2f} / {times[1]:.2f} / {times[2]:.2f} s)",
    }
    del pk, x, fk, y
    arrow = None
    try:
        binp = os.path.join(ROOT, "dpu_olap_amd", "host", "build", "olap-query-host")
        if not os.path.exists(binp):
            raise RuntimeError("dpu_olap_amd/host/build/olap-query-host is not built (python -m dpu_olap_amd.build --host)")
        nsf = max(1, min(16, sf))       # BM_JoinNative: {SF<<5 batches x 64Ki rows} = SF x 2Mi rows per side
        with tempfile.TemporaryDirectory() as td:
            oj = os.path.join(td, "native.json")
            r = subprocess.run([binp, "--benchmark_filter=BM_JoinNative", f"--benchmark_out={oj}", "--benchmark_out_format=json",
                                "--benchmark_repetitions=3"], env=dict(os.environ, SF=str(nsf), NR_GPUS="1"), capture_output=True,
                               text=True, timeout=900)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-500:])
            d = json.load(open(oj))
        nrows = (nsf << 5) * (64 << 10)
        arrow = {"unit": "Mrows/s", "cores": int(d["context"].get("arrow_cpu_threads", 0)),
                 "hardware_concurrency": int(d["context"].get("hardware_concurrency", 0)),
                 "kind": "reference CPU path: BM_JoinNative (Arrow Acero hash join, host/join/join_native.cc:31-49 restated on Arrow 25), "
                         "C++ harness, 3 repetitions",
                 "sample": f"SF={nsf}: {nsf << 5} batches x 64Ki rows per side ({nrows} rows/side)"}
        for part in (1, 0):
            runs = [b for b in d["benchmarks"] if f"Partitioned:{part}" in b["name"] and b.get("run_type", "iteration") == "iteration"
                    and not b.get("error_occurred")]
            ms = sorted(b["real_time"] for b in runs)
            if ms:
                arrow[f"partitioned_{part}"] = {"value": round(2 * nrows / (ms[0] * 1e-3) / 1e6, 2), "median": round(2 * nrows / (ms[len(ms) // 2] * 1e-3) / 1e6, 2),
                                                "ms": [round(v, 1) for v in ms]}
        arrow["value"] = max(arrow.get("partitioned_1", {}).get("value", 0), arrow.get("partitioned_0", {}).get("value", 0))
    except Exception as e:  # pragma: no cover
        arrow = {"error": repr(e)}
    return out, arrow


def alg_bytes_per_unit(phase, kernel, default):
    """Algorithmic bytes per tuple (radix passes) or per row pair (probe) of the kernel that served `phase`: what it has to
    read and write given the tuple formats on either side of it.  The packed kernels of csrc/pack6.h move 6-byte tuples
    {payload, low 16 hash bits}: the last radix pass reads 8 and writes 6 bytes per tuple, the probe reads 6 (or 8) bytes per
    side and writes the 12-byte result row."""
    if "k_part_scatter_ring6" in kernel:
        return 8 + 6
    if "k_join_probe_direct6_pf" in kernel or "k_join_probe_direct6_sub" in kernel:
        return 6 + 6 + 12
    if "k_join_probe_direct6" in kernel:
        flags = kernel[kernel.index("<") + 1:kernel.rindex(">")].replace(" ", "").split(",")[-2:]
        return sum(6 if f == "true" else 8 for f in flags) + 12
    return default


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from dpu_olap_amd import capi
    from dpu_olap_amd import datagen as dg
    from dpu_olap_amd.dist_join import DeviceBatches, GpuLocalOps, distributed_join

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    # Rehearsal on a one-GPU box (tests/test_bench_multirank_gpu.py): OLAP_BENCH_SHARE_GPU=1 puts every rank on
    # cuda:0 and OLAP_BENCH_BACKEND=gloo carries the exchange (RCCL refuses two ranks on one device).  The
    # driver's runs use neither: one rank per GPU over RCCL.
    share_gpu = os.environ.get("OLAP_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("OLAP_BENCH_BACKEND", "nccl")
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    force_exchange = args.force_exchange or os.environ.get("OLAP_BENCH_FORCE_EXCHANGE") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    elif force_exchange:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)
    ctx = capi.Context(dev_index)
    info = ctx.device_info()
    ops = GpuLocalOps(ctx, device)
    ops.force_exchange = force_exchange
    if args.exchange_order:
        ops.exchange_order = args.exchange_order
    ctx.join_configure(args.radix_bits, args.max_bits_per_pass)
    ctx.set_tuning("scatter_variant", args.scatter_variant)
    for kv in args.tune:
        k, _, v = kv.partition("=")
        ctx.set_tuning(k, int(v))

    # ---- workload size
    free = info["free_mem"]
    sf = args.sf or 2048
    def need_bytes(sf_total):  # per rank: inputs + outputs + arena (3 tuple buffers) (+ exchange buffers at N > 1)
        n = sf_total * BATCH_ROWS // world
        single = 16 * n + 12 * n + 24 * n
        extra = (16 * n + 16 * n) * 1.15 if world > 1 else 0
        return (single + extra) * 1.03 + (2 << 30)
    while not args.sf and sf > world and need_bytes(sf) > free:
        sf //= 2
    if sf % world:
        raise SystemExit(f"SF={sf} must be a multiple of the number of GPUs ({world})")
    my_batches = sf // world
    n_local = my_batches * BATCH_ROWS
    n_total = sf * BATCH_ROWS
    first_batch = rank * my_batches            # contiguous slice of the batches per rank
    first_row = first_batch * BATCH_ROWS

    def col():
        return torch.empty(n_local, dtype=torch.int32, device=device)
    # the four input columns: one allocation, carved with a skew, so that the columns are not an exact power of two
    # apart (GpuLocalOps.empty_columns; OLAP_BENCH_COLUMN_SKEW=0 for separate back-to-back allocations, A/B)
    if os.environ.get("OLAP_BENCH_COLUMN_SKEW", "1") == "0":
        ops.column_skew_bytes = 0
    fk, y, pk, x = ops.empty_columns(n_local, 4)
    ctx.gen_fk_u32(fk.data_ptr(), n_local, dg.SEED, dg.STREAM_FK, first_row, BATCH_ROWS, BATCH_ROWS)
    ctx.gen_uniform_u32(y.data_ptr(), n_local, dg.SEED, dg.STREAM_Y, first_row)
    ctx.gen_index_u32(pk.data_ptr(), n_local, first_row & 0xFFFFFFFF)
    ctx.gen_uniform_u32(x.data_ptr(), n_local, dg.SEED, dg.STREAM_X, first_row)
    batches = lambda k, p: [(k[b * BATCH_ROWS:(b + 1) * BATCH_ROWS], p[b * BATCH_ROWS:(b + 1) * BATCH_ROWS]) for b in range(my_batches)]
    # the batches carry their pointer / length tables, like Arrow record batches hold their buffer addresses
    left, right = DeviceBatches(batches(fk, y)), DeviceBatches(batches(pk, x))

    # expected digest of the join result, from the inputs alone: pk is the global row
    # index, so x(fk) = gen32(STREAM_X, fk).  Order independent; summed over ranks.
    exp = col()
    ctx.gen_uniform_u32_at(exp.data_ptr(), fk.data_ptr(), n_local, dg.SEED, dg.STREAM_X)
    exp_sum, _ = ctx.checksum_rows3_u32(fk.data_ptr(), y.data_ptr(), exp.data_ptr(), n_local)
    del exp
    torch.cuda.empty_cache()

    def sync_all():
        torch.cuda.synchronize(device)
        ctx.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    host_wall = {}                 # N > 1: rank 0's wall time per stage of the exchange path (ms, summed over the timed steps)

    def step():
        ctx.set_profiling(True)    # Prepare(): per-phase HIP-event timers (inside the timed region, like the reference)
        return distributed_join(ops, left, right, timers=host_wall)

    result = None
    cold_first_step_ms = None      # the very first join of this context: the scratch arena (~217 GiB at SF=2048) is allocated in it
    for w in range(args.warmup):
        result = None
        tc = time.perf_counter()
        result = step()
        if w == 0:
            sync_all()
            cold_first_step_ms = (time.perf_counter() - tc) * 1e3
    sync_all()
    ctx.reset_timers()
    host_wall.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = None          # drop the previous step's output before allocating the next
        result = step()
    sync_all()
    dt = time.perf_counter() - t0
    if cold_first_step_ms is None and args.steps == 1:
        cold_first_step_ms = dt * 1e3          # no warm-up: the one timed step was the cold one
    timers = ctx.timers()
    if os.environ.get("OLAP_BENCH_DEBUG_ADDR") == "1":   # placement study: where the buffers of this process lie
        print(json.dumps({"addr": {"fk": hex(fk.data_ptr()), "y": hex(y.data_ptr()), "pk": hex(pk.data_ptr()), "x": hex(x.data_ptr()),
                                   "out": [hex(t.data_ptr()) for t in result[:3]], "arena_base": hex(ctx.get_counter("arena_base")),
                                   "arena_bytes": ctx.get_counter("arena_bytes")}}), file=sys.stderr, flush=True)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- verification (outside the timed region)
    ofk, oy, ox, jinfo = result
    got_sum, _ = ctx.checksum_rows3_u32(ofk.data_ptr(), oy.data_ptr(), ox.data_ptr(), ofk.numel())
    rows_out = ofk.numel()
    if world > 1:
        v = torch.tensor([got_sum & 0x7FFFFFFF, got_sum >> 31, exp_sum & 0x7FFFFFFF, exp_sum >> 31, rows_out], dtype=torch.int64, device=device)
        parts = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(parts, v)
        M = (1 << 64) - 1
        got_sum = sum(int(p[0]) + (int(p[1]) << 31) for p in parts) & M
        exp_sum = sum(int(p[2]) + (int(p[3]) << 31) for p in parts) & M
        rows_out = sum(int(p[4]) for p in parts)
    verified = bool(got_sum == exp_sum and rows_out == n_total)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = 2 * n_total / (dt / args.steps) / 1e6
        # Per-kernel figures from the live HIP-event timers of the timed region (every phase scope below brackets exactly
        # ONE kernel launch on the context's stream); the kernel names are what rocprofv3 --kernel-trace prints.
        alg_per_unit = {"partition-scatter": SCATTER_ALG_BYTES_PER_TUPLE, "partition-scatter-claim": SCATTER_ALG_BYTES_PER_TUPLE,
                        "partition-scatter-capacity": SCATTER_ALG_BYTES_PER_TUPLE, "partition-hist": 4, "join-probe": JOIN_ALG_BYTES_PER_PAIR}
        # HBM traffic per launch: PMC counters cannot be read from inside this process; the figures come from the committed
        # rocprofv3 PMC passes of this same command (scripts/profile_bench.sh -> profiles/), only for the config they were
        # taken on, else null.
        traffic_src, traffic_by_kernel = None, {}
        for cand in ("r3_traffic_sf2048.json", "r2_traffic_sf2048.json", "r1_traffic_sf2048.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", cand)))
            except Exception:
                continue
            if tj.get("sf") == sf and tj.get("n_gpus") == world and not args.radix_bits >= 0 and not args.tune:
                traffic_src, traffic_by_kernel = f"profiles/{cand}", tj.get("kernels", {})
                break
        kernels = []
        for ph, v in timers.items():
            if ph not in alg_per_unit or not v["launches"]:
                continue
            avg_ms = v["ms"] / v["launches"]
            name = ctx.phase_kernel(ph)
            alg = alg_bytes_per_unit(ph, name, alg_per_unit[ph]) * n_local
            ach = alg / (avg_ms * 1e-3) / 1e9
            kernels.append({"phase": ph, "kernel": name, "launches_per_step": v["launches"] / args.steps, "avg_launch_ms": round(avg_ms, 4),
                            "total_ms_per_step": round(v["ms"] / args.steps, 3), "algorithmic_bytes_per_launch": alg,
                            "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBPS, 4),
                            "traffic": (traffic_by_kernel.get(name) or {}).get("bytes_per_launch")})
        kernels.sort(key=lambda k: -k["total_ms_per_step"])
        roofline = None
        if kernels:
            d = kernels[0]   # the kernel with the largest total time per step
            roofline = {"bound": "hbm", "kernel": d["kernel"], "phase": d["phase"], "achieved": d["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": d["frac"], "traffic": d["traffic"],
                        "traffic_source": (traffic_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2)") if d["traffic"] else None,
                        "avg_launch_ms": d["avg_launch_ms"], "launches": int(round(d["launches_per_step"] * args.steps)),
                        "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                        # the same rate against what a pure streaming read reaches on this chip (k_sum / k_part_hist:
                        # 6.0-6.2 TB/s) -- the "HBM read roofline" of BASELINE.json's target
                        "measured_read_ceiling": MEASURED_READ_CEILING_GBPS,
                        "frac_of_read_ceiling": round(d["achieved"] / MEASURED_READ_CEILING_GBPS, 4)}
        # The whole join on MEASURED traffic (a k-pass radix join moves several times its compulsory bytes; the 60 %
        # target is read on the bytes the kernels actually move): sum over the join's kernels of PMC bytes per launch x launches
        # per join, over the wall time of a step.
        measured = None
        if traffic_by_kernel and all(k["traffic"] for k in kernels):
            per_join = sum(k["traffic"] * k["launches_per_step"] for k in kernels)
            ach = per_join / (dt / args.steps) / 1e9
            measured = {"bound": "hbm", "bytes_per_join": int(per_join), "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBPS, 4), "source": traffic_src,
                        "note": "sum over the join's kernels of (FETCH_SIZE x2 + WRITE_SIZE) per launch x launches per join / wall time per step"}
        whole = JOIN_ALG_BYTES_PER_PAIR * n_total / (dt / args.steps) / 1e9
        line = {
            "metric": "Mrows/s for BM_Join (build+probe) at SF=2048",
            "value": round(value, 1), "unit": "Mrows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"BM_Join u32 PK-FK SF={sf}: {sf} batches x 2Mi rows per side ({n_total} rows/side), "
                                   f"{world} x MI355X" + (", hash-owner split + all-to-all" if world > 1 else ""),
                       "sf": sf, "rows_per_side": n_total, "batch_rows": BATCH_ROWS, "radix": jinfo["stats"],
                       "exchange": {"path": jinfo.get("path", "columns" if world > 1 else None),
                                    "backend": jinfo.get("backend", backend if world > 1 else None),
                                    "order": jinfo.get("order"), "groups": jinfo.get("groups"), "left_slices": jinfo.get("left_slices"),
                                    "forced_at_world_1": bool(force_exchange and world == 1),
                                    "sent_rows_rank0": jinfo.get("sent_rows", 0)},
                       "device": info["name"].strip()},
            "verified": verified, "rows_out": rows_out,
            "roofline": roofline,
            "kernels": kernels,
            "roofline_measured_traffic": measured,
            "cold_first_step_ms": round(cold_first_step_ms, 1) if cold_first_step_ms is not None else None,
            "roofline_whole_join": {"bound": "hbm", "achieved": round(whole, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "frac": round(whole / HBM_PEAK_GBPS, 4),
                                    "note": "28 B per row-pair (SURVEY 8d) x rows/side / wall time per step, all GPUs"},
            "phase_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in timers.items()},
        }
        if host_wall:
            line["rank0_wall_ms_per_step"] = {k: round(v / args.steps, 3) for k, v in host_wall.items()}
        if world == 1 and not args.no_cpu_baseline:
            del left, right, result, ofk, oy, ox
            cb, arrow = cpu_baseline(os.cpu_count() or 1, args.cpu_sf)
            line["cpu_baseline"] = cb
            line["cpu_baseline_arrow"] = arrow
        print(json.dumps(line), flush=True)
        if not verified:
            print("VERIFICATION FAILED: output digest does not match the inputs", file=sys.stderr)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()
    ctx.close()
    return 0 if verified else 1


if __name__ == "__main__":
    sys.exit(main())
