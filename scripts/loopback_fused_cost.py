#!/usr/bin/env python3
"""Per-rank device cost of the FUSED multi-GPU join, measured on ONE GPU (no exchange).
For G in {8,4,2} the whole SF=2048 problem is generated; every logical sender splits its
share into G x S = 512 buckets (olap_partition_tuples_u32), logical rank 0 collects its S
runs from every sender (the exchange, emulated by slicing) and runs
olap_join_presplit_u32 on the real received size.  Reports one sender's split time and rank
0's join time; the xGMI exchange must be added to project a G-GPU step.  Projection aid."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpu_olap_amd import capi, datagen as dg
from dpu_olap_amd.dist_join import GpuLocalOps, _chunk_tables

ROWS = 2 << 20
NB = int(os.environ.get("NB", "2048"))
ctx = capi.Context(0)
for kv in os.environ.get("OLAP_TUNE", "").split(","):   # e.g. OLAP_TUNE=presplit_sub=0,presplit_refine=1
    if "=" in kv:
        ctx.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
GS = tuple(int(g) for g in os.environ.get("GS", "8,4,2").split(","))
dev = torch.device("cuda", 0)
ops = GpuLocalOps(ctx, dev)
n = NB * ROWS
col = lambda: torch.empty(n, dtype=torch.int32, device=dev)
fk, y, pk, x, exp = col(), col(), col(), col(), col()
ctx.gen_fk_u32(fk.data_ptr(), n, dg.SEED, dg.STREAM_FK, 0, ROWS, ROWS)
ctx.gen_uniform_u32(y.data_ptr(), n, dg.SEED, dg.STREAM_Y, 0)
ctx.gen_index_u32(pk.data_ptr(), n, 0)
ctx.gen_uniform_u32(x.data_ptr(), n, dg.SEED, dg.STREAM_X, 0)
for G in GS:
    S, per, bits = 512 // G, NB // G, G.bit_length() - 1
    parts = {"L": [], "R": []}
    cnts = {"L": [], "R": []}
    split_ms, split_ph = [], None
    for q in range(G):
        sl = lambda t: [t[(q * per + b) * ROWS:(q * per + b + 1) * ROWS] for b in range(per)]
        ctx.set_profiling(True); ctx.reset_timers()
        t0 = time.perf_counter()
        lt, loff = ops.partition_tuples(sl(fk), sl(y), G * S)
        rt, roff = ops.partition_tuples(sl(pk), sl(x), G * S)
        ctx.synchronize()
        split_ms.append((time.perf_counter() - t0) * 1e3)
        split_ph = {k: round(v["ms"], 2) for k, v in ctx.timers().items()}
        parts["L"].append(lt[loff[0]:loff[S]].clone()); cnts["L"].append([loff[j + 1] - loff[j] for j in range(S)])
        parts["R"].append(rt[roff[0]:roff[S]].clone()); cnts["R"].append([roff[j + 1] - roff[j] for j in range(S)])
        del lt, rt
    RL, RR = torch.cat(parts["L"]), torch.cat(parts["R"])
    del parts
    l_lo, l_hi = _chunk_tables(torch.tensor(cnts["L"], dtype=torch.int64))
    r_lo, r_hi = _chunk_tables(torch.tensor(cnts["R"], dtype=torch.int64))
    best = None
    for rep in range(3):
        ctx.reset_timers()
        t0 = time.perf_counter()
        ofk, oy, ox, st = ops.join_presplit(RL, l_lo, l_hi, RR, r_lo, r_hi, S, G, bits)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        if best is None or ms < best[0]:
            best = (ms, {k: round(v["ms"], 2) for k, v in ctx.timers().items()})
    # rank 0's shard must be exactly the rows whose key it owns: digest against the generator
    ctx.gen_uniform_u32_at(exp.data_ptr(), ofk.data_ptr(), ofk.numel(), dg.SEED, dg.STREAM_X)
    ok = bool(torch.equal(ox, exp[:ofk.numel()])) and st["missing_keys"] == 0
    print(json.dumps({"G": G, "S": S, "rows_sent_per_rank_per_side": n // G, "recv_left": RL.numel(), "recv_right": RR.numel(),
                      "split_ms_per_rank": round(min(split_ms), 2), "split_phases": split_ph,
                      "join_ms": round(best[0], 2), "join_phases": best[1], "radix": st["radix_bits"],
                      "partitions": st["nr_partitions"], "table_slots": st["table_slots"], "x_matches_generator": ok}), flush=True)
    del RL, RR, ofk, oy, ox
    torch.cuda.empty_cache()
    # the same rank's work when the tables travel as K = 4 sub-bucket groups (one join per group)
    K, gb = 4, 2
    lb = (S // K).bit_length() - 1
    Lw = 1 << lb
    recvL = [[] for _ in range(K)]; recvR = [[] for _ in range(K)]
    cntL = [[] for _ in range(K)]; cntR = [[] for _ in range(K)]
    split_ms = []
    for q in range(G):
        sl = lambda t: [t[(q * per + b) * ROWS:(q * per + b + 1) * ROWS] for b in range(per)]
        t0 = time.perf_counter()
        lt, loff = ops.split_tuples(sl(fk), sl(y), bits, gb, lb)
        rt, roff = ops.split_tuples(sl(pk), sl(x), bits, gb, lb)
        ctx.synchronize()
        split_ms.append((time.perf_counter() - t0) * 1e3)
        for g in range(K):                       # owner 0's runs of group g: buckets (g, 0, 0..Lw)
            b0 = (g * G + 0) * Lw
            recvL[g].append(lt[loff[b0]:loff[b0 + Lw]].clone()); cntL[g].append([loff[b0 + l + 1] - loff[b0 + l] for l in range(Lw)])
            recvR[g].append(rt[roff[b0]:roff[b0 + Lw]].clone()); cntR[g].append([roff[b0 + l + 1] - roff[b0 + l] for l in range(Lw)])
        del lt, rt
    tot_ms, rows, ok = 0.0, 0, True
    ph = {}
    for g in range(K):
        RLg, RRg = torch.cat(recvL[g]), torch.cat(recvR[g])
        l_lo, l_hi = _chunk_tables(torch.tensor(cntL[g], dtype=torch.int64))
        r_lo, r_hi = _chunk_tables(torch.tensor(cntR[g], dtype=torch.int64))
        best = None
        for rep in range(2):
            ctx.reset_timers()
            t0 = time.perf_counter()
            ofk, oy, ox, st = ops.join_presplit(RLg, l_lo, l_hi, RRg, r_lo, r_hi, Lw, G, bits + gb)
            ctx.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            if best is None or ms < best[0]:
                best = (ms, {k: v["ms"] for k, v in ctx.timers().items()})
        tot_ms += best[0]
        for k, v in best[1].items():
            ph[k] = round(ph.get(k, 0.0) + v, 2)
        rows += ofk.numel()
        ctx.gen_uniform_u32_at(exp.data_ptr(), ofk.data_ptr(), ofk.numel(), dg.SEED, dg.STREAM_X)
        ok = ok and bool(torch.equal(ox, exp[:ofk.numel()])) and st["missing_keys"] == 0
        del RLg, RRg, ofk, oy, ox
    print(json.dumps({"G": G, "groups": K, "segments_per_group": Lw, "split_ms_per_rank": round(min(split_ms), 2),
                      "group_joins_total_ms": round(tot_ms, 2), "group_join_phases_total": ph, "rows": rows, "x_matches_generator": ok}), flush=True)
    del recvL, recvR
    torch.cuda.empty_cache()
