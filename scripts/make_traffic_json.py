#!/usr/bin/env python3
"""profiles/r1_traffic_sf2048.json from a summarize_rocprof.py JSON: HBM bytes per launch of
the join kernels (PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes), grouped by the phase
names bench.py's live timers use.  usage: make_traffic_json.py <summary.json> <out.json>"""
import json
import sys

PHASE = {"k_part_scatter": "partition-scatter", "k_join_probe": "join-probe", "k_part_hist": "partition-hist"}
src, dst = sys.argv[1], sys.argv[2]
summary = json.load(open(src))
kernels, acc = {}, {}
for name, k in summary.items():
    phase = next((p for pre, p in PHASE.items() if name.startswith(pre)), None)
    if phase is None or "read_gb_corrected" not in k:
        continue
    rd, wr = int(k["fetch_kib_raw"] * 1024 * 2), int(k["write_kib"] * 1024)
    kernels[name] = {"read_bytes": rd, "write_bytes": wr, "avg_ms": k["avg_ms"], "calls": k["calls"]}
    a = acc.setdefault(phase, [0, 0])
    a[0] += (rd + wr) * k["calls"]
    a[1] += k["calls"]
out = {
    "_comment": "HBM bytes per launch from rocprofv3 PMC passes of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` at "
                "SF=2048 on one MI355X (profiles/r1_bench_join_sf2048_rocprof.md): FETCH_SIZE and WRITE_SIZE collected in separate "
                "passes, KiB -> bytes, FETCH_SIZE doubled (gfx950 counts wide coalesced reads at half their bytes; calibrated on "
                "pass 0, which must read exactly 2^32 x 8 B = 34.36 GB and reports 17.18 GB raw).  phases = launch-weighted mean "
                "over the kernels of a phase.",
    "sf": 2048, "n_gpus": 1, "kernels": kernels,
    "phases": {p: a[0] // a[1] for p, a in acc.items()},
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["phases"]))
