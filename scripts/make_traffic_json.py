#!/usr/bin/env python3
"""profiles/rN_traffic_sf2048.json from a summarize_rocprof.py JSON: HBM bytes per launch of the join
kernels (PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes), keyed by the kernel name that
olap_phase_kernel() / rocprofv3 report -- bench.py joins its live per-kernel timers with this file.
usage: make_traffic_json.py <summary.json> <out.json> [source-md]"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
source_md = sys.argv[3] if len(sys.argv) > 3 else src.replace(".json", ".md")
summary = json.load(open(src))
kernels = {}
for name, k in summary.items():
    if not name.startswith(("k_part_", "k_join_")) or k.get("fetch_kib_raw") is None or k.get("write_kib") is None:
        continue
    rd, wr = int(k["fetch_kib_raw"] * 1024 * 2), int(k["write_kib"] * 1024)
    kernels["olap::" + name] = {"read_bytes": rd, "write_bytes": wr, "bytes_per_launch": rd + wr, "avg_ms": k["avg_ms"], "calls": k["calls"]}
out = {
    "_comment": "HBM bytes per launch from rocprofv3 PMC passes of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` at "
                f"SF=2048 on one MI355X ({source_md}): FETCH_SIZE and WRITE_SIZE collected in separate passes, KiB -> bytes, "
                "FETCH_SIZE doubled (gfx950 counts wide coalesced reads at half their bytes; calibrated on pass 0, which must read "
                "exactly 2^32 x 8 B = 34.36 GB and reports 17.18 GB raw).  avg_ms / calls are those of the kernel-trace pass.",
    "sf": 2048, "n_gpus": 1, "kernels": kernels,
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v["bytes_per_launch"] for k, v in kernels.items()}))
