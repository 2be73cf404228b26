#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel trace and/or PMC passes) per kernel.

  python scripts/summarize_rocprof.py --trace <dir> [--pmc <dir> ...] --out profiles/NAME.md

kernel trace: *_kernel_trace.csv (Kernel_Name, Start_Timestamp, End_Timestamp)
pmc passes : *_counter_collection.csv (Kernel_Name, Counter_Name, Counter_Value)
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts a
wide coalesced streaming read at exactly half its bytes (MI355X_MICROARCH.md, HBM), so the
table prints both the raw and the corrected (x2) read figure."""
import argparse
import collections
import csv
import glob
import json
import os
import re


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("olap::", "")
    return name.strip()[:60]


def read_csvs(d, pattern):
    rows = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(f, newline="") as fh:
            rows += list(csv.DictReader(fh))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--pmc", action="append", default=[])
    ap.add_argument("--out", required=True)
    ap.add_argument("--title", default="rocprofv3 summary")
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    dur = collections.defaultdict(list)
    if a.trace:
        for r in read_csvs(a.trace, "*kernel_trace.csv"):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in a.pmc:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in read_csvs(d, "*counter_collection.csv"):
            key = (r.get("Dispatch_Id") or r.get("Dispatch_ID"), r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[key] = short(r["Kernel_Name"])
        for (disp, cname), v in per_dispatch.items():
            ctr[names[(disp, cname)]][cname].append(v)
    kernels = sorted(set(dur) | set(ctr), key=lambda k: -sum(dur.get(k, [0])))
    total = sum(sum(v) for v in dur.values()) or 1.0
    lines = [f"# {a.title}", "", f"command: `{a.cmd}`" if a.cmd else "", "",
             "| kernel | calls | total ms | avg ms | % | FETCH_SIZE KiB/launch (raw) | read GB/launch (x2 corrected) | WRITE_SIZE KiB/launch | write GB/launch |",
             "|---|---|---|---|---|---|---|---|---|"]
    summary = {}
    for k in kernels:
        d = dur.get(k, [])
        f = ctr.get(k, {}).get("FETCH_SIZE", [])
        w = ctr.get(k, {}).get("WRITE_SIZE", [])
        favg = sum(f) / len(f) if f else None
        wavg = sum(w) / len(w) if w else None
        row = {
            "calls": len(d), "total_ms": round(sum(d), 3), "avg_ms": round(sum(d) / len(d), 4) if d else None,
            "fetch_kib_raw": favg, "read_gb_corrected": round(favg * 1024 * 2 / 1e9, 3) if favg is not None else None,
            "write_kib": wavg, "write_gb": round(wavg * 1024 / 1e9, 3) if wavg is not None else None,
        }
        summary[k] = row
        lines.append("| {} | {} | {} | {} | {:.1f} | {} | {} | {} | {} |".format(
            k, row["calls"], row["total_ms"], row["avg_ms"], 100 * sum(d) / total,
            "" if favg is None else f"{favg:.0f}", "" if favg is None else row["read_gb_corrected"],
            "" if wavg is None else f"{wavg:.0f}", "" if wavg is None else row["write_gb"]))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(re.sub(r"\.md$", ".json", a.out), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
