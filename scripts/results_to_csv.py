#!/usr/bin/env python3
"""Flatten Google-Benchmark JSON reports (what olap-query-host --benchmark_out writes) into
one CSV per report, so the reference's result tooling keeps working on the GPU numbers
(role of the reference's scripts/parse_results.py:8-39).

Every benchmark row becomes a CSV row with its JSON fields plus
  operator   the BM_* name (second component of "Fixture/BM_Op/Key:Val/...")
  <Key>      one column per Key:Val component of the name (Batches, Batch-Size, GPUs, ...)
Rows with error_occurred are skipped; NaN counters are written as 0.

usage: results_to_csv.py <dir with *.json> [more dirs...]"""
from __future__ import annotations

import csv
import json
import math
import pathlib
import sys


def name_fields(name: str) -> dict:
    parts = name.split("/")
    # "Fixture/BM_Op/..." for fixture benchmarks, "BM_Op/..." for plain functions
    out = {"operator": next((p for p in parts if p.startswith("BM_")), parts[1] if len(parts) > 1 else parts[0])}
    for part in parts:
        key, sep, val = part.partition(":")
        if sep:
            out[key] = val
    return out


def rows_of(report: dict):
    for bench in report.get("benchmarks", []):
        if bench.get("error_occurred"):
            continue
        row = dict(bench)
        row.update(name_fields(bench["name"]))
        yield {k: (0.0 if isinstance(v, float) and math.isnan(v) else v) for k, v in row.items()}


def convert(json_path: pathlib.Path) -> pathlib.Path | None:
    try:
        report = json.loads(json_path.read_text())
    except (OSError, ValueError) as err:
        print(f"skipped {json_path}: {err}", file=sys.stderr)
        return None
    rows = list(rows_of(report))
    columns: list[str] = []
    for row in rows:                      # stable column order: first appearance
        for key in row:
            if key not in columns:
                columns.append(key)
    csv_path = json_path.with_suffix(".csv")
    with csv_path.open("w", newline="") as fh:
        writer = csv.DictWriter(fh, fieldnames=columns)
        writer.writeheader()
        writer.writerows(rows)
    return csv_path


def main(argv) -> int:
    if len(argv) < 2:
        print(__doc__, file=sys.stderr)
        return 1
    for d in argv[1:]:
        for path in sorted(pathlib.Path(d).glob("*.json")):
            out = convert(path)
            if out:
                print(out)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
