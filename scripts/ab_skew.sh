#!/bin/bash
# Per-process variance of the kernels that touch caller-allocated columns, with and without the column skew.
SF=${1:-2048}; ROUNDS=${2:-3}
for r in $(seq $ROUNDS); do
  for v in 1 0; do
    OLAP_BENCH_COLUMN_SKEW=$v timeout -k 10 300 python bench.py --sf $SF --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('skew=$v', d['ms_per_step'], d['phase_ms_per_step'], d['verified'])" || echo "skew=$v FAILED"
  done
done
