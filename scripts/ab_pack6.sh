#!/bin/bash
# A/B on one box: packed (6-byte) tuples between the last pass and the probe vs 8-byte tuples; knob sets as arguments.
# usage: scripts/ab_pack6.sh ROUNDS "knob=v knob=v" "..." ...
N=${1:-2}; shift
for i in $(seq $N); do
  for cfg in "$@"; do
    args=""
    for kv in $cfg; do args="$args --tune $kv"; done
    echo -n "[$cfg] "
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phase_ms_per_step'], d['verified'])" || echo FAILED
  done
done
