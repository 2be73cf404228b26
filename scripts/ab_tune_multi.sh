#!/bin/bash
# Interleaved A/B of tuning settings on ONE box: scripts/ab_tune_multi.sh SF ROUNDS "k=v k2=v2" "k=v" ...   ("" = defaults)
SF=$1; ROUNDS=$2; shift 2
for r in $(seq $ROUNDS); do
  for cfg in "$@"; do
    T=""; for kv in $cfg; do T="$T --tune $kv"; done
    timeout -k 10 300 python bench.py --sf $SF --steps 3 --warmup 1 --no-cpu-baseline $T 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg]', d['ms_per_step'], d['phase_ms_per_step'], d['verified'])" || echo "[$cfg] FAILED"
  done
done
