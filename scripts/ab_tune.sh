#!/bin/bash
# Interleaved A/B of one tuning knob on ONE device: scripts/ab_tune.sh claim_mode 0 1
# runs bench.py three times per value, alternating (boxes differ by ~7 %, runs on one box by ~1 %).
KEY=$1; A=$2; B=$3
for r in 1 2 3; do
  for v in $A $B; do
    timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --tune $KEY=$v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$KEY=$v', d['ms_per_step'], d['phase_ms_per_step'], d['verified'])"
  done
done
