#!/bin/bash
# A/B of the ring scatter against the write-combining kernels on ONE box, interleaved:
#   scripts/ab_ring.sh [SF]     (default 2048)
SF=${1:-2048}
for r in 1 2; do
  for v in "ring_scatter=0" "ring_scatter=3" "ring_scatter=1" "ring_scatter=2"; do
    timeout -k 10 300 python bench.py --sf $SF --steps 3 --warmup 1 --no-cpu-baseline --tune $v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['phase_ms_per_step'], d['verified'])" || echo "$v FAILED"
  done
done
