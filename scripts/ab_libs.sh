#!/bin/bash
# Interleaved A/B of library builds on ONE box: scripts/ab_libs.sh "base nt_load ..." [SF] [ROUNDS]
LIBS=$1; SF=${2:-2048}; ROUNDS=${3:-2}
for r in $(seq $ROUNDS); do
  for v in $LIBS; do
    OLAP_GPU_LIB=$PWD/ab_libs/$v.so timeout -k 10 300 python bench.py --sf $SF --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['phase_ms_per_step'], d['verified'])" || echo "$v FAILED"
  done
done
