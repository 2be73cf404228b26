#!/bin/bash
# Interleaved A/B of two builds of libolap_gpu.so on ONE device (same process model, same
# box): ab_libs/old.so vs ab_libs/new.so, 3 rounds each.
for r in 1 2 3; do
  for v in old new; do
    OLAP_GPU_LIB=$PWD/ab_libs/$v.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['phase_ms_per_step'], d['verified'])"
  done
done
