#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of
# bench_ops.py (filter / sum / take at SF=2048); summary -> gpurun_out/<tag>_bench_ops_sf2048.md
set -u
TAG=${1:-r1}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ops_trace_$TAG -- python3 bench_ops.py --steps 2 --warmup 1 > $OUT/prof_ops_trace_$TAG.log 2>&1 || { echo trace failed; tail -5 $OUT/prof_ops_trace_$TAG.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_ops_fetch_$TAG -- python3 bench_ops.py --steps 1 --warmup 0 > $OUT/prof_ops_fetch_$TAG.log 2>&1 || echo fetch pmc failed
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_ops_write_$TAG -- python3 bench_ops.py --steps 1 --warmup 0 > $OUT/prof_ops_write_$TAG.log 2>&1 || echo write pmc failed
python3 scripts/summarize_rocprof.py --trace $OUT/prof_ops_trace_$TAG --pmc $OUT/prof_ops_fetch_$TAG --pmc $OUT/prof_ops_write_$TAG \
  --out $OUT/${TAG}_bench_ops_sf2048.md --title "rocprofv3 --kernel-trace --stats + PMC (FETCH_SIZE, WRITE_SIZE in separate passes): bench_ops.py --steps 2 --warmup 1 (filter / sum / take at SF=2048)" --cmd "rocprofv3 --kernel-trace --stats -- python3 bench_ops.py --steps 2 --warmup 1"
find $OUT -name "*.db" -delete 2>/dev/null
grep -h '"metric"\|"name"' $OUT/prof_ops_trace_$TAG.log | cut -c1-250
