#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + two PMC passes of the bench
# command; raw output under gpurun_out/prof_*, summaries are written by
# scripts/summarize_rocprof.py (copy those into profiles/).
set -u
TAG=${1:-r1}
ARGS=${2:---steps 2 --warmup 1 --no-cpu-baseline}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace_$TAG -- python3 bench.py $ARGS > $OUT/prof_trace_$TAG.log 2>&1 || { echo trace failed; tail -5 $OUT/prof_trace_$TAG.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_$TAG -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_fetch_$TAG.log 2>&1 || { echo fetch pmc failed; tail -5 $OUT/prof_fetch_$TAG.log; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_$TAG -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_write_$TAG.log 2>&1 || { echo write pmc failed; tail -5 $OUT/prof_write_$TAG.log; }
python3 scripts/summarize_rocprof.py --trace $OUT/prof_trace_$TAG --pmc $OUT/prof_fetch_$TAG --pmc $OUT/prof_write_$TAG \
  --out $OUT/${TAG}_bench_join_sf2048.md --title "rocprofv3 --kernel-trace --stats + PMC (FETCH_SIZE, WRITE_SIZE in separate passes): bench.py $ARGS" --cmd "rocprofv3 --kernel-trace --stats -- python3 bench.py $ARGS"
# keep the raw CSVs small enough to merge back: drop everything but stats/trace csv
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT/prof_* 2>/dev/null
grep -h '"metric"' $OUT/prof_trace_$TAG.log | tail -1
