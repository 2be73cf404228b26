#!/bin/bash
# rocprofv3 kernel trace (+ FETCH_SIZE / WRITE_SIZE passes) of the take benchmarks only; $1 = tag, $2 = --tune string
set -u
TAG=${1:-take}
TUNE=${2:-take_variant=0}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_take_trace_$TAG -- python3 bench_ops.py --only take --steps 2 --warmup 1 --tune $TUNE > $OUT/prof_take_trace_$TAG.log 2>&1 || { echo trace failed; tail -5 $OUT/prof_take_trace_$TAG.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_take_fetch_$TAG -- python3 bench_ops.py --only take --steps 1 --warmup 0 --tune $TUNE > $OUT/prof_take_fetch_$TAG.log 2>&1 || echo fetch pmc failed
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_take_write_$TAG -- python3 bench_ops.py --only take --steps 1 --warmup 0 --tune $TUNE > $OUT/prof_take_write_$TAG.log 2>&1 || echo write pmc failed
python3 scripts/summarize_rocprof.py --trace $OUT/prof_take_trace_$TAG --pmc $OUT/prof_take_fetch_$TAG --pmc $OUT/prof_take_write_$TAG \
  --out $OUT/${TAG}_take_sf2048.md --title "take at SF=2048 ($TUNE)" --cmd "rocprofv3 --kernel-trace --stats -- python3 bench_ops.py --only take --steps 2 --warmup 1 --tune $TUNE"
find $OUT -name "*.db" -delete 2>/dev/null
cat $OUT/${TAG}_take_sf2048.md
