#!/usr/bin/env bash
# Scale-factor sweep of the GPU operators and the Arrow-native baselines through the C++ host
# (role of the reference's scripts/run-upmem-2048.sh and run-cpp-baselines.sh): one
# Google-Benchmark JSON per (operator, SF) under $OUT_DIR, then CSVs next to them.
#   OUT_DIR=results/gpu SFS="1 2 4 8" REPEATS=1 OPS="Filter Join Sum Take Partition" scripts/run-gpu-sweep.sh
# Host-buffer path (PCIe inclusive); device-resident numbers come from bench.py / bench_ops.py.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
BIN="$ROOT/dpu_olap_amd/host/build/olap-query-host"
OUT_DIR="${OUT_DIR:-$ROOT/gpurun_out/results/gpu}"
SFS="${SFS:-1 2 4 8}"
REPEATS="${REPEATS:-1}"
OPS="${OPS:-Filter Join Sum Take Partition}"
NATIVE="${NATIVE:-0}"          # 1: also run the BM_*Native baselines (slow at large SF)
[ -x "$BIN" ] || make -C "$ROOT/dpu_olap_amd/host" >/dev/null || exit 1
mkdir -p "$OUT_DIR"
for op in $OPS; do
  lower="$(echo "$op" | tr '[:upper:]' '[:lower:]')"
  for sf in $SFS; do
    SF="$sf" "$BIN" --benchmark_filter="BM_${op}Gpu" --benchmark_out="$OUT_DIR/${lower}_gpu_${sf}.json" \
      --benchmark_out_format=json --benchmark_repetitions="$REPEATS" || echo "BM_${op}Gpu SF=$sf failed" >&2
    if [ "$NATIVE" = 1 ] && [ "$op" != Partition ]; then
      SF="$sf" "$BIN" --benchmark_filter="BM_${op}Native" --benchmark_out="$OUT_DIR/${lower}_native_${sf}.json" \
        --benchmark_out_format=json --benchmark_repetitions="$REPEATS" || echo "BM_${op}Native SF=$sf failed" >&2
    fi
  done
done
python3 "$ROOT/scripts/results_to_csv.py" "$OUT_DIR"
