#!/usr/bin/env bash
# The device-resident variants of the reference's benchmarks (BM_*GpuResident: columns generated in HBM, operators driven
# through the C ABI on device pointers) at one scale factor, from the C++ harness: GB JSON + CSV under $OUT_DIR.
#   SF=2048 OUT_DIR=gpurun_out/cpp_resident scripts/run-gpu-resident.sh
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
BIN="$ROOT/dpu_olap_amd/host/build/olap-query-host"
SF="${SF:-2048}"
OUT_DIR="${OUT_DIR:-$ROOT/gpurun_out/cpp_resident_sf$SF}"
[ -x "$BIN" ] || make -C "$ROOT/dpu_olap_amd/host" >/dev/null || exit 1
mkdir -p "$OUT_DIR"
SF="$SF" "$BIN" --benchmark_filter="BM_JoinGpuResident" --benchmark_out="$OUT_DIR/join_resident_$SF.json" --benchmark_out_format=json || echo "join failed" >&2
SF="$SF" "$BIN" --benchmark_filter="BM_(Filter|Sum|Take)GpuResident" --benchmark_out="$OUT_DIR/ops_resident_$SF.json" --benchmark_out_format=json || echo "ops failed" >&2
python3 "$ROOT/scripts/results_to_csv.py" "$OUT_DIR"
cat "$OUT_DIR"/*.csv
