#!/usr/bin/env bash
# Device-count sweep through the C++ host (role of the reference's scripts/run-upmem-scale.sh:15-38, which
# sweeps NR_DPUS = 1 .. 2048 at SF = 2048): every *Gpu operator at a fixed SF over NR_GPUS = 1, 2, 4, 8, one
# Google-Benchmark JSON per (operator, #GPUs) under $OUT_DIR, then CSVs.  A count the host cannot provide is
# skipped with the harness' error line -- or, with OLAP_GROUP_LOOPBACK=1, run as logical ranks of device 0 and
# labelled so (functional rehearsal on a one-GPU box; the timings then say nothing about xGMI).
#   OUT_DIR=results/scale SF=64 GPUS="1 2 4 8" REPEATS=1 OPS="Filter Join Sum Take" scripts/run-gpu-scale.sh
# Host-buffer path (PCIe inclusive); the device-resident strong-scaling figure is bench.py --gpus N.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
BIN="$ROOT/dpu_olap_amd/host/build/olap-query-host"
OUT_DIR="${OUT_DIR:-$ROOT/gpurun_out/results/scale}"
SF="${SF:-64}"
GPUS="${GPUS:-1 2 4 8}"
REPEATS="${REPEATS:-1}"
OPS="${OPS:-Filter Join Sum Take}"
[ -x "$BIN" ] || make -C "$ROOT/dpu_olap_amd/host" >/dev/null || exit 1
mkdir -p "$OUT_DIR"
for op in $OPS; do
  lower="$(echo "$op" | tr '[:upper:]' '[:lower:]')"
  for n in $GPUS; do
    SF="$SF" NR_GPUS="$n" "$BIN" --benchmark_filter="BM_${op}Gpu/" --benchmark_out="$OUT_DIR/${lower}_gpu_${n}.json" \
      --benchmark_out_format=json --benchmark_repetitions="$REPEATS" || echo "BM_${op}Gpu NR_GPUS=$n failed" >&2
  done
done
python3 "$ROOT/scripts/results_to_csv.py" "$OUT_DIR"
