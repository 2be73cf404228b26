#!/bin/bash
# Is the probe's slow mode (20.6 vs 23.8 ms at SF=2048) a clock / power / temperature state of the box?
# Runs bench.py N times back to back and logs rocm-smi readings before each run.
N=${1:-6}
for i in $(seq $N); do
  echo "--- run $i  $(date +%T)"
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory|edge)" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
  echo
  OLAP_BENCH_DEBUG_ADDR=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/.addr.$i | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phase_ms_per_step'])" || echo FAILED
  grep '"addr"' gpurun_out/.addr.$i
done
