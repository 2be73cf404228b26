for sf in 16 64 256 512 1024 2048; do
  timeout -k 10 300 python bench.py --sf $sf --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('SF', d['config']['sf'], d['ms_per_step'], 'ms', round(d['value']/1e3,1), 'Grows/s', d['config']['radix']['radix_bits'], d['phase_ms_per_step'], d['verified'])" || echo "SF $sf FAILED"
done
