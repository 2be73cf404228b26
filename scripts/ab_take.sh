set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "take" > gpurun_out/r2_take_tests.log 2>&1 || { tail -30 gpurun_out/r2_take_tests.log; exit 1; }
tail -3 gpurun_out/r2_take_tests.log
for v in 5 0 5 0; do
  timeout -k 10 300 python bench_ops.py --sf 2048 --only take --tune take_variant=$v >> gpurun_out/r2_take_ab.log 2>&1
done
grep -o '"metric": "[^"]*"\|"kernel_ms": [0-9.]*\|"verified": [a-z]*' gpurun_out/r2_take_ab.log | paste - - - 
