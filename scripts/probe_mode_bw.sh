#!/bin/bash
# On the GPU box: the packed probe's time on THIS box next to plain bandwidth figures of the same box -- the streaming read
# (k_sum over 16 GiB) and the filter (read + 25 % write) -- to see whether the probe's slow mode is the box's HBM rate.
set -u
J=$(timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print(d['ms_per_step'], p['partition-scatter-claim'], p['partition-scatter-capacity'], p['join-probe'])")
S=$(timeout -k 10 200 python bench_ops.py --only sum --steps 5 2>/dev/null | python3 -c "
import sys,json
print(' '.join(str(json.loads(l)['kernel_ms']) for l in sys.stdin if l.startswith('{')))")
F=$(timeout -k 10 200 python bench_ops.py --only filter --steps 3 2>/dev/null | python3 -c "
import sys,json
print(' '.join(str(json.loads(l)['kernel_ms']) for l in sys.stdin if l.startswith('{')))")
echo "join claim capacity probe: $J | sum kernels (u32 2Mi, u32 64Ki, i64): $S | filter kernels (u32, i32): $F"
# what the box says about itself (partition modes, clocks, power cap), for comparing slow and fast boxes
(rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -i "partition" | head -4; rocm-smi --showclocks 2>/dev/null | grep -i "mclk\|fclk\|sclk" | head -6; rocm-smi --showmaxpower --showpower 2>/dev/null | grep -i "power" | head -4; rocm-smi --showtemp 2>/dev/null | grep -i "hbm\|junction" | head -6; rocm-smi --showproductname 2>/dev/null | grep -i "vbios\|sku\|series" | head -4; uname -r; cat /sys/class/drm/card*/device/mem_info_vram_vendor 2>/dev/null | head -2) | tr '\n' ';' | cut -c1-1500
echo
