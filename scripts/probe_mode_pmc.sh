#!/bin/bash
# On the GPU box: checks which mode the packed probe runs in on THIS box (bench.py, 3 steps) and takes the address-translation /
# L2 PMC passes of scripts/pmc_ring.sh either way, tagged fast / slow -- to compare boxes (profiles/r3_probe_mode_pmc.txt).
set -u
mkdir -p gpurun_out
P=$(timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['phase_ms_per_step']['join-probe'])")
MODE=fast; python3 -c "import sys; sys.exit(0 if float('$P') > 19.5 else 1)" && MODE=slow
echo "probe $P ms -> $MODE"
PMC_ONLY="${PMC_SETS:-3 4 6}" bash scripts/pmc_ring.sh r3mode_$MODE 2>&1 | grep -A 12 "k_join_probe_direct6"
