#!/bin/bash
# PMC pass over the join kernels at a reduced SF (same 10+10-bit plan as SF=2048).
set -u
TAG=${1:-pmc}
SF=${2:-256}
EXTRA=${3:-}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $OUT/pmc1_$TAG -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --sf $SF $EXTRA > $OUT/pmc1_$TAG.log 2>&1 || tail -3 $OUT/pmc1_$TAG.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM \
  --output-format csv -d $OUT/pmc2_$TAG -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --sf $SF $EXTRA > $OUT/pmc2_$TAG.log 2>&1 || tail -3 $OUT/pmc2_$TAG.log
python3 - <<PY
import csv, glob, collections, re
for d in ("$OUT/pmc1_$TAG", "$OUT/pmc2_$TAG"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("olap::", "")[:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r.get("Dispatch_Id"))
    for k in acc:
        if not k.startswith(("k_part", "k_join")): continue
        print(k, "launches", len(n[k]))
        for c, v in sorted(acc[k].items()): print("   %-24s %.4g per launch" % (c, v / max(len(n[k]), 1)))
PY
