#!/bin/bash
# L2 (TCC) view of the join kernels at SF=2048: request counts and granularity towards HBM, hits / misses, stalls.
# Separate rocprofv3 --pmc passes (a pass holds only a few TCC counters).
set -u
OUT=gpurun_out
export TMPDIR=/tmp
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum" "TCC_TAG_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_l2_$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_l2_$i.log 2>&1 || tail -3 $OUT/pmc_l2_$i.log
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("$OUT/pmc_l2_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("olap::", "")[:44]
        if not k.startswith(("k_part", "k_join")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r.get("Dispatch_Id"))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()): print("   %-40s %.4g per launch" % (c, v / max(len(n[k][c]), 1)))
PY
