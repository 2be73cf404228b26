#!/bin/bash
# PMC view of the join kernels at SF=2048 (separate rocprofv3 --pmc passes, no trace domains mixed in):
# SQ occupancy / waits / LDS, instruction mix, address translation (UTCL1), L2 -> memory stalls.
set -u
OUT=gpurun_out
TAG=${1:-r2}
ONLY=${PMC_ONLY:-}          # e.g. PMC_ONLY="3" runs only the third counter set
BENCH_ARGS=${BENCH_ARGS:-}  # e.g. BENCH_ARGS="--tune claim_weave=0"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCC_TAG_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  if [ -n "$ONLY" ] && [[ " $ONLY " != *" $i "* ]]; then continue; fi
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_${TAG}_$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > $OUT/pmc_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pmc_${TAG}_$i.log; }
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("$OUT/pmc_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("olap::", "")[:60]
        if not k.startswith(("k_part", "k_join")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r.get("Dispatch_Id"))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()): print("   %-44s %.4g per launch" % (c, v / max(len(n[k][c]), 1)))
PY
find $OUT -name "*.db" -delete 2>/dev/null
