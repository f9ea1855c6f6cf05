#!/bin/bash
# L2 / memory-side PMC counters of one kernel of the train step.  usage (GPU box): bash tools/pmc_mem.sh <kernel> <tag>
set -u
KERN=${1:-k_blend_bwd}; TAG=${2:-x}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcm_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline"
i=0
for G in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G -f csv -d $OUT/g$i -- $CMD > $OUT/g$i.log 2>&1 || tail -3 $OUT/g$i.log
done
cd $ROOT
python - "$OUT" "$KERN" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    for k in acc: print(f"{k:32s} {acc[k] / n[k]:16.0f}  (avg of {n[k]} launches)")
PY
