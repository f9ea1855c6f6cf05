#!/bin/bash
# PMC counters of one kernel of the train step (kernel-trace only, one counter group per pass).
# usage (GPU box, repo root): bash tools/pmc_kernel.sh <kernel-substring> <tag>
set -u
KERN=${1:-k_blend_bwd}
TAG=${2:-x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmck_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline"
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G -f csv -d $OUT/g$i -- $CMD > $OUT/g$i.log 2>&1
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
    for k in acc: print(f"{k:28s} {acc[k] / n[k]:16.0f}  (avg of {n[k]} launches)")
PY
