#!/bin/bash
# MFMA utilisation of the matching kernel from hardware counters (GPU box):  bash tools/experiments/nn_pmc.sh <tag>
# Separate PMC passes, --kernel-trace only (the guide's recipe): SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES | SQ_INSTS_MFMA + SQ_INSTS_VALU.
TAG=${1:-nn}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/nnpmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/p1 -- python $ROOT/tools/time_nn.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 -f csv -d $OUT/p2 -- python $ROOT/tools/time_nn.py > $OUT/p2.log 2>&1
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"{out}/{p}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("k_nn_argmax"):
                acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
    for name, d in acc.items():
        vals = sorted(sum(v) for v in d.values())
        print(p, name, "launches", len(vals), "largest five (the n = 3072 calls):", [f"{v:.4g}" for v in vals[-5:]])
PY
