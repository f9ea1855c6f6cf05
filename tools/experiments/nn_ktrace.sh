#!/bin/bash
# rocprofv3 kernel trace of tools/time_nn.py, summarised per kernel (GPU box):  bash tools/experiments/nn_ktrace.sh <tag>
TAG=${1:-nn}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kt_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- python $ROOT/tools/time_nn.py > $OUT/trace.log 2>&1
cd $ROOT
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/rocprof_summary.py $F > $OUT/summary.md
grep -i "k_nn\|kernel" $OUT/summary.md | head -8
python - "$F" <<'PY'
import csv, sys
rows = [(r["Kernel_Name"].split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(sys.argv[1]))]
for name in ("k_nn_argmax", "k_nn_reduce"):
    d = [t for n, t in rows if n == name]
    print(name, "first 24 launches (us):", [round(t, 1) for t in d[:24]])
PY
find $OUT/trace -name "*.csv" ! -name "*kernel_stats*" -delete; find $OUT/trace -name "*.db" -delete
