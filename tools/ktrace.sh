#!/bin/bash
# rocprofv3 kernel trace of a short bench run, summarised per kernel (GPU box):  bash tools/ktrace.sh <tag> [bench args]
TAG=${1:-k}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kt_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-drift --train-only "$@" > $OUT/trace.log 2>&1
cd $ROOT
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/rocprof_summary.py $F > $OUT/summary.md
head -32 $OUT/summary.md
find $OUT/trace -name "*.csv" ! -name "*kernel_stats*" -delete; find $OUT/trace -name "*.db" -delete
