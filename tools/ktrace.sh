#!/bin/bash
# quick per-kernel times of the SYNTH-1M train step (run on the GPU box from the repo root):
#   bash tools/ktrace.sh <tag> [extra bench args]   -> gpurun_out/kt_<tag>.md
set -u
TAG=${1:-x}; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
cd $ROOT
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/rocprof_summary.py $F --skip-first 2 > gpurun_out/kt_$TAG.md 2>&1
head -30 gpurun_out/kt_$TAG.md
