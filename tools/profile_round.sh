#!/bin/bash
# rocprofv3 passes for profiles/: kernel trace + stats, then PMC counters in separate runs (kernel-trace only).
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh <tag>
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline"
# the kernel trace runs the default step counts so that its kernel means and the bench line's stage_ms describe the same run
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- python $ROOT/bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/pmc_$C -- $CMD > $OUT/pmc_$C.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -f csv -d $OUT/pmc_SQ1 -- $CMD > $OUT/pmc_SQ1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -f csv -d $OUT/pmc_SQ2 -- $CMD > $OUT/pmc_SQ2.log 2>&1
cd $ROOT
find $OUT -name "*.csv" | head -30
tail -2 $OUT/trace.log
