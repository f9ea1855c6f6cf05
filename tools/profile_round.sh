#!/bin/bash
# Profile passes of a round (run on the GPU box from the repo root):  bash tools/profile_round.sh <tag>
#   1. the driver's bench command, un-profiled                      -> gpurun_out/prof_<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats around the SAME command     -> .../trace
#   3. PMC passes (each in its own run, --kernel-trace only): FETCH_SIZE | WRITE_SIZE | SQ instruction counts | SQ cycles
# gpurun MERGES gpurun_out/ back into the local copy: delete the local gpurun_out/prof_<tag> before repeating a tag, or
# tools/make_profile_docs.py averages over the counter files of every earlier run as well.
set -u
TAG=${1:-r4a}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-drift --no-scaling-model --no-config1"
$CMD > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
SHORT="python $ROOT/bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-drift --no-scaling-model --no-config1"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/pmc_$C -- $SHORT > $OUT/pmc_$C.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -f csv -d $OUT/pmc_SQ1 -- $SHORT > $OUT/pmc_SQ1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -f csv -d $OUT/pmc_SQ2 -- $SHORT > $OUT/pmc_SQ2.log 2>&1
cd $ROOT
tail -c 400 $OUT/bench.json
