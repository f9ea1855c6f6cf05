#!/bin/bash
# per-kernel trace of library variants built by tools/abl.sh build:  bash tools/abl5.sh "<kernel regex>" [pytest files for the last variant]
PAT=${1:-k_blend}; shift
cp starst3r_amd/libst3r_hip.so /tmp/orig.so
for f in build_variants/v*.so; do
  n=$(basename ${f%.so}); echo "== $(cat ${f%.so}.txt)"; cp $f starst3r_amd/libst3r_hip.so
  timeout 600 bash tools/ktrace.sh $n > /dev/null 2>&1
  grep -E "$PAT" gpurun_out/kt_$n.md | cut -c1-120
  tail -1 gpurun_out/kt_$n/bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('it/s', round(d['value'], 2), {k: round(v, 3) for k, v in s.items()})"
done
if [ $# -gt 0 ]; then python -m pytest "$@" -x -q -m gpu 2>&1 | tail -3; fi
cp /tmp/orig.so starst3r_amd/libst3r_hip.so
