#!/bin/bash
# Ablation timing.  Here:   tools/abl.sh build "<defs1>" "<defs2>" ...   builds one library per define set into
# build_variants/ (git-ignored, travels with gpurun).  On the GPU box:  tools/abl.sh run  times each variant.
set -u
if [ "$1" = build ]; then
  shift; rm -rf build_variants; mkdir -p build_variants; i=0
  for D in "$@"; do
    i=$((i+1)); touch starst3r_amd/csrc/*.hip
    ST3R_DEFS="$D" python -m starst3r_amd.build > /dev/null 2>&1 || echo "build failed: $D"
    cp starst3r_amd/libst3r_hip.so build_variants/v$i.so; echo "$D" > build_variants/v$i.txt
  done
  touch starst3r_amd/csrc/*.hip; python -m starst3r_amd.build > /dev/null 2>&1
else
  cp starst3r_amd/libst3r_hip.so /tmp/orig.so
  for f in build_variants/v*.so; do
    echo "== $(cat ${f%.so}.txt)"; cp $f starst3r_amd/libst3r_hip.so
    ST3R_BENCH_FREEZE=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('ms', round(d['ms_per_step'], 3), 'fwd', round(s['blend_fwd'], 3), 'bwd', round(s['blend_bwd'], 3), 'loss', round(s['loss'], 3), 'pbwd', round(s['project_bwd'], 3))"
  done
  cp /tmp/orig.so starst3r_amd/libst3r_hip.so
fi
