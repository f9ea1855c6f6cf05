#!/bin/bash
# Build-variant timing.  Here:  tools/variants.sh build "<defs1>" "<defs2>" ...  builds one library per define set
# into build_variants/ (git-ignored, travels with gpurun; "" = the production build).  On the GPU box:
# tools/variants.sh run [bench args]  times the driver's bench command with each variant (all stage times).
set -u
if [ "$1" = build ]; then
  shift; rm -rf build_variants; mkdir -p build_variants; i=0
  for D in "$@"; do
    i=$((i+1))
    ST3R_DEFS="$D" python -m starst3r_amd.build --force > /dev/null 2>&1 || echo "build failed: $D"
    cp starst3r_amd/libst3r_hip.so build_variants/v$i.so; echo "$D" > build_variants/v$i.txt
  done
  python -m starst3r_amd.build --force > /dev/null 2>&1
else
  shift
  cp starst3r_amd/libst3r_hip.so /tmp/orig.so
  for f in build_variants/v*.so; do
    cp $f starst3r_amd/libst3r_hip.so
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-drift --no-scaling-model --no-config1 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('== [$(cat ${f%.so}.txt)]', round(d['ms_per_step'], 3), 'ms', {k: round(v, 3) for k, v in s.items()})"
  done
  cp /tmp/orig.so starst3r_amd/libst3r_hip.so
fi
