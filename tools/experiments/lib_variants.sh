#!/bin/bash
# Whole-library ablations.  Here:   tools/experiments/lib_variants.sh build "<defs1>" "<defs2>" ...
# builds one library per define set into build_variants/ (git-ignored, travels with gpurun) and rebuilds the plain one.
# On the GPU box:  tools/experiments/lib_variants.sh run <flags>   times the frozen bench scene with each, under
# ST3R_DEBUG_FLAGS=<flags>.
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
  cp starst3r_amd/libst3r_hip.so /tmp/orig.so
  for f in build_variants/v*.so; do
    echo "== $(cat ${f%.so}.txt)"; cp $f starst3r_amd/libst3r_hip.so
    bash tools/experiments/ab_flags.sh "$2" | tail -1
  done
  cp /tmp/orig.so starst3r_amd/libst3r_hip.so
fi
