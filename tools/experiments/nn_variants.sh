#!/bin/bash
# Ablation timing of the matching kernel.  Here:  tools/experiments/nn_variants.sh build "<defs1>" "<defs2>" ...
# builds one library per define set into build_variants/ (git-ignored, travels with gpurun).
# On the GPU box:  tools/experiments/nn_variants.sh run   times tools/time_nn.py with each.
set -u
if [ "$1" = build ]; then
  shift; rm -rf build_variants; mkdir -p build_variants; i=0
  for D in "$@"; do
    i=$((i+1)); touch starst3r_amd/csrc/recip_nn.hip
    ST3R_DEFS="$D" python -m starst3r_amd.build > /dev/null 2>&1 || echo "build failed: $D"
    cp starst3r_amd/libst3r_hip.so build_variants/v$i.so; echo "$D" > build_variants/v$i.txt
  done
  touch starst3r_amd/csrc/recip_nn.hip; python -m starst3r_amd.build > /dev/null 2>&1
else
  cp starst3r_amd/libst3r_hip.so /tmp/orig.so
  for f in build_variants/v*.so; do
    echo "== $(cat ${f%.so}.txt)"; cp $f starst3r_amd/libst3r_hip.so
    python tools/time_nn.py 2>&1 | grep "n=3072\|n=1024\|device-resident"
  done
  cp /tmp/orig.so starst3r_amd/libst3r_hip.so
fi
