#!/bin/bash
# build_variants/ from ST3R_DEFS sets of the WORKING TREE:  tools/experiments/build_defs_variants.sh "" "-DFOO=1" ...
set -u
rm -rf build_variants; mkdir -p build_variants; i=0
for D in "$@"; do
  i=$((i+1))
  ST3R_DEFS="$D" python -m starst3r_amd.build --force > /dev/null 2>&1 || echo "build failed: $D"
  cp starst3r_amd/libst3r_hip.so build_variants/v$i.so; echo "[$D]" > build_variants/v$i.txt
done
python -m starst3r_amd.build --force > /dev/null 2>&1
