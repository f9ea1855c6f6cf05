#!/bin/bash
# Register / LDS / occupancy report of the kernels of one csrc file, compiled with the flags of starst3r_amd/build.py
# usage: tools/kres.sh gs_blend.hip [kernel-name-substring] [extra -D flags]
F=$1; K=${2:-}; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNDEBUG -Xclang -target-feature -Xclang -packed-fp32-ops "$@" \
  -Iinclude -Istarst3r_amd/csrc -c starst3r_amd/csrc/$F -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "Function Name|VGPRs:|Spill|Occupancy|LDS Size" | sed 's/.*remark: *//' | paste - - - - - - | grep "$K" | sed 's/\[-Rpass[^]]*\]//g'
