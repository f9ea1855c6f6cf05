#!/bin/bash
# build experiment variants of the library (ST3R_EXP=n) and time the bench stages for each
set -e
for e in "$@"; do
  ST3R_DEFS="-DST3R_EXP=$e" python -m starst3r_amd.build --force >/dev/null 2>&1
  cp starst3r_amd/libst3r_hip.so /tmp/lib_exp$e.so
done
