#!/bin/bash
# the configs[1] example's stage times under every library of build_variants/ (same box)
cp starst3r_amd/libst3r_hip.so /tmp/orig.so
for r in 1 2; do for f in build_variants/v*.so; do
  echo "== $(cat ${f%.so}.txt)"; cp $f starst3r_amd/libst3r_hip.so
  python tools/experiments/cfg1_stage_times.py 300 2>&1 | tail -2 | cut -c1-330
done; done
cp /tmp/orig.so starst3r_amd/libst3r_hip.so
