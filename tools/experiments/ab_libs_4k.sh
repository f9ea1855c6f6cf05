#!/bin/bash
# A/B of whole libraries (build_variants/v*.so) on a 4K workload: 1 M Gaussians, 4 views of 3840x2160 (tile grid 240 x 135:
# the 64-bit rectangle form since round 5, the 32-bit form before).  GPU box:  bash tools/experiments/ab_libs_4k.sh
cp starst3r_amd/libst3r_hip.so /tmp/orig.so
for rep in 1 2; do
for f in build_variants/v*.so; do
  echo "== $(cat ${f%.so}.txt)"; cp $f starst3r_amd/libst3r_hip.so
  ST3R_BENCH_FREEZE=1 python bench.py --width 3840 --height 2160 --views 4 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('ms', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in s.items()})"
done; done
cp /tmp/orig.so starst3r_amd/libst3r_hip.so
