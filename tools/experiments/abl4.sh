#!/bin/bash
# A/B of run-time variants selected by ST3R_DEBUG_FLAGS (see common.h): per-kernel trace + stage times per variant
for FL in ${@:-0 16}; do
  echo "== ST3R_DEBUG_FLAGS=$FL"
  ST3R_DEBUG_FLAGS=$FL timeout 600 bash tools/ktrace.sh fl$FL > /dev/null 2>&1
  grep -E "k_gather|k_blend_bwd|k_project_sh_bwd" gpurun_out/kt_fl$FL.md | cut -c1-110
  ST3R_DEBUG_FLAGS=$FL timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('it/s', round(d['value'], 2), {k: round(v, 3) for k, v in s.items()})"
done
