#!/bin/bash
# loss stage (k_ssim_fused) against the number of row bands (ST3R_SSIM_BANDS), frozen SYNTH-1M scene
for B in "$@"; do
  echo "== bands $B"
  ST3R_SSIM_BANDS=$B ST3R_BENCH_FREEZE=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-drift 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('ms', round(d['ms_per_step'], 3), 'loss', round(s['loss'], 3))"
done
