#!/bin/bash
# loss-stage time against the number of row bands of the SSIM kernel:  bash tools/ssim_bands.sh <views> <bands...>
V=$1; shift
for B in "$@"; do
  ST3R_SSIM_BANDS=$B python bench.py --views $V --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('views', $V, 'bands', $B, 'loss', round(s['loss'], 4), 'ms/step', round(d['ms_per_step'], 3))"
done
