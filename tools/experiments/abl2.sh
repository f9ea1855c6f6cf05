#!/bin/bash
# run-time ablations of the blend backward (debug flags >> 8 = abl bits of k_blend_bwd), frozen scene
for A in 0 1 2 3 4 8 12; do
  F=$((A*256))
  echo "== abl=$A"
  ST3R_DEBUG_FLAGS=$F ST3R_BENCH_FREEZE=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('ms', round(d['ms_per_step'], 3), 'fwd', round(s['blend_fwd'], 3), 'bwd', round(s['blend_bwd'], 3), 'loss', round(s['loss'], 3), 'pbwd', round(s['project_bwd'], 3))"
done
echo "== old kernel"; ST3R_DEBUG_FLAGS=2 ST3R_BENCH_FREEZE=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('ms', round(d['ms_per_step'], 3), 'fwd', round(s['blend_fwd'], 3), 'bwd', round(s['blend_bwd'], 3))"
