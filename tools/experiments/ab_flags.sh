#!/bin/bash
# A/B of st3r_ctx_set_debug flag sets on the frozen SYNTH-1M scene (one library, several ST3R_DEBUG_FLAGS values):
#   tools/ab_flags.sh 0 128 ...      (run on the GPU box)
for F in "$@"; do
  echo "== ST3R_DEBUG_FLAGS=$F"
  ST3R_DEBUG_FLAGS=$F ST3R_BENCH_FREEZE=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('ms', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in s.items()}, 'records', d['config']['n_isects_kept_after_exact_culling'])"
done
