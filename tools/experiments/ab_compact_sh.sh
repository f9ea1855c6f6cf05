#!/bin/bash
# A/B: SH rows 0..3 trained in place (288-byte stride) vs as a compact [N, 4, 3] copy (what gs.run_3dgs_optim does since round 5)
for rep in 1 2; do for F in 0 1; do
  echo "== ST3R_BENCH_COMPACT_SH=$F"
  ST3R_BENCH_COMPACT_SH=$F python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drift --no-scaling-model --train-only 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('ms', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in s.items() if k in ('project','project_bwd','adam')})"
done; done
