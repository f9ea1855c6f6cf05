#!/bin/bash
# stage times of the train step at 1, 2, 4, 8 local views (what a rank of an 8-, 4-, 2-, 1-GPU job runs)
for V in ${@:-1 2 4 8}; do
  python bench.py --views $V --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('views', $V, 'ms/step', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in s.items()})"
done
