#!/bin/bash
# per-stage HIP-event times of the SYNTH-1M train step (run on the GPU box)
python bench.py --steps ${1:-30} --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('it/s', round(d['value'], 2), 'ms', round(d['ms_per_step'], 3))
print({k: round(v, 3) for k, v in d['roofline']['stage_ms'].items()})
"
