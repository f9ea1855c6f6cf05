#!/bin/bash
# current library: sort tests + sort probe + bench stage times
python -m pytest tests/test_gpu_sort.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
timeout 200 tools/probe/rsort_probe 0
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('it/s', round(d['value'], 2), {k: round(v, 3) for k, v in s.items()})"
