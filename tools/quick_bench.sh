#!/bin/bash
# one short un-profiled run of the driver's bench command; prints it/s, ms/step and the stage times
# usage: bash tools/quick_bench.sh <tag> [extra bench args]
TAG=${1:-q}; shift
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-drift --no-scaling-model --no-config1 "$@" > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$TAG.json").read().strip().splitlines()[-1])
print("$TAG", round(d["value"], 2), "it/s", round(d["ms_per_step"], 4), "ms")
print({k: round(v, 4) for k, v in d["roofline"]["stage_ms"].items()})
PY
