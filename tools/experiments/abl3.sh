#!/bin/bash
# time library variants built by tools/abl.sh build (normal training, 5+20 steps): stage times + sort probe
cp starst3r_amd/libst3r_hip.so /tmp/orig.so
for f in build_variants/v*.so; do
  echo "== $(cat ${f%.so}.txt)"; cp $f starst3r_amd/libst3r_hip.so
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['roofline']['stage_ms']
print('it/s', round(d['value'], 2), 'sort_depth', round(s['sort_depth'], 3), 'sort', round(s['sort'], 3), 'loss', round(s['loss'], 3), 'bwd', round(s['blend_bwd'], 3), 'fwd', round(s['blend_fwd'], 3))"
  timeout 100 tools/probe/rsort_probe 0
done
cp /tmp/orig.so starst3r_amd/libst3r_hip.so
