mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_full.json 2> gpurun_out/r6_bench_full.err; echo "rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_bench_full.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("value_steps_180_200"), d.get("config_1_iters_per_sec"), d["n_ranks_seen"], d["config"]["gt_moments_once_per_call_ms"])
print({k: round(v, 4) for k, v in d["roofline"]["stage_ms"].items()})
PY
bash tools/experiments/bench_emulated_ranks.sh 2 > gpurun_out/r6_emul2.log 2>&1; tail -3 gpurun_out/r6_emul2.log | cut -c1-600
