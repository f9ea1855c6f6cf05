#!/usr/bin/env python
"""Alternating same-box A/B of the libraries in build_variants/ (tools/experiments/ab_worktree.sh build, or any script that
leaves v<i>.so + v<i>.txt there):  python tools/experiments/ab_run.py [rounds=6] [--train]
Every round times each variant once on the frozen SYNTH-1M scene (ST3R_BENCH_FREEZE=1; --train: the driver's bench command
instead, Adam included); prints mean +- standard error per variant of the step and of the stages, and the paired differences
against v1.  Boxes differ by several per cent and consecutive runs on one box by ~0.5 %: only alternation resolves 0.01 ms."""
import glob, json, os, shutil, statistics as st, subprocess, sys

rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
train = "--train" in sys.argv
libs = sorted(glob.glob("build_variants/v*.so"))
names = [open(l[:-3] + ".txt").read().strip() for l in libs]
shutil.copy("starst3r_amd/libst3r_hip.so", "/tmp/orig.so")
res = {l: [] for l in libs}
env = dict(os.environ)
if not train:
    env["ST3R_BENCH_FREEZE"] = "1"
cmd = [sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-drift", "--no-scaling-model",
       "--no-config1", "--train-only"]
try:
    for r in range(rounds):
        for l in (libs if r % 2 == 0 else libs[::-1]):
            shutil.copy(l, "starst3r_amd/libst3r_hip.so")
            out = subprocess.run(cmd, env=env, capture_output=True, text=True).stdout.strip().splitlines()
            d = json.loads(out[-1])
            s = d["roofline"]["stage_ms"]
            s = dict(s, step=d["ms_per_step"], blend=s["blend_fwd"] + s["blend_bwd"])
            s["rest"] = s["step"] - s["blend"]
            res[l].append(s)
finally:
    shutil.copy("/tmp/orig.so", "starst3r_amd/libst3r_hip.so")
keys = ["step", "blend", "rest", "blend_fwd", "blend_bwd", "loss", "sort", "sort_depth", "project", "project_bwd", "emit", "scan"]
se = lambda v: (st.pstdev(v) / max(len(v) - 1, 1) ** 0.5) if len(v) > 1 else 0.0
for l, n in zip(libs, names):
    print(f"== {n}")
    print("   " + "  ".join(f"{k} {st.mean(x[k] for x in res[l]):.4f}+-{se([x[k] for x in res[l]]):.4f}" for k in keys if k in res[l][0]))
for l, n in zip(libs[1:], names[1:]):
    print(f"-- {n} minus {names[0]} (paired by round)")
    print("   " + "  ".join(f"{k} {st.mean(b[k] - a[k] for a, b in zip(res[libs[0]], res[l])):+.4f}+-{se([b[k] - a[k] for a, b in zip(res[libs[0]], res[l])]):.4f}"
                            for k in keys if k in res[l][0]))
