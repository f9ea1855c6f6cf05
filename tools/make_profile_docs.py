#!/usr/bin/env python
"""Turn the output of tools/profile_round.sh <tag> (gpurun_out/prof_<tag>) and a default bench line into the files kept
under profiles/:  <tag>_kernel_stats.md, <tag>_rocprof_kernel_stats.csv, <tag>_pmc.md, <tag>_bench_default.json.
    python tools/make_profile_docs.py r2a gpurun_out/prof_r2a/bench.json "title note" [commit]
(round 2: tools/profile_round2.sh; also writes profiles/pmc_traffic.json for bench.py's roofline.traffic) """
import collections, csv, glob, json, re, shutil, subprocess, sys

tag, bench_json, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
commit = sys.argv[4] if len(sys.argv) > 4 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
ROUND = "Round " + (tag[1] if tag[0] == "r" and tag[1].isdigit() else "?")
T = f"gpurun_out/prof_{tag}"
log = open(f"{T}/trace.log").read()
line = json.loads([l for l in log.splitlines() if l.startswith("{") and '"metric"' in l][-1])
stage = {k: round(v, 3) for k, v in line["roofline"]["stage_ms"].items()}
body = subprocess.run([sys.executable, "tools/rocprof_summary.py", glob.glob(f"{T}/trace/runc/*_kernel_trace.csv")[0]],
                      capture_output=True, text=True).stdout.splitlines()
with open(f"profiles/{tag}_kernel_stats.md", "w") as f:
    f.write(f"# {ROUND} ({note}; commit {commit}) -- rocprofv3 --kernel-trace --stats of the driver's bench command\n\n"
            "SYNTH-1M (1 M Gaussians, 8 x 1920x1080), 1 MI355X; the run also executes the alignment, matching and condensation benches.\n"
            f"Command: `cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_{tag}/trace -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline` (tools/profile_round.sh <tag>).\n"
            f"bench.py's own line in the same (profiled) run: {line['value']:.2f} iters/s; stage_ms (HIP events inside bench.py): {stage}\n"
            "(blend_bwd stage = k_blend_bwd -- the per-pair sums of its slots are taken by k_project_sh_bwd since round 5 --; loss = k_ssim_fused; sort / sort_depth = k_rs_hist + k_rs_scan_hist + k_rs_pass of radix_sort.hip; scan = k_scan_chained; emit = k_isect_gather + k_isect_wg_scan + k_isect_emit_d.)\n"
            f"(the bench line of the same build without the profiler is {tag}_bench_default.json.)\n\n")
    f.write("\n".join(body[2:44]) + "\n")
shutil.copy(glob.glob(f"{T}/trace/runc/*_kernel_stats.csv")[0], f"profiles/{tag}_rocprof_kernel_stats.csv")
shutil.copy(bench_json, f"profiles/{tag}_bench_default.json")


def per_kernel(counter):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for fn in glob.glob(f"{T}/pmc_{counter}/runc/*_counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0]
            acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: acc[k] / len(n[k]) for k in acc}


F, W = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
rows = sorted(((2 * F[k] + W.get(k, 0)) * 1024, k) for k in F
              if not (k.startswith("__amd") or "rocprim" in k or k.startswith("at::") or "elementwise" in k))[::-1]
pmc = subprocess.run([sys.executable, "tools/pmc_summary.py"] + glob.glob(f"{T}/pmc_*/runc/*_counter_collection.csv"),
                     capture_output=True, text=True).stdout
with open(f"profiles/{tag}_pmc.md", "w") as f:
    f.write(f"# {ROUND} ({note}; commit {commit}) -- PMC counters per kernel launch (rocprofv3 --pmc, one counter set per run, --kernel-trace only)\n\n"
            "Passes (tools/profile_round.sh <tag>): FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU | SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY,\n"
            "each around `python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline` (SYNTH-1M, 1 MI355X).  FETCH_SIZE / WRITE_SIZE are KiB as reported.\n\n"
            "HBM traffic per launch as MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2 for 16 B/lane reads on gfx950, WRITE_SIZE as is):\n\n"
            "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic = (2 FETCH + WRITE) x 1024 B |\n|---|---|---|---|\n")
    for t, k in rows:
        if t > 5e7:
            f.write(f"| {k} | {F[k]:.4g} | {W.get(k, 0):.4g} | {t / 1e9:.2f} GB |\n")
    f.write("\nThe x2 rule is calibrated for coalesced streams and over-counts the 48-byte record gathers of the blend kernels, so\n"
            "their figures are upper estimates.\n\n")
    f.write(pmc)
for t, k in rows[:12]:
    print(f"{k:28s} {t / 1e9:6.2f} GB")
subprocess.run([sys.executable, "tools/pmc_summary.py", "--traffic-json", "profiles/pmc_traffic.json", commit] +
               glob.glob(f"{T}/pmc_FETCH_SIZE/runc/*_counter_collection.csv") + glob.glob(f"{T}/pmc_WRITE_SIZE/runc/*_counter_collection.csv") +
               glob.glob(f"{T}/pmc_SQ1/runc/*_counter_collection.csv"))
