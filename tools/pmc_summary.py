#!/usr/bin/env python
"""Per-kernel totals of rocprofv3 --pmc passes (counter_collection.csv), as a markdown table.
    python tools/pmc_summary.py gpurun_out/prof_x/pmc_*/runc/*_counter_collection.csv > profiles/x_pmc.md
Counter values are summed over the dispatches of a kernel and divided by the launch count (per-launch means)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").split("(")[0]
    return name


def main(paths):
    vals = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum
    calls = defaultdict(lambda: defaultdict(set))
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            vals[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    counters = sorted({c for k in vals for c in vals[k]})
    print("| kernel | launches | " + " | ".join(f"{c} / launch" for c in counters) + " |")
    print("|---|---|" + "---|" * len(counters))
    order = sorted(vals, key=lambda k: -max(vals[k].values()))
    for k in order:
        if k.startswith("__amd") or "rocprim" in k or k.startswith("at::") or "elementwise" in k:
            continue
        n = max(len(s) for s in calls[k].values())
        cells = []
        for c in counters:
            m = len(calls[k][c])
            cells.append(f"{vals[k][c] / m:.4g}" if m else "")
        print(f"| {k} | {n} | " + " | ".join(cells) + " |")


# kernels of each stage of the fused train step (the stages bench.py times)
STAGE_KERNELS = {
    "blend_bwd": ("k_blend_bwd", "k_gather_vtile"), "blend_fwd": ("k_blend_fwd_cells",), "loss": ("k_ssim_fused",),
    "project": ("k_project_sh_fwd", "k_reg_reduce"), "project_bwd": ("k_project_sh_bwd",), "adam": ("k_adam",),
    "emit": ("k_isect_gather", "k_isect_wg_scan", "k_isect_emit_d"), "offsets": ("k_isect_offsets32",), "scan": ("k_scan_chained",),
}


def traffic_json(paths, out_path, commit, workload):
    """profiles/pmc_traffic.json for bench.py: HBM bytes per launch and stage = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
    (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950; both counters in KiB),
    from separate --pmc passes, together with the workload and the fingerprint of the kernel sources measured."""
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    vals = defaultdict(lambda: defaultdict(float)); calls = defaultdict(lambda: defaultdict(set))
    dsum, dn = defaultdict(float), defaultdict(int)
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"]).split("<")[0].strip()
            vals[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
        kt = p.replace("counter_collection.csv", "kernel_trace.csv")   # mean kernel durations of the same pass
        if os.path.exists(kt):
            for r in csv.DictReader(open(kt)):
                k = short(r["Kernel_Name"]).split("<")[0].strip()
                dsum[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); dn[k] += 1
    dur = {k: dsum[k] / dn[k] for k in dsum}
    per_kernel = {}
    for k in vals:
        if "FETCH_SIZE" in vals[k] and "WRITE_SIZE" in vals[k]:
            f = vals[k]["FETCH_SIZE"] / len(calls[k]["FETCH_SIZE"]); w = vals[k]["WRITE_SIZE"] / len(calls[k]["WRITE_SIZE"])
            per_kernel[k] = dict(fetch_kib=f, write_kib=w, bytes=(2 * f + w) * 1024)
    stages = {}
    for st, ks in STAGE_KERNELS.items():
        b = sum(per_kernel[k]["bytes"] for k in ks if k in per_kernel)
        if b:
            stages[st] = b
    # SQ_INSTS_VALU per launch of the blend kernels themselves (its own pass) and the kernel's share of its stage's time
    # (kernel-trace durations of the same passes): bench.py's roofline.valu_issue
    valu, share = {}, {}
    for st, main in (("blend_fwd", "k_blend_fwd_cells"), ("blend_bwd", "k_blend_bwd")):
        if "SQ_INSTS_VALU" in vals.get(main, {}):
            valu[st] = vals[main]["SQ_INSTS_VALU"] / len(calls[main]["SQ_INSTS_VALU"])
        tot = sum(dur.get(k, 0.0) for k in STAGE_KERNELS[st])
        if tot > 0 and main in dur:
            share[st] = dur[main] / tot
    rec = dict(commit=commit, csrc_fingerprint=bench.csrc_fingerprint(), workload=workload,
               source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bytes = (2 x FETCH + WRITE) x 1024 per launch",
               traffic_bytes_per_launch=stages, valu_insts_per_launch=valu, kernel_share_of_stage_time=share,
               kernels=per_kernel)
    json.dump(rec, open(out_path, "w"), indent=1)
    print("wrote", out_path, {k: round(v / 1e9, 3) for k, v in stages.items()})


if __name__ == "__main__":
    if sys.argv[1] == "--traffic-json":   # pmc_summary.py --traffic-json OUT COMMIT csv...
        wl = dict(gaussians=1_000_000, views=8, width=1920, height=1080, n_gpus=1)
        traffic_json(sys.argv[4:], sys.argv[2], sys.argv[3], wl)
    else:
        main(sys.argv[1:])
