#!/usr/bin/env python
"""Per-kernel totals of rocprofv3 --pmc passes (counter_collection.csv), as a markdown table.
    python tools/pmc_summary.py gpurun_out/prof_x/pmc_*/runc/*_counter_collection.csv > profiles/x_pmc.md
Counter values are summed over the dispatches of a kernel and divided by the launch count (per-launch means)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name).split("(")[0]
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


if __name__ == "__main__":
    main(sys.argv[1:])
