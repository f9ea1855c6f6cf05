#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or kernel_trace.csv) into a
markdown table: per kernel calls / total / mean / min / max (us), VGPR/SGPR/LDS.
    python tools/rocprof_summary.py gpurun_out/prof/x_results.db [--skip-first N] > profiles/x.md
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    db = sqlite3.connect(path)
    q = "select name, start, end, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x*grid_y*grid_z, workgroup_x*workgroup_y*workgroup_z from kernels order by start"
    return [dict(name=r[0], start=r[1], end=r[2], vgpr=r[3], agpr=r[4], sgpr=r[5], lds=r[6], grid=r[7], wg=r[8])
            for r in db.execute(q)]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append(dict(name=r["Kernel_Name"], start=int(r["Start_Timestamp"]), end=int(r["End_Timestamp"]),
                        vgpr=r.get("VGPR_Count"), agpr=r.get("Accum_VGPR_Count"), sgpr=r.get("SGPR_Count"),
                        lds=r.get("LDS_Block_Size"), grid=r.get("Grid_Size"), wg=r.get("Workgroup_Size")))
    return out


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    return name if len(name) < 70 else name[:67] + "..."


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = defaultdict(list)
    meta = {}
    for r in rows:
        agg[r["name"]].append((r["end"] - r["start"]) / 1e3)
        meta[r["name"]] = r
    tot = sum(sum(v) for v in agg.values())
    print(f"# rocprofv3 kernel trace summary: {path}\n")
    print(f"total kernel time {tot / 1e3:.2f} ms over {len(rows)} dispatches\n")
    print("| kernel | calls | total ms | % | mean us | min us | max us | vgpr | sgpr | lds B | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        m = meta[k]
        print(f"| {short(k)} | {len(v)} | {sum(v) / 1e3:.3f} | {100 * sum(v) / tot:.1f} | {sum(v) / len(v):.1f} | "
              f"{min(v):.1f} | {max(v):.1f} | {m['vgpr']} | {m['sgpr']} | {m['lds']} | {m['wg']} |")


if __name__ == "__main__":
    main()
