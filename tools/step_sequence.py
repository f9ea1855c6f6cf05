"""Kernel sequence of ONE training step from a rocprofv3 kernel trace (the launches between the last two k_adam4 calls but one):
name, duration, gap to the previous kernel; totals.  usage: python tools/step_sequence.py <rocprofv3 output dir>
(e.g. after `rocprofv3 --kernel-trace -f csv -d gpurun_out/seq -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-only`)."""
import csv, glob, sys
fn = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_adam4")]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["End_Timestamp"])
prev_end = t0
tot_gap = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end; tot_gap += max(gap, 0)
    print("%-60s dur %8.1f us  gap %6.1f us" % (r["Kernel_Name"][:60], (e - s) / 1e3, gap / 1e3))
    prev_end = e
print("step", (prev_end - t0) / 1e3, "us; gaps", tot_gap / 1e3, "us; kernels", b - a)
