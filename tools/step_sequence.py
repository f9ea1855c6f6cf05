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
