#!/bin/bash
# Which shader clock do the step's kernels actually run at?  GRBM_GUI_ACTIVE (cycles the chip was active, per XCD) against
# the kernel's duration from the same pass's kernel trace.  GPU box:  bash tools/experiments/clock_pmc.sh
ROOT=$(pwd); OUT=$ROOT/gpurun_out/clockpmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -f csv -d $OUT/p -- python $ROOT/bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-drift --no-scaling-model --train-only > $OUT/p.log 2>&1
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = {}
for fn in glob.glob(f"{out}/p/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cyc = collections.defaultdict(float)
for fn in glob.glob(f"{out}/p/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[r["Dispatch_Id"]] += float(r["Counter_Value"])
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for d, (name, ns) in dur.items():
    if d in cyc and ns > 20000:
        a = agg[name]; a[0] += cyc[d] / 8.0; a[1] += ns; a[2] += 1     # eight XCDs report their own count
print("kernel | launches | mean us | mean active cycles per XCD | implied clock GHz")
for name, (c, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:40]:40s} | {n:3d} | {ns / n / 1e3:8.1f} | {c / n:12.0f} | {c / ns:.2f}")
PY
