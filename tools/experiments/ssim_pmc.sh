#!/bin/bash
# What bounds the fused loss kernel: hardware counters of k_ssim_fused* (GPU box): bash tools/experiments/ssim_pmc.sh [lib.so]
# Separate PMC passes with --kernel-trace only.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ssimpmc; rm -rf $OUT; mkdir -p $OUT
if [ -n "${1:-}" ]; then cp starst3r_amd/libst3r_hip.so /tmp/orig_ssim.so; cp $1 starst3r_amd/libst3r_hip.so; fi
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT/p$i -- python $ROOT/tools/time_loss.py 4 > $OUT/p$i.log 2>&1
done
cd $ROOT
[ -n "${1:-}" ] && cp /tmp/orig_ssim.so starst3r_amd/libst3r_hip.so
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in sorted(glob.glob(f"{out}/p?")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in glob.glob(f"{p}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            if r["Kernel_Name"].startswith("k_ssim_fused"):
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for name, d in acc.items():
        vals = sorted(d.values())
        print(name, "launches", len(vals), "median per launch %.5g" % vals[len(vals) // 2])
PY
