#!/bin/bash
# Row bands of the fused loss (ST3R_SSIM_BANDS) for every library of build_variants/ (GPU box), tools/time_loss.py
cp starst3r_amd/libst3r_hip.so /tmp/orig_b.so
for f in build_variants/v*.so; do
  cp $f starst3r_amd/libst3r_hip.so
  for B in 3 4 5 6 7 8 9 10 12; do
    echo "$(cat ${f%.so}.txt) bands=$B $(ST3R_SSIM_BANDS=$B python tools/time_loss.py 40 | grep -o '[0-9.]* ms')"
  done
done
cp /tmp/orig_b.so starst3r_amd/libst3r_hip.so
