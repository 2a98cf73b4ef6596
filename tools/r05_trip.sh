#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05_cnn_wgrad_issue.txt; : > $O
for a in 16 22; do
  export CLIMSIM_HIP_LIB=$PWD/climsim_amd/libabl_$a.so
  echo "== CW_ABL=$a (16 = loaders request rows without the per-lane source selection; 22 = that, no fragment reads, no MFMAs)" >> $O
  timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | tail -16 | head -7 >> $O
done
cat $O
