#!/bin/bash
mkdir -p gpurun_out
{
for v in 40 8; do
  L=$PWD/climsim_amd/variants/sabl$v.so
  echo "== staged CV3_ABL=${v:-0}"; BRIEF=1 CLIMSIM_HIP_LIB=$L timeout 120 python tools/cnn_stamps.py 512 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r05_e.log 2>&1
cat gpurun_out/r05_e.log
