#!/bin/bash
mkdir -p gpurun_out
{
for v in "" 64 "" 64; do
  if [ -z "$v" ]; then L=$PWD/climsim_amd/libclimsim_hip.so; else L=$PWD/climsim_amd/variants/abl$v.so; fi
  echo "== CV3_ABL=${v:-0}"; BRIEF=1 CLIMSIM_HIP_LIB=$L timeout 120 python tools/cnn_stamps.py 512 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r05_e.log 2>&1
cat gpurun_out/r05_e.log
