#!/bin/bash
# round 6, trip D: compute-loop experiments of k_conv_wgrad3l (-DCW3_EXP: 1 = no sched_barriers, 2 = no boundary redirect, 4 = s_setprio)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
for v in 1 2 3 4; do
  echo "== k_conv_wgrad3l -DCW3_EXP=$v"; CLIMSIM_HIP_LIB=$PWD/climsim_amd/variants/lib_cw3exp$v.so timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | grep "kernel span\|clocks per slab"
done
} > gpurun_out/r06_cw3_exp.txt 2>&1
cat gpurun_out/r06_cw3_exp.txt
