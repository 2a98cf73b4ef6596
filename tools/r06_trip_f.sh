#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_cnn_gpu.py -m gpu -q -x 2>&1 | tail -2
for v in abl1 abl1_exp16 abl1_exp2; do
  echo "== k_conv_wgrad3l $v"; CLIMSIM_HIP_LIB=$PWD/climsim_amd/variants/lib_cw$v.so timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | grep "kernel span\|tap tiles"
done
} > gpurun_out/r06_cw3_f.txt 2>&1
cat gpurun_out/r06_cw3_f.txt
