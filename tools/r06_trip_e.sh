#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_cnn_gpu.py -m gpu -q -x 2>&1 | tail -3
echo "== k_conv_wgrad3l (default)"; timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | grep "kernel span\|clocks per slab"
echo "== -DCW_ABL=1 (no requests)"; CLIMSIM_HIP_LIB=$PWD/climsim_amd/variants/lib_cwabl1.so timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | grep "kernel span\|clocks per slab"
for r in 1 2; do
  echo -n "rot $r CS_CW3=0: "; CS_CW3=0 timeout 300 python tools/cnn_train_time.py 512 2>&1 | tail -1
  echo -n "rot $r CS_CW3=1: "; CS_CW3=1 timeout 300 python tools/cnn_train_time.py 512 2>&1 | tail -1
done
} > gpurun_out/r06_cw3_e.txt 2>&1
cat gpurun_out/r06_cw3_e.txt
