#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
echo "== default (no prefetch)"; timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | grep "kernel span\|tap tiles"
for v in pf4 pf8 pf4_abl6 pf8_abl6; do
  echo "== k_conv_wgrad3l $v"; CLIMSIM_HIP_LIB=$PWD/climsim_amd/variants/lib_cw3$v.so timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | grep "kernel span\|tap tiles"
done
for v in pf4 pf8; do echo -n "$v parity: "; CLIMSIM_HIP_LIB=$PWD/climsim_amd/variants/lib_cw3$v.so timeout 600 python -m pytest tests/test_cnn_gpu.py -m gpu -q -x -k "queue_forms or batch512_loss or loss_and_gradients" 2>&1 | tail -1; done
} > gpurun_out/r06_cw3_g.txt 2>&1
cat gpurun_out/r06_cw3_g.txt
