#!/bin/bash
# round 6, trip C: stamps of k_conv_wgrad3l (tap-shared tiles) and its ablation builds (-DCW_ABL: 1 = no LDS-DMA requests, 2 = no fragment
# reads, 4 = no MFMAs, 6 = requests only), the old kernel's stamps beside them
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
echo "== k_conv_wgrad3l (default)"; timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | tail -20
echo "== k_conv_wgrad2l (CS_CW3=0)"; CS_CW3=0 timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | tail -9 | head -8
for v in 1 2 4 6; do
  echo "== k_conv_wgrad3l -DCW_ABL=$v"; CLIMSIM_HIP_LIB=$PWD/climsim_amd/variants/lib_cwabl$v.so timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | grep "kernel span\|clocks per slab"
done
} > gpurun_out/r06_cw3_stamps.txt 2>&1
cat gpurun_out/r06_cw3_stamps.txt
