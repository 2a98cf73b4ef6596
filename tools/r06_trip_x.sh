#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
F="amdgpu.ids\|Warning"
V=$PWD/climsim_amd/variants
{
echo "## tests"; timeout 2400 python -m pytest tests/test_mlp_large_gpu.py tests/test_hot_mlp_gpu.py tests/test_coop_gpu.py tests/test_mlp_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | head
echo "## stamps: bfi select"; timeout 300 python tools/chain_stamps.py 8192 2>&1 | grep -v "$F" | sed -n 6p
echo "## stamps: before"; CLIMSIM_HIP_LIB=$V/lib_prev.so timeout 300 python tools/chain_stamps.py 8192 2>&1 | grep -v "$F" | sed -n 6p
for r in 1 2 3; do
  echo "## rotation $r bfi select"; timeout 300 python tools/step_time.py 8192 3072 16384 2>&1 | grep -v "$F"
  echo "## rotation $r before"; CLIMSIM_HIP_LIB=$V/lib_prev.so timeout 300 python tools/step_time.py 8192 3072 16384 2>&1 | grep -v "$F"
done
} > gpurun_out/r06_chain_bfi_ab.txt 2>&1
cat gpurun_out/r06_chain_bfi_ab.txt
