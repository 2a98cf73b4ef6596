#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
echo "## default"; timeout 300 python tools/chainw_stamps.py 3072 2>&1 | grep -v "amdgpu.ids\|Warning\|ret = \|d = lambda\|print(f"
echo "## split probe"; SPLIT=1 CLIMSIM_HIP_LIB=$PWD/climsim_amd/variants/lib_split.so timeout 300 python tools/chainw_stamps.py 3072 2>&1 | grep -v "amdgpu.ids\|Warning\|ret = \|d = lambda\|print(f"
echo "## step time"; timeout 300 python tools/pub_mlp_time.py 2>&1 | grep -v amdgpu.ids
echo "## tests"; timeout 1500 python -m pytest tests -m gpu -q -x -k "wide or chainw or pub or hot or group or online or dropout or elu" 2>&1 | tail -5
} > gpurun_out/r06_chainw_stamps.txt 2>&1
cat gpurun_out/r06_chainw_stamps.txt
