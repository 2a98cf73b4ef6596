#!/bin/bash
# round 6, trip M: wide chain - one pass over three / four column tiles per wave (chainw_mma4) against passes of two (CWD_WIDE_PASS=0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hot_mlp_gpu.py tests/test_mlp_large_gpu.py tests/test_group_gpu.py tests/test_online_mlp_gpu.py tests/test_mlp_gpu.py -m gpu -q 2>&1 | tail -6 > gpurun_out/r06_tests_m.log; tail -4 gpurun_out/r06_tests_m.log
{
for r in 1 2 3; do
  echo "rotation $r passes of two (CWD_WIDE_PASS=0)"; CLIMSIM_HIP_LIB=$PWD/climsim_amd/variants/lib_cwd_pass2.so python tools/pub_mlp_time.py 2>&1 | grep -v amdgpu.ids | head -2
  echo "rotation $r one pass of 3 / 4 tiles"; python tools/pub_mlp_time.py 2>&1 | grep -v amdgpu.ids | head -2
done
python tools/pub_mlp_time.py 2>&1 | grep -v amdgpu.ids | tail -1
python tools/chainw_width_time.py 2>&1 | grep -v amdgpu.ids | tail -12
} > gpurun_out/r06_chainw_pass_ab.txt 2>&1
cat gpurun_out/r06_chainw_pass_ab.txt
