#!/bin/bash
# round 6, trip A: the new / changed GPU tests, then the wide chain's bias-in-accumulator A/B (three rotations)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python -m pytest tests/test_hot_mlp_gpu.py tests/test_mlp_large_gpu.py tests/test_cnn_gpu.py tests/test_group_gpu.py tests/test_online_mlp_gpu.py tests/test_mlp_gpu.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r06_tests_a.log
cp gpurun_out/test_margins.json gpurun_out/r06_test_margins_a.json 2>/dev/null
for r in 1 2 3; do
  echo "rotation $r base(CWD_BIAS_ACC=0)"; CLIMSIM_HIP_LIB=$PWD/climsim_amd/variants/lib_cwd_nobias.so python tools/pub_mlp_time.py 2>&1 | grep -v amdgpu.ids | head -2
  echo "rotation $r new (CWD_BIAS_ACC=1)"; python tools/pub_mlp_time.py 2>&1 | grep -v amdgpu.ids | head -2
done > gpurun_out/r06_chainw_bias_ab.txt 2>&1
CLIMSIM_HIP_LIB=$PWD/climsim_amd/variants/lib_cwd_nobias.so python -m pytest tests/test_group_gpu.py -m gpu -q -k "elu_family_and_wide" 2>&1 | tail -3 > gpurun_out/r06_group_nobias.log
cp gpurun_out/test_margins.json gpurun_out/r06_test_margins_nobias.json 2>/dev/null
tail -12 gpurun_out/r06_tests_a.log; cat gpurun_out/r06_chainw_bias_ab.txt; cat gpurun_out/r06_group_nobias.log
