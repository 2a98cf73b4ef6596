#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
V=$PWD/climsim_amd/variants
F="amdgpu.ids\|Warning"
{
for r in 1 2 3; do
  echo "## rotation $r touch"; timeout 300 python tools/cnn_train_time.py 512 2>&1 | grep "ms/step\|rror\|fault"
  echo "## rotation $r no touch"; CLIMSIM_HIP_LIB=$V/lib_cnn_notouch.so timeout 300 python tools/cnn_train_time.py 512 2>&1 | grep "ms/step"
done
echo "## MLP"; timeout 300 python tools/step_time.py 8192 3072 2>&1 | grep -v "$F"; timeout 300 python tools/pub_mlp_time.py 2>&1 | grep "0, 640) B 3072"
echo "## tests"; timeout 2400 python -m pytest tests/test_chainw_stream_gpu.py tests/test_hot_mlp_gpu.py tests/test_mlp_large_gpu.py tests/test_cnn_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | head
} > gpurun_out/r06_cnn_touch_ab.txt 2>&1
cat gpurun_out/r06_cnn_touch_ab.txt
