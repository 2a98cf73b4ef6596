#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
F="amdgpu.ids\|Warning\|ret = \|d = lambda\|print(f"
{
echo "## tests"; timeout 1500 python -m pytest tests/test_chainw_stream_gpu.py tests/test_hot_mlp_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20
echo "## stream"; timeout 300 python tools/chainw_stamps.py 3072 2>&1 | grep -v "$F"
echo "## per pass"; CS_CHAINW_STREAM=0 timeout 300 python tools/chainw_stamps.py 3072 2>&1 | grep -v "$F"
for r in 1 2 3; do
  echo "## rotation $r stream"; timeout 300 python tools/pub_mlp_time.py 2>&1 | grep "B 3072\|B 8192\|chain_fb"
  echo "## rotation $r per pass"; CS_CHAINW_STREAM=0 timeout 300 python tools/pub_mlp_time.py 2>&1 | grep "B 3072\|B 8192\|chain_fb"
done
} > gpurun_out/r06_chainw_stream.txt 2>&1
cat gpurun_out/r06_chainw_stream.txt
