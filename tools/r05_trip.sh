#!/bin/bash
mkdir -p gpurun_out
{
echo "== tests"; timeout 1500 python -m pytest tests/test_mlp_gpu.py tests/test_mlp_large_gpu.py -x -q -m gpu 2>&1 | tail -4
for w in 8 4 8 4; do echo "== CS_WGRAD3_WAVES=$w"; CS_WGRAD3_WAVES=$w timeout 120 python tools/step_time.py 8192 2>&1 | grep -v amdgpu | tail -1; done
for ab in 8 9; do echo "== waves 8 CS_WGRAD_ABLATE=$ab"; CS_WGRAD_ABLATE=$ab timeout 120 python tools/step_time.py 8192 2>&1 | grep -v amdgpu | tail -1; done
for b in 3072 4096 6144; do for w in 8 4; do echo "== batch $b waves $w"; CS_WGRAD3_WAVES=$w timeout 120 python tools/step_time.py $b 2>&1 | grep -v amdgpu | tail -1; done; done
} > gpurun_out/r05_wgrad8.log 2>&1
cat gpurun_out/r05_wgrad8.log
