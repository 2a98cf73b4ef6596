#!/bin/bash
# round 5, trip f: k_conv3 with the register-staged loaders: parity, stamps, step time
mkdir -p gpurun_out
{
echo "== tests"; timeout 900 python -m pytest tests/test_cnn_gpu.py -x -q -m gpu 2>&1 | tail -8
echo "== stamps"; BRIEF=1 timeout 120 python tools/cnn_stamps.py 512 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
CS_CNN_STREAM=1 timeout 120 python tools/cnn_train_time.py 512 2>&1 | tail -1
CS_CNN_STREAM=0 timeout 120 python tools/cnn_train_time.py 512 2>&1 | tail -1
done
} > gpurun_out/r05_f.log 2>&1
cat gpurun_out/r05_f.log
