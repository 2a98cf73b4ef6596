#!/bin/bash
# round 5, trip a: k_conv3 (continuous operand stream) parity + A/B against k_conv2
mkdir -p gpurun_out
{
echo "== tests (k_conv3)"; timeout 900 python -m pytest tests/test_cnn_gpu.py -x -q -m gpu 2>&1 | tail -15
echo "== bench_cnn k_conv3"; CS_CNN_STREAM=1 timeout 300 python bench_cnn.py 512 20
echo "== bench_cnn k_conv2"; CS_CNN_STREAM=0 timeout 300 python bench_cnn.py 512 20
echo "== bench_cnn k_conv3"; CS_CNN_STREAM=1 timeout 300 python bench_cnn.py 512 20
echo "== bench_cnn k_conv2"; CS_CNN_STREAM=0 timeout 300 python bench_cnn.py 512 20
} > gpurun_out/r05_a.log 2>&1
tail -40 gpurun_out/r05_a.log
