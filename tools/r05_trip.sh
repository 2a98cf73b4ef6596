#!/bin/bash
mkdir -p gpurun_out
{
echo "== metrics tests"; timeout 600 python -m pytest tests/test_metrics_gpu.py tests/test_accept_real_gpu.py -x -q -m gpu 2>&1 | tail -5
for v in 1 1 0; do echo "== CS_METRICS_V5=$v"; CS_METRICS_V5=$v timeout 120 python bench_metrics.py 2>&1 | tail -1 | cut -c1-220; done
} > gpurun_out/r05_metrics.log 2>&1
cat gpurun_out/r05_metrics.log
