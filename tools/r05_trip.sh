#!/bin/bash
mkdir -p gpurun_out
{
echo "== new tests"; timeout 1500 python -m pytest tests/test_mlp_large_gpu.py tests/test_coop_gpu.py tests/test_loader_gpu.py "tests/test_mlp_gpu.py::test_fit_predict_evaluate_api" -x -q -m gpu 2>&1 | tail -8
} > gpurun_out/r05_h.log 2>&1
cat gpurun_out/r05_h.log
