#!/bin/bash
# round 6, trip H: whole GPU suite, then the CNN profile trip (bench line, rocprofv3 kernel stats, MFMA counters)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r06_gpu_suite.log; tail -5 gpurun_out/r06_gpu_suite.log
cp gpurun_out/test_margins.json gpurun_out/r06_test_margins.json 2>/dev/null
bash tools/cnn_trip.sh 06 2>&1 | tail -25
