#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_mlp_gpu.py tests/test_mlp_large_gpu.py tests/test_bench_gpu.py tests/test_hpo_gpu.py tests/test_group_gpu.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee gpurun_out/r04_o_tests.log
