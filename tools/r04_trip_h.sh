#!/bin/bash
# Round-4 trip H: the one-kernel permutation and the streamed trainer on it.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_shuffle_gpu.py tests/test_stream_gpu.py tests/test_dp_gpu.py tests/test_dp_two_ranks_gpu.py tests/test_mlp_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/r04_h_tests.log
timeout 600 python bench_stream.py 2>&1 | tail -1 | tee gpurun_out/r04_h_stream.json
timeout 600 python tools/stream_stamps.py 4 8 8192 2>&1 | tail -12 | tee gpurun_out/r04_h_stream_stamps.log
