#!/bin/bash
# Round-4 trip F: after the revert of early priming - parity subset, stream phase marks.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_mlp_large_gpu.py tests/test_mlp_gpu.py tests/test_stream_gpu.py tests/test_dp_gpu.py tests/test_dp_two_ranks_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/r04_f_tests.log
timeout 600 python tools/stream_stamps.py 4 8 8192 2>&1 | tail -16 | tee gpurun_out/r04_f_stream_stamps.log
timeout 120 tools/dephase_probe.bin 2>&1 | tee gpurun_out/r04_dephase_probe.txt
