#!/bin/bash
# Round-4 trip G: the whole GPU suite, then the stream marks.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5 | tee gpurun_out/r04_g_pytest_gpu.log
cp gpurun_out/test_margins.json gpurun_out/r04_test_margins.json 2>/dev/null
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/r04_g_smoke.log
timeout 600 python tools/stream_stamps.py 4 8 8192 2>&1 | tail -16 | tee gpurun_out/r04_g_stream_stamps.log
