#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_loader_gpu.py tests/test_stream_gpu.py -q -x 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench_loader.py 64 21600 2>&1 | tail -1 | cut -c1-260; done
