#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_online_mlp_gpu.py -q -x 2>&1 | tail -12
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_online_mlp_gpu.py 2>&1 | tail -6
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
