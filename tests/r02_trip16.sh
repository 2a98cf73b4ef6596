#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cnn_gpu.py tests/test_dp_gpu.py tests/test_dp_two_ranks_gpu.py -q -x -p no:cacheprovider 2>&1 | grep -v -i "rccl\|hostname" | tail -15
CS_CNN_OPT_TILES=1 timeout 300 python bench_cnn.py 512 20 2>&1 | tail -4
timeout 300 python bench_cnn.py 512 20 2>&1 | tail -4
