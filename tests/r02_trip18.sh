#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pair_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -5
CS_COOP_WARM=4 timeout 600 python -m pytest tests/test_pair_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tests/pair_time.py 4096 8192 2>&1 | grep -E "^[0-9]"
timeout 300 python tests/pair_stamps.py 8192 2>&1 | tail -19
