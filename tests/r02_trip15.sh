#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_coop_gpu.py -q -x 2>&1 | tail -3
CS_COOP_WARM=4 timeout 600 python -m pytest tests/test_coop_gpu.py -q -x 2>&1 | tail -2
timeout 300 python tests/coop_time.py 2>&1 | grep -E "^[0-9]"
timeout 300 python tests/coop_stamps.py 1024 2>&1 | tail -15
