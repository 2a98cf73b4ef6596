#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_coop_gpu.py -q -x 2>&1 | tail -15
timeout 300 python tests/coop_time.py 2>&1 | tail -8
