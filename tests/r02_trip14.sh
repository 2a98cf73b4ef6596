#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_coop_gpu.py -q -x 2>&1 | tail -3
CS_COOP_WARM=4 timeout 600 python -m pytest tests/test_coop_gpu.py -q -x 2>&1 | tail -2
echo "no L2 prefetcher:"; CS_COOP_WARM=8 timeout 300 python tests/coop_time.py 2>&1 | grep -E "^(256|1024|2048)"
echo "write-through (sc1) path forced:"; CS_COOP_WARM=4 timeout 300 python tests/coop_time.py 2>&1 | grep -E "^(256|1024|2048)"
