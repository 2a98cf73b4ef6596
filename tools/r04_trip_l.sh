#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do
for v in "0 0" "1 0"; do set -- $v
  CS_COOP_LL=$1 CS_COOP_WARM=$2 timeout 300 python tools/coop_time1.py 256 1024 2048 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/r04_l_variants.txt
timeout 600 python -m pytest tests/test_coop_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
