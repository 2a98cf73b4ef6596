#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bench_hpo 1024"; timeout 900 python bench_hpo.py 1024 100 2>&1 | tail -3 | tee gpurun_out/hpo_1024.json | cut -c1-3500
for sk in 2 3 4 5 6 7; do CS_WGRAD_SPLITK=$sk python tests/wgrad_ablate.py 8192 2>&1 | tail -1; done
CS_WGRAD_ABLATE=1 python tests/wgrad_ablate.py 8192 2>&1 | tail -1
CS_WGRAD_ABLATE=2 python tests/wgrad_ablate.py 8192 2>&1 | tail -1
