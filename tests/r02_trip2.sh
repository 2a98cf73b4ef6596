#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== group tests"; timeout 900 python -m pytest tests/test_group_gpu.py tests/test_hpo_gpu.py -q -x 2>&1 | tail -30
echo "== large + rest"; timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_group_gpu.py --deselect tests/test_hpo_gpu.py 2>&1 | tail -30 > gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log
echo "== bench_hpo 1024"; timeout 900 python bench_hpo.py 1024 100 2>&1 | tail -1 | tee gpurun_out/hpo_1024.json | cut -c1-3000
echo "== bench_hpo 3072"; timeout 900 python bench_hpo.py 3072 60 2>&1 | tail -1 | tee gpurun_out/hpo_3072.json | cut -c1-3000
