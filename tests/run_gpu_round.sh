#!/bin/bash
# One GPU-box round trip: parity tests, smoke, bench, rocprof kernel stats. Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke" ; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench" ; timeout 900 python bench.py --steps 100 --warmup 10 2>&1 | tail -3 | tee gpurun_out/bench.log
