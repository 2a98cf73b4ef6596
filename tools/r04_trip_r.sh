#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
for w in 16 4; do echo "== tests, $w waves"; CS_METRICS_WAVES=$w timeout 600 python -m pytest tests/test_metrics_gpu.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -3; done | tee gpurun_out/r04_r_tests.log
for rep in 1 2; do for w in 16 4; do
  CS_METRICS_WAVES=$w timeout 300 python bench_metrics.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('waves per workgroup $w:', d['ms_per_call'], d['roofline']['frac'])"
done; done | tee gpurun_out/r04_r_metrics_waves.txt
