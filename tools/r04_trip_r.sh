#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_metrics_gpu.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 | tee gpurun_out/r04_r_tests.log
for v in 1 0 1 0; do
  CS_METRICS_V4=$v timeout 300 python bench_metrics.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('v4=$v', d['ms_per_call'], d['roofline']['frac'])"
done | tee gpurun_out/r04_r_metrics_ab.txt
