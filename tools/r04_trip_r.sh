#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_metrics_gpu.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 | tee gpurun_out/r04_r_tests.log
for rep in 1 2; do for v in 2 1; do
  CS_METRICS_CW=$v timeout 300 python bench_metrics.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('columns per workgroup $v:', d['ms_per_call'], d['roofline']['frac'])"
done; done | tee gpurun_out/r04_r_metrics_cw.txt
