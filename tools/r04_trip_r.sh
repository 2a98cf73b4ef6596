#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for u in 1 2 4 8; do
  CS_METRICS_U=$u timeout 300 python bench_metrics.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('v4 U=$u', d['ms_per_call'], d['roofline']['frac'])"
done; done | tee gpurun_out/r04_r_metrics_u.txt
