#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for b in 8192 65536; do
  timeout 300 python bench.py --batch $b --steps 50 --warmup 10 --cpu-budget 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['per_gpu_batch'], d['value'], d['predict'])"
done
