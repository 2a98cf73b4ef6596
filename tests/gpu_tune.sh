#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== forced wgrad2"
CS_WGRAD2=1 timeout 900 python -m pytest tests -m gpu -x -q -k "gradients or additivity or training or smoke or curve" 2>&1 | tail -6
for w in 0 1; do
for b in 8192 32768 65536; do
  CS_WGRAD2=$w timeout 300 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 2>/dev/null | python tests/summ.py wgrad2=$w
done
done
