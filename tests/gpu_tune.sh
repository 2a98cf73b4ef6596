#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for w in 0 1; do
for b in 1024 8192; do
  CS_WGRAD3=$w timeout 300 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 2>/dev/null | python tests/summ.py wgrad3=$w
done
done
for sk in 3 5 10; do
  CS_WGRAD_SPLITK=$sk timeout 300 python bench.py --batch 8192 --steps 100 --warmup 10 --cpu-budget 0 2>/dev/null | python tests/summ.py wgrad3 splitk=$sk
done
