#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for f in 0 8; do
for b in 8192 65536; do
  CS_FLAGS=$f timeout 300 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 2>/dev/null | python tests/summ.py flags=$f
done
done
CS_FLAGS=0 python tests/chain_stamps.py 65536
CS_FLAGS=8 python tests/chain_stamps.py 65536
