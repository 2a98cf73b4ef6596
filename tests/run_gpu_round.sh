#!/bin/bash
# One GPU-box round trip: parity tests, smoke, bench, batch sweep, rocprof kernel stats.
# Outputs under gpurun_out/ (copy what should be judged into profiles/).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke" ; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench" ; timeout 900 python bench.py 2>&1 | tail -2 | tee gpurun_out/bench.log
echo "== sweep"
for b in 1024 3072 8192 32768 65536; do
  timeout 600 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 2>&1 | tail -1 | tee -a gpurun_out/bench_sweep.log
done
echo "== rocprof stats"
cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $REPO/bench.py --steps 100 --warmup 10 --cpu-budget 0 --no-profile > $REPO/gpurun_out/rocprof_run.log 2>&1
cd $REPO
find /tmp/prof -name "*kernel_stats*" -exec cp {} gpurun_out/rocprof_kernel_stats.csv \;
head -20 gpurun_out/rocprof_kernel_stats.csv
