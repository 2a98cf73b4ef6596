#!/bin/bash
# CNN refresh after the strip-form optimiser: bench line, kernel-trace summary, FETCH/WRITE of the optimiser launch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python3 bench_cnn.py 512 20 2>/dev/null | tail -1 > gpurun_out/cnn_bench.json; cat gpurun_out/cnn_bench.json
cd /tmp
rm -rf /tmp/cnnprof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cnnprof -- python3 $REPO/bench_cnn.py 512 20 > $REPO/gpurun_out/cnnprof.log 2>&1
find /tmp/cnnprof -name "*kernel_stats*" -exec cp {} $REPO/gpurun_out/cnn_kernel_stats.csv \;
head -8 $REPO/gpurun_out/cnn_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $REPO/bench_cnn.py 512 5 > $REPO/gpurun_out/pmc_$c.log 2>&1
  find /tmp/pmc_$c -name "*counter_collection*" -exec cp {} $REPO/gpurun_out/pmc_cnn_$c.csv \;
  echo "== $c"; python3 $REPO/tests/pmc_summ.py $REPO/gpurun_out/pmc_cnn_$c.csv | grep -i optimizer
done
