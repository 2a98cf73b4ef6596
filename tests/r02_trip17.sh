#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
bash tests/gpu_diag_cnn.sh 2>&1 | grep -E "^==|k_conv|k_cnn|k_loader" > gpurun_out/r02_cnn_pmc.txt
cd /tmp && rm -rf /tmp/profc && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc -- python3 $REPO/bench_cnn.py 512 20 > $REPO/gpurun_out/rocprof_cnn.log 2>&1
cd $REPO; find /tmp/profc -name "*kernel_stats*" -exec cp {} gpurun_out/r02_cnn_rocprofv3_kernel_stats_b512.csv \;
grep -E "k_conv|k_cnn" gpurun_out/r02_cnn_rocprofv3_kernel_stats_b512.csv | cut -c1-150
head -40 gpurun_out/r02_cnn_pmc.txt
