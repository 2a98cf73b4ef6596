#!/bin/bash
# round 5, trip b: k_conv3 phase stamps + rocprofv3 kernel stats
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
{
echo "== stamps"; timeout 300 python tools/cnn_stamps.py 512
echo "== kernel stats k_conv3"
cd /tmp && rm -rf /tmp/profc && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc -- python3 $REPO/bench_cnn.py 512 20 > /dev/null 2>&1
cd $REPO; find /tmp/profc -name "*kernel_stats*" -exec head -8 {} \; | cut -c1-120
} > gpurun_out/r05_b.log 2>&1
cat gpurun_out/r05_b.log
