#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_loader_gpu.py tests/test_stream_gpu.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert|Mismatch" | tail -4 | tee gpurun_out/r04_t_tests.log
for rep in 1 2; do for v in 0 1 3 2; do
  CS_LOADER_V5=$v timeout 300 python bench_loader.py 64 21600 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('64 timesteps, mode $v (0 = two passes; one pass: 1 = 64 columns x 8 waves, 3 = x 16 waves, 2 = 128 columns x 16 waves):', d['value'], d['ms_per_call'], d['roofline']['frac'])"
done; done | tee gpurun_out/r04_t_loader5_ab.txt
for v in 0 3; do
  CS_LOADER_V5=$v timeout 300 python bench_loader.py 8 21600 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('8 timesteps, mode $v:', d['value'], d['ms_per_call'], d['roofline']['frac'])"
done | tee -a gpurun_out/r04_t_loader5_ab.txt
for v in 3 0; do
  CS_LOADER_V5=$v timeout 600 python bench_stream.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stream, mode $v:', d['value'], d['train_only_columns_per_s'], d['loader_only_columns_per_s'], round(d['value']/d['train_only_columns_per_s'],4))"
done | tee -a gpurun_out/r04_t_loader5_ab.txt
