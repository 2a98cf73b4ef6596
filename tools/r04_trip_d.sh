#!/bin/bash
# Round-4 trip D: trunk + contraction split of the 128-wide stages + stream changes: parity suites, A/B, stamps, stream bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== parity suites"
timeout 1800 python -m pytest tests/test_mlp_large_gpu.py tests/test_mlp_gpu.py tests/test_group_gpu.py tests/test_hpo_gpu.py tests/test_coop_gpu.py tests/test_stream_gpu.py tests/test_online_mlp_gpu.py tests/test_dp_gpu.py tests/test_dp_two_ranks_gpu.py -x -q 2>&1 | tail -8 | tee gpurun_out/r04_d_tests0.log
echo "== A/B (ms per step, kernels us): trunk x ksplit"
for rep in 1 2; do for t in 0 1; do for a in 128 0; do
  CS_CHAIN_TRUNK=$t CS_CHAIN_ABLATE=$a timeout 300 python bench.py --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('trunk=$t ablate=$a', d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})"
done; done; done 2>&1 | tee gpurun_out/r04_d_ab.log
echo "== stamps"; timeout 300 python tools/chain_stamps.py 8192 2>&1 | tail -12 | tee gpurun_out/r04_d_stamps.log
echo "== stream bench"; timeout 600 python bench_stream.py 2>&1 | tail -1 | tee gpurun_out/r04_d_stream.json
timeout 600 python tools/stream_stamps.py 4 8 8192 2>&1 | tail -8 | tee gpurun_out/r04_d_stream_stamps.log
echo "== sweep"
for b in 1024 3072 16384 65536; do
  timeout 300 python bench.py --batch $b --steps 50 --warmup 10 --cpu-budget 0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch=$b', d['value'], d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})"
done 2>&1 | tee gpurun_out/r04_d_sizes.log
