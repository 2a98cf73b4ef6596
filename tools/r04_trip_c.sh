#!/bin/bash
# Round-4 trip C: chain trunk (continuous weight stream) + fused optimiser (batched tail): parity, A/B, stamps.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== parity: fused optimiser, trunk (mlp suites)"
timeout 1200 python -m pytest tests/test_mlp_large_gpu.py tests/test_mlp_gpu.py tests/test_group_gpu.py tests/test_hpo_gpu.py -x -q 2>&1 | tail -8 | tee gpurun_out/r04_c_tests0.log
echo "== A/B (ms per step, kernels us): trunk x fuse"
for rep in 1 2; do for t in 0 1; do for f in 0 1; do
  CS_CHAIN_TRUNK=$t CS_WGRAD_FUSE_OPT=$f timeout 300 python bench.py --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('trunk=$t fuse=$f', d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})"
done; done; done 2>&1 | tee gpurun_out/r04_c_ab.log
echo "== stamps trunk=1"; CS_CHAIN_TRUNK=1 timeout 300 python tools/chain_stamps.py 8192 2>&1 | tail -12 | tee gpurun_out/r04_c_stamps1.log
echo "== stamps trunk=0"; CS_CHAIN_TRUNK=0 timeout 300 python tools/chain_stamps.py 8192 2>&1 | tail -12 | tee gpurun_out/r04_c_stamps0.log
echo "== other batch sizes (trunk x fuse), step ms"
for b in 1024 3072 4096; do for t in 0 1; do
  CS_CHAIN_TRUNK=$t timeout 300 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch=$b trunk=$t', d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})"
done; done 2>&1 | tee gpurun_out/r04_c_sizes.log
echo "== bench tests"
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q 2>&1 | tail -6 | tee gpurun_out/r04_c_tests1.log
echo "== stream stamps"
timeout 600 python tools/stream_stamps.py 4 8 8192 2>&1 | tail -30 | tee gpurun_out/r04_c_stream_stamps.log
