#!/bin/bash
# Round-4 trip K: LL exchange in the cooperative chain - parity, A/B against the flag protocol, stamps.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== coop tests (LL)"; timeout 900 python -m pytest tests/test_coop_gpu.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee gpurun_out/r04_k_tests_ll.log
echo "== coop tests (flag protocol)"; CS_COOP_LL=0 timeout 900 python -m pytest tests/test_coop_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r04_k_tests_flag.log
echo "== coop_time LL"; timeout 600 python tools/coop_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_k_coop_time_ll.txt
echo "== coop_time flag"; CS_COOP_LL=0 timeout 600 python tools/coop_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_k_coop_time_flag.txt
echo "== stamps LL 1024"; timeout 300 python tools/coop_stamps.py 1024 2>&1 | grep -v amdgpu.ids | tail -18 | tee gpurun_out/r04_k_coop_stamps_1024.txt
echo "== clock"; timeout 300 python bench.py --steps 50 --warmup 5 --cpu-budget 0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing'].get('gpu_sclk_mhz'))" | tee gpurun_out/r04_k_clock.txt
