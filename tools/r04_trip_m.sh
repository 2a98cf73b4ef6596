#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_coop_gpu.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6 | tee gpurun_out/r04_m_tests_coop.log
timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_mlp_gpu.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6 | tee gpurun_out/r04_m_tests_bench.log
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r04_m_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_m_bench.json').read())
print(d['value'], d['ms_per_step'], d['timing'].get('gpu_sclk_mhz'))
print({b:{k:v[k] for k in ('value','ms_per_step','whole_step_frac','kernels_us')} for b,v in d['sweep'].items()})
PY
