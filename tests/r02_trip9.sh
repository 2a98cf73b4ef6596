#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --steps 200 --cpu-budget 0 --no-extras 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:round(v['avg_us_per_launch'],1) for k,v in d['kernels'].items()})"
timeout 600 python -m pytest tests/test_mlp_gpu.py tests/test_group_gpu.py -q -x 2>&1 | tail -3
timeout 300 python bench_hpo.py 1024 50 2>&1 | tail -1 | cut -c1-600
