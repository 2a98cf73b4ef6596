#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest -m gpu"; timeout 2700 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -16
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r02_bench_b8192.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench_b8192.json').read())
print(d['value'], d['ms_per_step'], d['timing']['gpu_sclk_mhz'], d['roofline']['frac'], {k:round(v['avg_us_per_launch'],1) for k,v in d['kernels'].items()})
PY
echo "== bench forced dist"; CS_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 100 --cpu-budget 0 --no-extras --no-profile 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['comm'], d['strong']['value'])"
