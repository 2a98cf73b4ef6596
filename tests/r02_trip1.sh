#!/bin/bash
# round-2 first GPU trip: full -m gpu suite, bench (1 GPU + forced one-rank dist path), chain stamps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -45 > gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['timing'], d['roofline'], {k:(round(v['ms_per_step'],4),v['launches_per_step']) for k,v in d['kernels'].items()}, d['cpu_baseline'])
PY
echo "== bench forced dist (one-rank RCCL)"; CS_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 100 --cpu-budget 0 --no-extras --no-profile 2>&1 | tail -1 > gpurun_out/bench_dist1.log; cut -c1-1500 gpurun_out/bench_dist1.log
for b in 1024 4096 8192; do echo "== stamps $b"; timeout 300 python tests/chain_stamps.py $b 2>&1 | tail -12; done > gpurun_out/stamps.log 2>&1
cat gpurun_out/stamps.log
