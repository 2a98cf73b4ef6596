#!/bin/bash
# One GPU-box round trip: parity tests, smoke, bench, batch sweep, rocprof kernel stats.
# Outputs under gpurun_out/ (copy what should be judged into profiles/).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
echo "== smoke" ; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== bench" ; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'], {k:(round(v['ms_per_step'],4),v['launches_per_step']) for k,v in d['kernels'].items()}, d['cpu_baseline'])
PY
echo "== sweep"; rm -f gpurun_out/bench_sweep.log
for b in 1024 3072 4096 8192 16384 32768 65536; do
  timeout 600 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>&1 | tail -1 >> gpurun_out/bench_sweep.log
done
python - <<'PY'
import json
for line in open('gpurun_out/bench_sweep.log'):
    try: d=json.loads(line)
    except Exception: print(line[:200]); continue
    print(d['config']['per_gpu_batch'], d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items() if v['launches_per_step']>0})
PY
if [ "${1:-}" != "noprof" ]; then
echo "== rocprof stats"
cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $REPO/bench.py --steps 100 --warmup 10 --cpu-budget 0 --no-profile --no-extras > $REPO/gpurun_out/rocprof_run.log 2>&1
cd $REPO
find /tmp/prof -name "*kernel_stats*" -exec cp {} gpurun_out/rocprof_kernel_stats.csv \;
grep -E "k_|rocclr" gpurun_out/rocprof_kernel_stats.csv | cut -c1-160
fi
