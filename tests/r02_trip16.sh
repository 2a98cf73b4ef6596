#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest -m gpu"; timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== sweep (bench.py takes the cooperative chain at <= 2048 columns)"; rm -f gpurun_out/r02_bench_sweep.jsonl
for b in 256 1024 2048 3072 4096 8192 16384 32768 65536; do
  timeout 600 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>&1 | tail -1 >> gpurun_out/r02_bench_sweep.jsonl
done
for b in 256 1024 2048; do
  CS_COOP=0 timeout 600 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); d['note']='CS_COOP=0: one workgroup per row tile'; print(json.dumps(d))" >> gpurun_out/r02_bench_sweep.jsonl
done
python - <<'PY'
import json
for line in open('gpurun_out/r02_bench_sweep.jsonl'):
    d=json.loads(line)
    print(d['config']['per_gpu_batch'], d.get('note',''), d['value'], d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})
PY
