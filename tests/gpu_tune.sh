#!/bin/bash
# quick tuning sweep: prints value + per-kernel ms for several env settings
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
summ() { python - "$@" <<'PY'
import json,sys
for line in sys.stdin:
    try: d=json.loads(line)
    except Exception: print(line[:300]); continue
    print(sys.argv[1:], d['config']['per_gpu_batch'], round(d['value']/1e6,2),'Mcol/s', d['ms_per_step'],'ms', {k:round(v['ms_per_step'],4) for k,v in (d['kernels'] or {}).items()})
PY
}
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for sk in 0 2 4 16; do
  for b in 8192 65536; do
    CS_WGRAD_SPLITK=$sk timeout 300 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 2>&1 | tail -1 | summ splitk=$sk
  done
done
