#!/bin/bash
# Round-4 trip Q: row splits of k_wgrad3 at 8192 columns with the current kernel (3 = default: plain stores into three buffers; > 3: atomics)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
for k in 3 2 4 5 6 7 3; do
  CS_WGRAD_SPLITK=$k timeout 300 python bench.py --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('splitk=$k step ms', d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})"
done | tee gpurun_out/r04_q_wgrad_splits.txt
