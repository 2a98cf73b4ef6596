#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for v in 1 3 4; do
  CS_LOADER_V5=$v timeout 300 python bench_loader.py 64 21600 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('64 timesteps, 64 columns, mode $v (1 = 8 waves, 3 = 16 waves, 4 = 4 waves per workgroup):', d['value'], d['ms_per_call'], d['roofline']['frac'])"
done; done | tee gpurun_out/r04_t_loader5_waves.txt
