#!/bin/bash
# Round-4 trip I: LL hand-off probe, stream priority probe, shader-clock source check.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== hand-off probe"; timeout 300 tools/handoff_probe.bin 2>&1 | tee gpurun_out/r04_i_handoff.txt
echo "== stream priority probe"; timeout 600 python tools/stream_prio_probe.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r04_i_stream_prio.txt
echo "== clock source"; timeout 300 python bench.py --steps 50 --warmup 5 --cpu-budget 0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing'].get('gpu_sclk_mhz'))" | tee gpurun_out/r04_i_clock.txt
