#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for b in 1024 8192; do
  for ab in 0 128; do echo "== B=$b CS_CHAIN_ABLATE=$ab"; CS_CHAIN_ABLATE=$ab timeout 300 python tests/chain_stamps.py $b 2>&1 | grep -E "^\{|fwd grid|bwd grid|step timeline"; done
done
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_group_gpu.py tests/test_mlp_large_gpu.py -q -x 2>&1 | tail -4
for ab in 0 128; do CS_CHAIN_ABLATE=$ab timeout 600 python bench.py --steps 200 --cpu-budget 0 --no-extras --no-profile 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ablate $ab', d['value'], d['timing']['ms_per_step'])"; done
for b in 1024 3072 16384 65536; do timeout 600 python bench.py --batch $b --steps 100 --cpu-budget 0 --no-extras 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($b, d['value'], d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})"; done
