#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_group_gpu.py tests/test_mlp_large_gpu.py tests/test_cnn_gpu.py -m gpu -q 2>&1 | tail -12 > gpurun_out/r06_tests_i.log; tail -6 gpurun_out/r06_tests_i.log
timeout 900 python bench.py --steps 5 --warmup 2 --min-seconds 0.05 --rows 65536 --cpu-budget 3 --no-profile --extras pub_mlp > gpurun_out/r06_bench_small.json 2> gpurun_out/r06_bench_small.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_small.json').read().strip().splitlines()[-1])
a=d["heldout"]["against_cpu_restatement"]
print(json.dumps({k:a[k] for k in a if k in ("seconds","check","rel_diff_mae_all_outputs","min_R2","error")}, indent=None)[:3000])
PY
