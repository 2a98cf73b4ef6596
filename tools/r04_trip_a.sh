#!/bin/bash
# Round-4 trip A: the tests this round added or touched, then the default bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== new tests"
timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_dp_ipc_gpu.py "tests/test_mlp_gpu.py::test_fit_predict_evaluate_api" tests/test_group_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/r04_a_tests1.log
timeout 1500 python -m pytest tests/test_cnn_gpu.py -x -q -k "batch512" -s 2>&1 | tail -15 | tee gpurun_out/r04_a_tests2.log
echo "== bench (driver's command)"
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r04_a_bench.err | tail -1 > gpurun_out/r04_a_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_a_bench.json').read())
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})
print('sweep', json.dumps(d.get('sweep'))[:1500])
print('stream', d.get('stream')); print('cnn', d.get('cnn'))
PY
