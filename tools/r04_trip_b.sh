#!/bin/bash
# Round-4 trip B: fused optimiser parity + A/B, the tests trip A never reached, default bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== fused optimiser parity"
timeout 900 python -m pytest tests/test_mlp_large_gpu.py -x -q -k "folded or without_gradient_atomics" 2>&1 | tail -8 | tee gpurun_out/r04_b_tests0.log
echo "== A/B fused optimiser (ms per step, kernels)"
for rep in 1 2; do for f in 0 1; do
  CS_WGRAD_FUSE_OPT=$f timeout 300 python bench.py --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fuse=$f', d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})"
done; done 2>&1 | tee gpurun_out/r04_b_ab.log
echo "== new tests"
timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_dp_ipc_gpu.py "tests/test_mlp_gpu.py::test_fit_predict_evaluate_api" tests/test_group_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/r04_b_tests1.log
timeout 1500 python -m pytest tests/test_cnn_gpu.py -x -q -k "batch512" -s 2>&1 | tail -15 | tee gpurun_out/r04_b_tests2.log
echo "== mlp suites"
timeout 1500 python -m pytest tests/test_mlp_gpu.py tests/test_mlp_large_gpu.py tests/test_coop_gpu.py tests/test_hpo_gpu.py tests/test_stream_gpu.py tests/test_dp_gpu.py tests/test_dp_two_ranks_gpu.py -x -q 2>&1 | tail -8 | tee gpurun_out/r04_b_tests3.log
echo "== bench (driver's command)"
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r04_b_bench.err | tail -1 > gpurun_out/r04_b_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_b_bench.json').read())
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})
print('sweep', json.dumps(d.get('sweep'))[:1500])
print('stream', d.get('stream')); print('cnn', d.get('cnn')); print('pub', d.get('pub_mlp'))
PY
