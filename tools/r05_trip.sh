#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05_adb4.txt; : > $O
timeout 900 python -m pytest tests/test_mlp_large_gpu.py tests/test_mlp_gpu.py -x -q 2>&1 | tail -3 >> $O
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export CLIMSIM_HIP_LIB=$PWD/climsim_amd/libabl_base.so; else unset CLIMSIM_HIP_LIB; fi
  echo "== $lib" >> $O
  timeout 300 python tools/predict_time.py 2>&1 | grep -E "^(65536|131072)" >> $O
  for b in 65536 32768; do timeout 300 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['per_gpu_batch'], d['value'], d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})" >> $O; done
done; done
cat $O
