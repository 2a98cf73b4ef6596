#!/bin/bash
# round 5, trip d: staggered row tiles (CS_CNN_STAGGER=groups,us)
mkdir -p gpurun_out
{
for s in "1,0" "2,20" "2,12" "4,10" "4,6" "3,12" "8,5"; do
  echo "== CS_CNN_STAGGER=$s"; CS_CNN_STAGGER=$s BRIEF=1 timeout 120 python tools/cnn_stamps.py 512 2>&1 | grep -v amdgpu.ids
  CS_CNN_STAGGER=$s timeout 120 python tools/cnn_train_time.py 512 2>&1 | tail -1
done
} > gpurun_out/r05_d.log 2>&1
cat gpurun_out/r05_d.log
