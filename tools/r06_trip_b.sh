#!/bin/bash
# round 6, trip B: the tap-shared conv weight-gradient kernel - parity first, then A/B step times against k_conv_wgrad2l
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cnn_gpu.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r06_tests_b.log
tail -8 gpurun_out/r06_tests_b.log
for r in 1 2 3; do
  echo -n "rot $r CS_CW3=0: "; CS_CW3=0 timeout 300 python tools/cnn_train_time.py 512 2>&1 | tail -1
  echo -n "rot $r CS_CW3=1: "; CS_CW3=1 timeout 300 python tools/cnn_train_time.py 512 2>&1 | tail -1
done 2>&1 | tee gpurun_out/r06_cw3_ab.txt
