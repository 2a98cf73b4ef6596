#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_mlp_gpu.py tests/test_mlp_large_gpu.py tests/test_group_gpu.py tests/test_hpo_gpu.py tests/test_coop_gpu.py tests/test_dp_gpu.py tests/test_online_mlp_gpu.py tests/test_stream_gpu.py tests/test_dp_two_ranks_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/r05_parity.log; cat gpurun_out/r05_parity.log
