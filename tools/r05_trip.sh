#!/bin/bash
mkdir -p gpurun_out
{
for ab in 0 4 16 20 0 4; do echo "== CS_CHAIN_ABLATE=$ab"; CS_CHAIN_ABLATE=$ab timeout 120 python tools/step_time.py 8192 2>&1 | grep -v amdgpu | tail -1; done
} > gpurun_out/r05_chain_ablate.log 2>&1
cat gpurun_out/r05_chain_ablate.log
