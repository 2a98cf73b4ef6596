#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
F="amdgpu.ids\|Warning\|ret = \|d = lambda\|print(f"
V=$PWD/climsim_amd/variants
{
for r in 1 2 3 4; do
  for v in default chaintouch; do
    L=""; [ $v != default ] && L=$V/lib_$v.so
    echo "## rotation $r $v"; CLIMSIM_HIP_LIB=$L timeout 300 python tools/step_time.py 8192 3072 2>&1 | grep -v "$F"
  done
done
} > gpurun_out/r06_touch2_ab.txt 2>&1
cat gpurun_out/r06_touch2_ab.txt
