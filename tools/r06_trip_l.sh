#!/bin/bash
# round 6, trip L: cache-policy bits of k_wgrad3's operand requests (H, dZ written by the chain kernel a moment earlier)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
echo "cfg-MLP, 8192 columns: step us, {kernel: us}; variants: 1 = nt, 2 = sc1, 3 = sc0 sc1, 4 = sc0"
for r in 1 2 3; do
  echo -n "rot $r default : "; timeout 200 python tools/step_time.py 8192 2>&1 | tail -1
  for i in 1 2 3 4; do echo -n "rot $r variant $i: "; CLIMSIM_HIP_LIB=$PWD/climsim_amd/variants/lib_w3mod$i.so timeout 200 python tools/step_time.py 8192 2>&1 | tail -1; done
done
} > gpurun_out/r06_wgrad3_policy.txt 2>&1
cat gpurun_out/r06_wgrad3_policy.txt
