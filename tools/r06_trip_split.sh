#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
V=$PWD/climsim_amd/variants
{
echo "## wave stamps, CWD_PRIO=3 (turns where the tiles divide evenly)"; CLIMSIM_HIP_LIB=$V/lib_wave.so timeout 300 python tools/chainw_wave_stamps.py 3072 2>&1 | grep -v "amdgpu.ids"
for r in 1 2 3; do
  for v in default p0; do
    L=""; [ $v != default ] && L=$V/lib_$v.so
    echo "## rotation $r $v"; CLIMSIM_HIP_LIB=$L timeout 300 python tools/pub_mlp_time.py 2>&1 | grep "0, 640) B 3072\|0, 640) B 8192\|chain_fb"
  done
done
echo "## widths (default | p0)"; timeout 600 python tools/chainw_width_time.py 2>&1 | grep "chain_fb"; CLIMSIM_HIP_LIB=$V/lib_p0.so timeout 600 python tools/chainw_width_time.py 2>&1 | grep "chain_fb"
} > gpurun_out/r06_chainw_prio3.txt 2>&1
cat gpurun_out/r06_chainw_prio3.txt
