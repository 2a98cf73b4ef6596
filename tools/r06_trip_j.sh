#!/bin/bash
# round 6, trip J: teams - persistent workgroups per XCD a multiple of the group size (10 tiles), full-size groups first
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
for cfg in "0 0" "240 1" "256 1" "240 0" "160 1"; do
  set -- $cfg
  for r in 1 2; do echo -n "CS_CW_WGS=$1 CS_CW_TEAMS=$2 rot $r: "; CS_CW_WGS=$1 CS_CW_TEAMS=$2 timeout 300 python tools/cnn_train_time.py 512 2>&1 | tail -1; done
done
echo "== stamps CS_CW_WGS=240 CS_CW_TEAMS=1"; CS_CW_WGS=240 CS_CW_TEAMS=1 timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | grep "kernel span\|tap tiles\|sum of loops"
} > gpurun_out/r06_cw3_teams.txt 2>&1
cat gpurun_out/r06_cw3_teams.txt
