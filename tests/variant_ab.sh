#!/bin/bash
# Development: A/B library variants built under climsim_amd/variants/ within ONE GPU-box call (box-to-box variance
# is larger than most effects).  usage: variant_ab.sh "<python command>" name1 name2 ...   ("base" = the regular build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cmd=$1; shift
cp climsim_amd/libclimsim_hip.so /tmp/lib_base.so
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/lib_base.so climsim_amd/libclimsim_hip.so; else cp climsim_amd/variants/lib_$v.so climsim_amd/libclimsim_hip.so; fi
  echo "== $v"; eval "$cmd" 2>&1 | tail -${TAILN:-1}
done
done
cp /tmp/lib_base.so climsim_amd/libclimsim_hip.so
