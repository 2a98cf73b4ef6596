#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
B=${1:-8192}
cd /tmp
pass() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  CS_CHAIN_ABLATE=${ABL:-0} timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -- python3 $REPO/bench.py --batch $B --steps 10 --warmup 3 --cpu-budget 0 --no-profile > $REPO/gpurun_out/pmc_$name.log 2>&1
  find /tmp/pmc_$name -name "*counter_collection*" -exec cp {} $REPO/gpurun_out/pmc_${name}_b$B.csv \;
  python3 $REPO/tests/pmc_summ.py $REPO/gpurun_out/pmc_${name}_b$B.csv | grep chain
}
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass b SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES
