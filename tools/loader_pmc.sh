#!/bin/bash
# Round 4: where does the device loader spend its clocks?  PMC passes (own runs, --pmc with --kernel-trace only) over bench_loader.py.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
pass() {
  name=$1; shift
  rm -rf /tmp/pmcl_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcl_$name -- python3 $REPO/bench_loader.py 64 21600 > $REPO/gpurun_out/pmcl_$name.log 2>&1
  find /tmp/pmcl_$name -name "*counter_collection*" -exec cp {} $REPO/gpurun_out/pmcl_$name.csv \;
  echo "== $name"; python3 $REPO/tools/pmc_summ.py $REPO/gpurun_out/pmcl_$name.csv 2>&1 | grep -i "loader" | head -8
}
pass busy SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass vmem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
