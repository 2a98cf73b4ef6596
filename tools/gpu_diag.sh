#!/bin/bash
# PMC passes (each its own run, --pmc with --kernel-trace only) for the bench command at batch $1.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
B=${1:-8192}
cd /tmp
pass() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -- python3 $REPO/bench.py --batch $B --steps 20 --warmup 5 --cpu-budget 0 --no-profile --no-extras --train-only > $REPO/gpurun_out/pmc_$name.log 2>&1
  find /tmp/pmc_$name -name "*counter_collection*" -exec cp {} $REPO/gpurun_out/pmc_${name}_b$B.csv \;
  python3 $REPO/tools/pmc_summ.py $REPO/gpurun_out/pmc_${name}_b$B.csv
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
