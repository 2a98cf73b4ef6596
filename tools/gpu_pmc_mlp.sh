#!/bin/bash
# PMC pass (own run, --pmc with --kernel-trace only) for the MLP step: LDS bank conflicts and MFMA busy per kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
B=${1:-8192}
cd /tmp
pass() {  # tag, counters...
  tag=$1; shift 1
  rm -rf /tmp/pmc_$tag
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python3 $REPO/bench.py --batch $B --steps 20 --warmup 5 --cpu-budget 0 --no-extras --no-profile > $REPO/gpurun_out/pmc_$tag.log 2>&1
  find /tmp/pmc_$tag -name "*counter_collection*" -exec cp {} $REPO/gpurun_out/pmc_$tag.csv \;
  echo "== $tag"; python3 $REPO/tools/pmc_summ.py $REPO/gpurun_out/pmc_$tag.csv
}
pass mlp_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
