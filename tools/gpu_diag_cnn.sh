#!/bin/bash
# PMC passes (each its own run, --pmc with --kernel-trace only) for the CNN step and the device loader.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
pass() {  # tag, script, counters...
  tag=$1; script=$2; shift 2
  rm -rf /tmp/pmc_$tag
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python3 $REPO/$script > $REPO/gpurun_out/pmc_$tag.log 2>&1
  find /tmp/pmc_$tag -name "*counter_collection*" -exec cp {} $REPO/gpurun_out/pmc_$tag.csv \;
  echo "== $tag"; python3 $REPO/tools/pmc_summ.py $REPO/gpurun_out/pmc_$tag.csv
}
pass cnn_fetch "bench_cnn.py 512 5" FETCH_SIZE
pass cnn_write "bench_cnn.py 512 5" WRITE_SIZE
pass cnn_mfma "bench_cnn.py 512 5" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass cnn_tcc "bench_cnn.py 512 5" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass ld_fetch "bench_loader.py 64 21600" FETCH_SIZE
pass ld_write "bench_loader.py 64 21600" WRITE_SIZE
pass ld_tcc "bench_loader.py 64 21600" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
