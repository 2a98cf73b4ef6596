#!/bin/bash
# Diagnostics: PMC counters for the engine kernels (several passes; each pass its own run).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
B=${1:-8192}
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|TA|TD|GRBM|SPI|MALL|EA)_[A-Za-z0-9_]+" | sort -u > $REPO/gpurun_out/counters_list.txt
wc -l $REPO/gpurun_out/counters_list.txt
pass() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -- python3 $REPO/bench.py --batch $B --steps 10 --warmup 3 --cpu-budget 0 --no-profile > $REPO/gpurun_out/pmc_$name.log 2>&1
  find /tmp/pmc_$name -name "*counter_collection*" -exec cp {} $REPO/gpurun_out/pmc_$name.csv \;
  python3 $REPO/tests/pmc_summ.py $REPO/gpurun_out/pmc_$name.csv
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
