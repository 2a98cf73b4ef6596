#!/bin/bash
# One GPU-box trip for the CNN side figures: bench line, rocprofv3 kernel stats, MFMA / LDS PMC pass (each --pmc pass its own run).
set -u
R=r${1:-05}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench_cnn.py 512 40 2>&1 | tail -1 > gpurun_out/${R}_cnn_bench_b512.json; cut -c1-400 gpurun_out/${R}_cnn_bench_b512.json
cd /tmp && rm -rf /tmp/profc && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc -- python3 $REPO/bench_cnn.py 512 40 > $REPO/gpurun_out/rocprof_cnn.log 2>&1
cd $REPO; find /tmp/profc -name "*kernel_stats*" -exec cp {} gpurun_out/${R}_cnn_rocprofv3_kernel_stats_b512.csv \;
head -8 gpurun_out/${R}_cnn_rocprofv3_kernel_stats_b512.csv | cut -c1-150
cd /tmp && rm -rf /tmp/pmc_cnn && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_cnn -- python3 $REPO/bench_cnn.py 512 5 > $REPO/gpurun_out/pmc_cnn_mfma.log 2>&1
cd $REPO; find /tmp/pmc_cnn -name "*counter_collection*" -exec cp {} gpurun_out/pmc_cnn_mfma.csv \;
python3 tools/pmc_summ.py gpurun_out/pmc_cnn_mfma.csv | tee gpurun_out/${R}_pmc_cnn_mfma.txt | head -30
